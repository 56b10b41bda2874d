#!/bin/bash
# Quick fabric-traffic check of the headline step on the GPU box (via gpurun): two PMC passes, per-kernel MB per launch.
# usage: tools/pmc_quick.sh <tag> [extra bench args]      (environment knobs are inherited)
export TMPDIR=/tmp
R=$PWD
TAG=${1:-q}
shift
O=$R/gpurun_out/pmcq_$TAG
rm -rf $O; mkdir -p $O
cd /tmp
HEAD="python $R/bench.py --steps 3 --warmup 1 --extras none --no-cpu $*"
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d $O/rd -o b --output-format csv -- $HEAD > $O/rd.log 2>&1
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $O/wr -o b --output-format csv -- $HEAD > $O/wr.log 2>&1
find $O -name "*.db" -delete
cd $R
python - $O <<'PY'
import collections, csv, glob, os, re, sys
src = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"k_\w+(<[^>(]*>)?", row["Kernel_Name"]); k = m.group(0) if m else row["Kernel_Name"][:40]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k][row["Counter_Name"]].add(row["Dispatch_Id"])
def pl(k, c): return acc[k][c] / len(n[k][c]) if n[k][c] else 0.0
tot_r = tot_w = 0
for k in sorted(acc):
    if not (k.startswith("k_sw_") or k.startswith("k_sr_") or k.startswith("k_rs_")): continue
    r32, r64, r128, r = pl(k, "TCC_EA0_RDREQ_32B_sum"), pl(k, "TCC_EA0_RDREQ_64B_sum"), pl(k, "TCC_EA0_RDREQ_128B_sum"), pl(k, "TCC_EA0_RDREQ_sum")
    rd = (r32 * 32 + r64 * 64 + r128 * 128 + max(0.0, r - r32 - r64 - r128) * 64) / 1e6
    w, w64 = pl(k, "TCC_EA0_WRREQ_sum"), pl(k, "TCC_EA0_WRREQ_64B_sum")
    wr = (w64 * 64 + (w - w64) * 32) / 1e6
    tot_r += rd; tot_w += wr
    if rd + wr > 50: print(f"{k:40s} read {rd:9.1f} MB  write {wr:9.1f} MB")
print(f"per step: read {tot_r/1e3:.2f} GB + write {tot_w/1e3:.2f} GB = {(tot_r+tot_w)/1e3:.2f} GB")
PY
grep -h '^{' $O/rd.log | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('under pmc: kernel_ms', j['roofline']['kernel_ms'], j['roofline']['phases_ms'])"
