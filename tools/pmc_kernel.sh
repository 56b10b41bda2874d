#!/bin/bash
# Per-kernel counters of one bench extra on the GPU box (via gpurun): usage tools/pmc_kernel.sh <tag> <kernel substring> <bench args...>
export TMPDIR=/tmp
R=$PWD; TAG=$1; KSUB=$2; shift 2
O=$R/gpurun_out/pmck_$TAG; rm -rf $O; mkdir -p $O; cd /tmp
CMD="python $R/bench.py --no-cpu --steps 2 --warmup 1 $*"
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d $O/rd -o b --output-format csv -- $CMD > $O/rd.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/sq -o b --output-format csv -- $CMD > $O/sq.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM -d $O/sq2 -o b --output-format csv -- $CMD > $O/sq2.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/tr -o b --output-format csv -- $CMD > $O/tr.log 2>&1
find $O -name "*.db" -delete; cd $R
python - $O "$KSUB" <<'PY'
import collections, csv, glob, os, sys
src, ksub = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if ksub not in k: continue
        k = k[:90]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k][row["Counter_Name"]].add(row["Dispatch_Id"])
for k in acc:
    print(k)
    for c in sorted(acc[k]): print(f"   {c:28s} {acc[k][c]/len(n[k][c]):.4g} per launch ({len(n[k][c])} launches)")
for f in glob.glob(os.path.join(src, "tr", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if ksub in row["Name"]: print("   trace:", row["Name"][:80], "calls", row["Calls"], "avg ns", row["AverageNs"])
PY
