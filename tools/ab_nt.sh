#!/bin/bash
# Runs ON the GPU box: line loads with / without the non-temporal hint, both forms of the fused lines; then read-request counters of csa[i]
O=gpurun_out/ab_fused; mkdir -p $O
export SDSL_HIP_WALK_LANES=0
for V in k4nt k4 k3nt k3; do
  export SDSL_HIP_LIB=$PWD/sdsl-lite_amd/lib/libsdsl_hip_$V.so
  for p in sa extract; do timeout 300 python tools/kernel_probe.py $p 2>&1 | grep "G.*/s" | sed "s/^/$V /" | tee -a $O/nt.txt; done
  timeout 600 python tools/fm_probe.py 1024 1e8 default,dropped,lean 2>&1 | grep "Mcount" | sed "s/^/$V /" | tee -a $O/nt.txt
done
export TMPDIR=/tmp
for V in k4 k4nt k3; do
  export SDSL_HIP_LIB=$PWD/sdsl-lite_amd/lib/libsdsl_hip_$V.so
  P=$PWD/gpurun_out/prof_nt_$V; rm -rf $P; mkdir -p $P; R=$PWD; cd /tmp
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_HIT_sum TCC_MISS_sum -d $P/rd -o p --output-format csv -- python $R/tools/kernel_probe.py sa > $P/rd.log 2>&1
  cd $R
  python - <<PY | tee -a $O/nt.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$P/rd/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        if "k_fm_walk" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    print("$V", k, {c: "%.3e per launch" % (v / max(1, n[(k, c)])) for c, v in d.items()})
PY
  rm -rf $P
done
