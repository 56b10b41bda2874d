#!/bin/bash
# Runs ON the GPU box: counters of one probe on builds of the library with different forms of the fused lines (one pass per counter
# set, each under a timeout).  usage: [PROBE="sa"] [SET=tcc|sq] tools/ab_pmc.sh variant...
O=gpurun_out/ab_fused; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
case ${SET:-tcc} in
  tcc) C="TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" ;;
  sq)  C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" ;;
esac
for V in ${@:-k4 k3}; do
  export SDSL_HIP_LIB=$R/sdsl-lite_amd/lib/libsdsl_hip_$V.so
  P=$R/gpurun_out/prof_pmc_$V; rm -rf $P; mkdir -p $P; cd /tmp
  timeout -k 10 240 rocprofv3 --pmc $C -d $P/rd -o p --output-format csv -- python $R/tools/kernel_probe.py ${PROBE:-sa} > $P/rd.log 2>&1; echo "$V exit=$?"
  cd $R
  python - <<PY | tee -a $O/pmc.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$P/rd/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        if "k_fm_walk" not in k and "k_fm_count_flat" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    print("$V", k[:50], {c: "%.4e" % (v / max(1, n[(k, c)])) for c, v in d.items()}, "launches", max(n.values()))
PY
  grep -h "F2026\|Could not" $P/rd.log | head -3 | cut -c1-200; rm -rf $P
done
