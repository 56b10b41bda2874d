#!/bin/bash
# After `gpurun -- 'tools/collect_profiles.sh; pytest ... > gpurun_out/pytest_gpu_rNN.txt; python bench.py > gpurun_out/bench_rNN_line.json; ...'`:
# copies what the round commits from gpurun_out/ (scratch) into profiles/ (tracked) under the round's names.
# usage: tools/publish_profiles.sh r05
R=${1:?round tag, e.g. r05}
N=gpurun_out/profiles_new
for f in $N/pmc_*.md; do
  t=$(basename $f .md); t=${t#pmc_}
  cp $f profiles/${t}_${R}_pmc.md
  [ -f $N/stdout_$t.txt ] && cp $N/stdout_$t.txt profiles/${t}_${R}_probe.txt
  [ -f $N/kernel_stats_$t.csv ] && cp $N/kernel_stats_$t.csv profiles/${t}_${R}_kernel_stats.csv
done
cp $N/kernel_stats_full.csv profiles/bench_${R}_kernel_stats_full.csv
[ -f $N/line_full.json ] && cp $N/line_full.json profiles/bench_${R}_line_under_rocprof.json
cp gpurun_out/pmc_latest_${R}.json profiles/pmc_latest.json
cp gpurun_out/bench_${R}_line.json profiles/bench_${R}_line.json
cp gpurun_out/bench_extras_${R}.json profiles/bench_extras_${R}.json
cp gpurun_out/pytest_gpu_${R}.txt profiles/pytest_gpu_${R}.txt
python - <<PY
import json, sys
sys.path.insert(0, ".")
import bench_common
print("pmc sha", json.load(open("profiles/pmc_latest.json"))["kernel_sources_sha"], "sources", bench_common.kernel_sources_sha())
PY
