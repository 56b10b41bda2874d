#!/bin/bash
# gpurun_out/profiles_new/ (tools/collect_profiles.sh, merged back from the GPU box) -> profiles/ under this round's names
# usage: tools/publish_profiles.sh r04
T=${1:-r04}; S=gpurun_out/profiles_new; D=profiles
[ -d $S ] || { echo "no $S"; exit 1; }
cp $S/pmc_latest.json $D/pmc_latest.json
cp $S/pmc_bench.md $D/bench_${T}_pmc.md
cp $S/kernel_stats_bench.csv $D/bench_${T}_kernel_stats.csv
cp $S/kernel_stats_full.csv $D/bench_${T}_kernel_stats_full.csv
cp $S/line_bench.json $D/bench_${T}_line_under_rocprof.json
cp $S/line_full.json $D/bench_${T}_full_line_under_rocprof.json
for v in default k8 dropped; do cp $S/pmc_fm_$v.md $D/fm_count_${T}_pmc_$v.md; cp $S/stdout_fm_$v.txt $D/fm_count_${T}_probe_$v.txt; done
for v in rank select; do cp $S/pmc_rrr_$v.md $D/rrr_bucketed_${T}_pmc_$v.md; cp $S/stdout_rrr_$v.txt $D/rrr_bucketed_${T}_probe_$v.txt; done
cp $S/pmc_wt.md $D/wt_${T}_pmc.md
ls -la $D | grep $T
