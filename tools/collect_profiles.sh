#!/bin/bash
# Runs ON the GPU box (via gpurun): every profile the round commits, through ONE recipe (tools/prof.sh: a kernel trace and four
# PMC passes per command).  tools/pmc_json.py + tools/prof_summary.py turn gpurun_out/prof_*/ into profiles/.
# usage: tools/collect_profiles.sh [what ...]    what = bench fm rrr wt walks sd full (default: all)
R=$PWD
WHAT=${@:-bench fm rrr wt walks sd full}
for w in $WHAT; do
  case $w in
    bench) # the headline command without extras: every k_sw_* / k_sr_* dispatch belongs to a bucketed rank step
      tools/prof.sh bench python $R/bench.py --steps 5 --warmup 1 --extras none --no-cpu ;;
    fm)    # count() on the bench text and patterns, one table variant per run
      tools/prof.sh fm_default python $R/tools/fm_probe.py 1024 1e8 default
      tools/prof.sh fm_k8 python $R/tools/fm_probe.py 1024 1e8 k8
      tools/prof.sh fm_dropped python $R/tools/fm_probe.py 1024 1e8 dropped
      tools/prof.sh fm_lean python $R/tools/fm_probe.py 1024 1e8 lean ;;
    rrr)   # configs[2], default dispatch (bucketed), one operation per run
      tools/prof.sh rrr_rank python $R/tools/rrr_probe.py rank
      tools/prof.sh rrr_select python $R/tools/rrr_probe.py select ;;
    wt)    tools/prof.sh wt python $R/bench.py --steps 4 --warmup 1 --extras wt --no-cpu
           tools/prof.sh wt_select python $R/tools/kernel_probe.py wt_select ;;
    walks) # the LF walks behind csa[i] / extract / locate on SDSL's default samples, and count() on the rrr-compressed index
      tools/prof.sh walk_sa python $R/tools/kernel_probe.py sa
      tools/prof.sh walk_extract python $R/tools/kernel_probe.py extract
      tools/prof.sh walk_locate python $R/tools/kernel_probe.py locate
      tools/prof.sh rrr_count python $R/tools/kernel_probe.py rrr_count
      tools/prof.sh rrr_count_lean python $R/tools/kernel_probe.py rrr_count_lean ;;
    sd)    tools/prof.sh sd_rank python $R/tools/kernel_probe.py sd_rank
           tools/prof.sh sd_select1 python $R/tools/kernel_probe.py sd_select1
           tools/prof.sh sd_select0 python $R/tools/kernel_probe.py sd_select0 ;;
    full)  # kernel trace only, of the default command
      export TMPDIR=/tmp; O=$R/gpurun_out/prof_full; rm -rf $O; mkdir -p $O; cd /tmp
      timeout -k 10 900 rocprofv3 --kernel-trace --stats -d $O/trace -o p --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu > $O/stdout.txt 2> $O/trace.err
      find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; cd $R ;;
  esac
done
# condensed results (what travels back: gpurun merges at most 64 MiB): the json of bench.py's roofline blocks, one per-kernel
# table per profiled command, the kernel-trace statistics; the raw counter files stay on the box
OUT=$R/gpurun_out/profiles_new; rm -rf $OUT; mkdir -p $OUT
python $R/tools/pmc_json.py > $OUT/pmc_json.log 2>&1; cp $R/profiles/pmc_latest.json $OUT/ 2>/dev/null
for d in $R/gpurun_out/prof_*/; do
  t=$(basename $d); t=${t#prof_}
  [ -d $d/rd ] && python $R/tools/prof_summary.py $t --min-ms 0.02 --md $OUT/pmc_$t.md --json $OUT/pmc_$t.json --stats-csv $OUT/kernel_stats_$t.csv > /dev/null 2>&1
  [ -d $d/rd ] || python $R/tools/prof_summary.py $t --stats-csv $OUT/kernel_stats_$t.csv > /dev/null 2>&1
  grep -h "^{" $d/stdout.txt 2>/dev/null | tail -1 > $OUT/line_$t.json; [ -s $OUT/line_$t.json ] || rm -f $OUT/line_$t.json
  grep -h "Mcount/s\|Gq/s\|G.*/s\|PROBE_UNITS" $d/stdout.txt 2>/dev/null > $OUT/stdout_$t.txt; [ -s $OUT/stdout_$t.txt ] || rm -f $OUT/stdout_$t.txt
done
[ -z "$PROF_KEEP_RAW" ] && rm -rf $R/gpurun_out/prof_*
