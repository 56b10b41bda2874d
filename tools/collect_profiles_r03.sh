#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + PMC passes of the bench command, round 3.
# Outputs land in gpurun_out/prof_r03/ ; tools/summarize_profiles_r03.py turns them into profiles/*.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/prof_r03
mkdir -p $O
cd /tmp
# the headline command without extras: every bucketed-rank kernel in the trace belongs to a full step, so the sum of
# the per-kernel averages x launches per step can be compared with bench.py's own HIP-event figure (roofline.kernel_ms)
HEAD="python $R/bench.py --steps 5 --warmup 1 --extras none --no-cpu"
rocprofv3 --kernel-trace --stats -d $O/trace -o bench --output-format csv -- $HEAD > $O/bench_trace.log 2>&1
echo "trace exit=$?"
grep -h '^{' $O/bench_trace.log | tail -1 > $O/bench_line_under_trace.json
# PMC passes: counters only (no trace domains), separate runs per counter group (TCC has 4 slots)
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d $O/pmc_rd -o bench --output-format csv -- $HEAD > $O/bench_pmc_rd.log 2>&1
echo "pmc_rd exit=$?"
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum -d $O/pmc_wr -o bench --output-format csv -- $HEAD > $O/bench_pmc_wr.log 2>&1
echo "pmc_wr exit=$?"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_sq -o bench --output-format csv -- $HEAD > $O/bench_pmc_sq.log 2>&1
echo "pmc_sq exit=$?"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES -d $O/pmc_sq2 -o bench --output-format csv -- $HEAD > $O/bench_pmc_sq2.log 2>&1
echo "pmc_sq2 exit=$?"
# kernel trace of the FULL default bench (all extras): per-kernel averages of the secondary kernels
rocprofv3 --kernel-trace --stats -d $O/trace_full -o bench --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu > $O/bench_trace_full.log 2>&1
echo "trace_full exit=$?"
# fabric traffic of the fused wavelet-tree / FM-index kernels: two PMC passes of the wt + fm extras
FULL="python $R/bench.py --steps 3 --warmup 1 --no-cpu --extras rrr,wt,fm"
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d $O/pmcfull_rd -o bench --output-format csv -- $FULL > $O/bench_pmcfull_rd.log 2>&1
echo "pmcfull_rd exit=$?"
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $O/pmcfull_wr -o bench --output-format csv -- $FULL > $O/bench_pmcfull_wr.log 2>&1
echo "pmcfull_wr exit=$?"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmcfull_sq -o bench --output-format csv -- $FULL > $O/bench_pmcfull_sq.log 2>&1
echo "pmcfull_sq exit=$?"
find $O -name "*.db" -delete
cd $R
python tools/summarize_profiles_r03.py
