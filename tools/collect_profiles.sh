#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + PMC passes of the bench command.
# Outputs land in gpurun_out/prof_$TAG/ ; tools/summarize_profiles.py turns them into profiles/*.
TAG=${1:-r01}
Q=${2:-1000000000}
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp
BENCH="python $R/bench.py --steps 5 --warmup 1 --extras select --no-cpu --queries $Q"
# the trace of the headline command has no extras, so that every k_rank launch in it is a full step and the
# --stats average can be compared directly with bench.py's own HIP-event figure (roofline.kernel_ms)
rocprofv3 --kernel-trace --stats -d $O/trace -o bench --output-format csv -- python $R/bench.py --steps 5 --warmup 1 --extras none --no-cpu --queries $Q > $O/bench_trace.log 2>&1
echo "trace exit=$?"
rocprofv3 --kernel-trace --stats -d $O/trace_select -o bench --output-format csv -- $BENCH > $O/bench_trace_select.log 2>&1
echo "trace_select exit=$?"
# PMC passes: counters only (no trace domains), separate runs per counter group (TCC has 4 slots)
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d $O/pmc_rd -o bench --output-format csv -- $BENCH > $O/bench_pmc_rd.log 2>&1
echo "pmc_rd exit=$?"
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum -d $O/pmc_wr -o bench --output-format csv -- $BENCH > $O/bench_pmc_wr.log 2>&1
echo "pmc_wr exit=$?"
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o bench --output-format csv -- $BENCH > $O/bench_pmc_fetch.log 2>&1
echo "pmc_fetch exit=$?"
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o bench --output-format csv -- $BENCH > $O/bench_pmc_write.log 2>&1
echo "pmc_write exit=$?"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/pmc_sq -o bench --output-format csv -- $BENCH > $O/bench_pmc_sq.log 2>&1
echo "pmc_sq exit=$?"
grep -h '^{' $O/bench_trace.log | tail -1 > $O/bench_line_under_trace.json
cd $R
ls $O
# kernel trace of the FULL default bench (all extras): per-kernel averages of the secondary kernels (wavelet tree, FM-index,
# rrr, sd_vector, locate/extract walks)
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/trace_full -o bench --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu > $O/bench_trace_full.log 2>&1
echo "trace_full exit=$?"
cd $R
