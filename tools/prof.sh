#!/bin/bash
# Runs ON the GPU box (via gpurun): one kernel trace + four PMC passes of a command, each pass on its own (gpurun refuses
# --pmc together with trace domains; TCC has four slots, SQ eight).  Output: gpurun_out/prof_<tag>/{trace,rd,wr,sq,sq2}/ and the
# command's stdout of the traced run in gpurun_out/prof_<tag>/stdout.txt.  tools/prof_summary.py condenses it.
# usage: tools/prof.sh <tag> <command ...>
export TMPDIR=/tmp
R=$PWD; TAG=$1; shift
O=$R/gpurun_out/prof_$TAG; rm -rf $O; mkdir -p $O; cd /tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats -d $O/trace -o p --output-format csv -- "$@" > $O/stdout.txt 2> $O/trace.err; echo "trace exit=$?"
timeout -k 10 900 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d $O/rd -o p --output-format csv -- "$@" > $O/rd.log 2>&1; echo "rd exit=$?"
timeout -k 10 900 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum -d $O/wr -o p --output-format csv -- "$@" > $O/wr.log 2>&1; echo "wr exit=$?"
timeout -k 10 900 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/sq -o p --output-format csv -- "$@" > $O/sq.log 2>&1; echo "sq exit=$?"
timeout -k 10 900 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES -d $O/sq2 -o p --output-format csv -- "$@" > $O/sq2.log 2>&1; echo "sq2 exit=$?"
find $O -name "*.db" -delete
# the per-dispatch trace can be large: keep the statistics, drop the rest unless asked
[ -z "$PROF_KEEP_TRACE" ] && find $O/trace -name "*kernel_trace.csv" -size +8M -delete
cd $R
