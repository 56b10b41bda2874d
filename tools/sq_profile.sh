#!/bin/bash
# SQ counter passes for one command on the GPU box:  tools/sq_profile.sh TAG -- cmd args...
# Prints per-kernel sums (tools/sq_summary.py) and leaves the csv files in gpurun_out/sq_TAG/.
TAG=$1; shift; shift
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/sq_$TAG
mkdir -p $O
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/p1 -o x --output-format csv -- "$@" > $O/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES -d $O/p2 -o x --output-format csv -- "$@" > $O/p2.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_WAVE_CYCLES -d $O/p3 -o x --output-format csv -- "$@" > $O/p3.log 2>&1
cd $R
python tools/sq_summary.py $O
