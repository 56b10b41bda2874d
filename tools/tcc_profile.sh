#!/bin/bash
# L2 / fabric counter passes for one command on the GPU box:  tools/tcc_profile.sh TAG -- cmd args...
TAG=$1; shift; shift
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/tcc_$TAG
mkdir -p $O
cd /tmp
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d $O/p1 -o x --output-format csv -- "$@" > $O/p1.log 2>&1
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum -d $O/p2 -o x --output-format csv -- "$@" > $O/p2.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/trace -o x --output-format csv -- "$@" > $O/trace.log 2>&1
cd $R
python tools/sq_summary.py $O
