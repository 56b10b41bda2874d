#!/bin/bash
# Runs ON the GPU box: the LF walks (csa[i], extract, locate on samples) on builds of the library with different forms of the fused lines
O=gpurun_out/ab_fused; mkdir -p $O
for V in ${@:-k4 k3}; do
  export SDSL_HIP_LIB=$PWD/sdsl-lite_amd/lib/libsdsl_hip_$V.so
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wt_layouts.py tests/test_gpu_random_campaign.py tests/test_gpu_beyond_2_32.py -q -m gpu -k "fm or csa or locate or extract or index or symbols" > $O/pytest_walks_$V.txt 2>&1; tail -3 $O/pytest_walks_$V.txt
  for p in sa extract locate; do timeout 300 python tools/kernel_probe.py $p 2>&1 | grep "G.*/s" | sed "s/^/$V /" | tee -a $O/walks.txt; done
done
