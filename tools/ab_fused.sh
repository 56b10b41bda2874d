#!/bin/bash
# Runs ON the GPU box: builds of the library with different forms of the fused wavelet-tree lines side by side
# (sdsl-lite_amd/lib/libsdsl_hip_<variant>.so): the tests that walk the lines, then the probes.
# usage: tools/ab_fused.sh <tests|probes|all> variant...
W=${1:-all}; shift
O=gpurun_out/ab_fused; mkdir -p $O
T="tests/test_gpu_wt_layouts.py tests/test_gpu_random_campaign.py tests/test_gpu_wt_sorted.py tests/test_gpu_fm_fast.py tests/test_gpu_fm_footprint.py tests/test_gpu_fm_verify.py tests/test_gpu_parity.py"
for V in ${@:-k4 k3}; do
  export SDSL_HIP_LIB=$PWD/sdsl-lite_amd/lib/libsdsl_hip_$V.so
  if [ $W != probes ]; then
    timeout 1500 python -m pytest $T -q -m gpu > $O/pytest_$V.txt 2>&1; tail -5 $O/pytest_$V.txt
  fi
  if [ $W != tests ]; then
    for p in sa extract locate wt_select; do timeout 300 python tools/kernel_probe.py $p 2>&1 | grep "G.*/s" | sed "s/^/$V /" | tee -a $O/probes.txt; done
    timeout 600 python tools/fm_probe.py 1024 1e8 default,dropped,lean 2>&1 | grep "Mcount\|footprint" | sed "s/^/$V /" | tee -a $O/probes.txt
  fi
done
