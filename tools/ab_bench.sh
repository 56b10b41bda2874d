#!/bin/bash
# A/B of two builds of the library on ONE box (boxes differ by +-2 %): sdsl-lite_amd/lib/A.so against sdsl-lite_amd/lib/B.so,
# alternating.  usage (via gpurun): tools/ab_bench.sh [bench args]     (default: the headline only)
L=sdsl-lite_amd/lib
ARGS=${*:---extras none --no-cpu --steps 8}
for v in ${VARIANTS:-A B A B}; do
  cp $L/$v.so $L/libsdsl_hip.so
  python bench.py $ARGS 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('$v', round(r['kernel_ms_per_step']['median'],3), {k: round(v,3) for k,v in r['phases_ms'].items()}, j['reference_digest_match'])
for k,v in j.get('extras',{}).items():
    if isinstance(v,dict) and 'kernel_ms' in v: print('   ',k, round(v['kernel_ms'],3), v.get('phases_ms'))
"
done
cp $L/B.so $L/libsdsl_hip.so
