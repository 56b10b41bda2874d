#!/bin/bash
# A/B of two builds of the library on ONE box (boxes differ by +-2 %): sdsl-lite_amd/lib/A.so against sdsl-lite_amd/lib/B.so,
# alternating.  usage (via gpurun): tools/ab_bench.sh [bench args]     (default: the headline only)
L=sdsl-lite_amd/lib
ARGS=${*:---extras none --no-cpu --steps 8}
for v in ${VARIANTS:-A B A B}; do
  cp $L/$v.so $L/libsdsl_hip.so
  python bench.py $ARGS --sidecar /tmp/ab_$v.json 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('$v', 'headline kernel ms', round(r['kernel_ms_per_step']['median'],3), 'digest', j['reference_digest_match'])
sec=j.get('secondary') or {}
print('    count Mcount/s', sec.get('value'), 'lean', (sec.get('lean') or {}).get('Mcount/s'))
for k,v in (j.get('summary') or {}).items(): print('   ',k, {a:b for a,b in v.items() if not isinstance(b,bool)})
try:
    ex=json.load(open('/tmp/ab_$v.json'))['extras']; r=ex.get('fm_count_repetitive') or {}
    print('    repetitive text:', {k: (round(r[k]['Mcount/s'],1), r[k]['reference_digest_match']) for k in ('count_default','count_sa_dropped','count_lean') if k in r})
    print('    kmer8:', round(ex['fm_count_kmer8']['Mcount/s'],1), ex['fm_count_kmer8']['reference_digest_match'])
except Exception as e: print('    (no sidecar)', e)
"
done
cp $L/B.so $L/libsdsl_hip.so
