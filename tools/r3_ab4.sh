#!/bin/bash
# same-box A/B of probe builds over a workload: tools/r3_ab4.sh <workload> [extra bench args]
W=$1; shift
for so in tools/probe_libs/*.so; do
  n=$(basename $so .so)
  for rep in 1 2; do
    TROYHIP_LIB=$PWD/$so python bench.py --workload $W --steps 20 --warmup 3 --no-cpu-baseline --no-verify "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1])
ks={k['name']:k['us'] for k in d['roofline']['per_kernel']}
top=sorted(ks.items(), key=lambda kv:-kv[1])[:4]
print('$n', d['value'], d['unit'], ' '.join('%s=%.0f'%(k.replace('ntt2_','').replace('_kernel',''),v) for k,v in top))"
  done
done
