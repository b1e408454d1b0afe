#!/bin/bash
# res4 identity tails fused on the eight-wave 128-pixel chunked kernel (pw_chain_rt8_kernel<256, 1024, 256>: y never re-read by the next
# reduce conv, -157 MB of HBM reads per block) vs the un-fused p8 launches, in the four-in-flight loop.  Same routing file.
O=gpurun_out; F="--no-cpu-baseline --no-boundary --no-fp32-path --no-accuracy --no-other-configs --no-tape --steps 40 --warmup 8"
for rep in 1 2; do for v in unfused fused_rt8 fused_stream; do
  case $v in unfused) E="";; fused_rt8) E="NOPESAC_TAIL_RES4_FUSED=1 NOPESAC_TAIL_RT8_WIDE=1";; fused_stream) E="NOPESAC_TAIL_RES4_FUSED=1";; esac
  env $E python bench.py $F > $O/res4_ab_$v.json 2>> $O/res4_ab.err
  python - <<PY
import json
d=json.load(open('$O/res4_ab_$v.json')); r=d['roofline']; cf=r['conv_family']
print('$v rep $rep', d['value'], 'pairs/s', d['ms_per_step'], 'ms | conv family', cf['ms'], 'ms', cf['launches_per_step'], 'launches')
PY
done; done
