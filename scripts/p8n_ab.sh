#!/bin/bash
# 256x128-tile kernel A/B in the four-in-flight loop: routing B = profiles/routing_r5.json as committed; routing A = the same file with
# every entry the new kernel is eligible for removed, so that ONLY those shapes are tuned again (now with configurations 13 / 14 on offer).
O=gpurun_out; mkdir -p $O
F="--no-cpu-baseline --no-boundary --no-fp32-path --no-accuracy --no-other-configs --no-tape --steps 40 --warmup 8"
python - <<'PY'
import json
d=json.load(open('profiles/routing_r5.json'))
keep={}
for k,v in d['routing'].items():
    f=k.split('|')
    elig = f[0]=='bfloat16' and f[1]=='bfloat16' and f[2]=='bfloat16' and f[12]=='False' and int(f[7])%128==0 and int(f[6])%64==0 and f[15]=='False' and int(f[18]) in (0,1,2)
    if not elig: keep[k]=v
print('entries', len(d['routing']), '-> kept', len(keep))
d['routing']=keep
json.dump(d,open('gpurun_out/routing_p8n_A.json','w'),indent=1)
PY
python bench.py $F --routing $O/routing_p8n_A.json > $O/p8n_ab_tune.json 2> $O/p8n_ab.err
python - <<'PY'
import json
a=json.load(open('gpurun_out/routing_p8n_A.json'))['routing']; b=json.load(open('profiles/routing_r5.json'))['routing']
for k in sorted(a):
    if a[k]!=b.get(k): print('  ', k, b.get(k), '->', a[k])
PY
for rep in 1 2; do for r in A B; do
  RT=$([ $r = A ] && echo $O/routing_p8n_A.json || echo profiles/routing_r5.json)
  NOPESAC_P8N=$([ $r = A ] && echo 1 || echo 0) python bench.py $F --routing $RT --layers $O/p8n_ab_${r}_layers.tsv > $O/p8n_ab_$r$rep.json 2>> $O/p8n_ab.err
  python - <<PY
import json
d=json.load(open('$O/p8n_ab_$r$rep.json')); r=d['roofline']; b=r['by_bound']; cf=r['conv_family']
print('$r$rep', d['value'], 'pairs/s', d['ms_per_step'], 'ms | conv family', cf['ms'], 'ms', cf['TFLOP/s'], 'TF | p8 frac', r['frac'], 'other', {k:(v['ms'],v['TFLOP/s']) for k,v in r['other_kernels'].items()})
PY
done; done
