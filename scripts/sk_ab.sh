#!/bin/bash
# Stream-K A/B in the four-in-flight loop: ONE tuning pass with the stream-K form on offer (routing A), the same routing with every
# stream-K entry turned back into the plain p8 kernel (routing B), then the headline loop on A, B, A, B on the same box.
O=gpurun_out; mkdir -p $O
F="--no-cpu-baseline --no-boundary --no-fp32-path --no-accuracy --no-other-configs --no-tape --steps 40 --warmup 8"
NOPESAC_P8_SK=1 python bench.py $F --retune --routing $O/routing_sk_A.json > $O/sk_ab_tune.json 2> $O/sk_ab.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/routing_sk_A.json'))
n=sum(1 for v in d['routing'].values() if v==12)
d['routing']={k:(11 if v==12 else v) for k,v in d['routing'].items()}
json.dump(d,open('gpurun_out/routing_sk_B.json','w'),indent=1)
print('stream-K entries in routing A:', n)
for k,v in json.load(open('gpurun_out/routing_sk_A.json'))['routing'].items():
    if v==12: print('  ', k)
PY
for rep in 1 2; do for r in A B; do
  NOPESAC_P8_SK=$([ $r = A ] && echo 1 || echo 0) python bench.py $F --routing $O/routing_sk_$r.json --layers $O/sk_ab_${r}_layers.tsv > $O/sk_ab_$r$rep.json 2>> $O/sk_ab.err
  python - <<PY
import json
d=json.load(open('$O/sk_ab_$r$rep.json')); r=d['roofline']; b=r['by_bound']
print('$r$rep', d['value'], 'pairs/s', d['ms_per_step'], 'ms | p8 frac', r['frac'], 'mfma-bound', b['mfma_bound_layers']['TFLOP/s'], b['mfma_bound_layers']['ms'], 'hbm-bound', b['hbm_bound_layers']['TB/s_algorithmic'], b['hbm_bound_layers']['ms'], 'sclk', (r.get('engine_clock') or {}).get('sclk_mhz_under_benchmark_load'))
PY
done; done
