#!/bin/bash
# four-in-flight loop with the 1x1 (HBM-bound) p8 launches capped at n persistent workgroups; routing B of the stream-K A/B (plain p8)
O=gpurun_out; F="--no-cpu-baseline --no-boundary --no-fp32-path --no-accuracy --no-other-configs --no-tape --steps 40 --warmup 8"
for rep in 1 2; do for cap in 0 192 160 128 96; do
  NOPESAC_P8_SK=0 NOPESAC_P8_CAP_1X1=$cap python bench.py $F --routing profiles/routing_r5.json > $O/cap_ab_$cap.json 2>> $O/cap_ab.err
  python - <<PY
import json
d=json.load(open('$O/cap_ab_$cap.json')); r=d['roofline']; b=r['by_bound']
print('cap $cap rep $rep', d['value'], 'pairs/s', d['ms_per_step'], 'ms | hbm-bound', b['hbm_bound_layers']['TB/s_algorithmic'], b['hbm_bound_layers']['ms'], 'sclk', (r.get('engine_clock') or {}).get('sclk_mhz_under_benchmark_load'))
PY
done; done
