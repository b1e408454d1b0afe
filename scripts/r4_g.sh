#!/bin/bash
# full default bench line (as the driver runs it) on the current tree
O=gpurun_out
python bench.py --gpus 1 --steps 20 --warmup 5 --routing $O/routing_r4.json > $O/r4_g_bench.json 2> $O/r4_g_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4_g_bench.json')); r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], r['by_bound']['mfma_bound_layers'], r['by_bound']['hbm_bound_layers'])
print('tape', d.get('launch_tape',{}).get('value'))
print('pose', d.get('pose_err_vs_fp32_path',{}).get('bench_workload'))
b=d.get('boundary',{}); print('boundary', {k:v for k,v in b.items() if k!='one_pair_per_call'}); print('one pair', b.get('one_pair_per_call'))
print('other', {k:(v.get('value'), v.get('ms_per_step')) for k,v in d.get('other_configs',{}).items()})
print('cpu', d.get('cpu_baseline'), d.get('fp32_parity_path'))
PY
