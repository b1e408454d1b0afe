#!/bin/bash
# bf16 pose error budget: MODEL.AMD.POSE_FP32_PARTS "" vs "aim" (the two re-embedding MLPs on f32 operands) on the benchmark workload
O=gpurun_out; F="--no-cpu-baseline --no-boundary --no-fp32-path --no-other-configs --no-tape --steps 30 --warmup 6"
for parts in "" "aim" "aim fc"; do
  python bench.py $F --pose-fp32-parts "$parts" > $O/aim_ab.json 2>> $O/aim_ab.err
  python - <<PY
import json
d=json.load(open('$O/aim_ab.json'))
bw=d['pose_err_vs_fp32_path']['bench_workload']
print(repr("$parts"), d['value'], 'pairs/s', d['ms_per_step'], 'ms |', {k:(round(v['R_err_deg_mean'],3), round(v['R_err_deg_max'],3), round(v.get('T_err_mean',0),4), round(v.get('T_err_max',0),4)) for k,v in bw.items() if isinstance(v, dict)})
PY
done
