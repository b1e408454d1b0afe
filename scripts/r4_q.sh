#!/bin/bash
O=gpurun_out
python -m pytest tests/test_stages_gpu.py tests/test_kernels_gpu.py -x -q -k "posenet or pose or sinkhorn or matcher or bf16" 2>&1 | tail -3
F="--no-cpu-baseline --no-boundary --no-fp32-path --no-other-configs --no-tape --routing $O/routing_r4.json --steps 40"
python bench.py $F --layers $O/r4_q_layers.tsv 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); bw=d['pose_err_vs_fp32_path']['bench_workload']
print('pad320', d['value'], d['ms_per_step'], d['config']['routing_entries_measured_now'], {k:(bw[k]['R_err_deg_mean'], bw[k]['R_err_deg_max']) for k in ('camera_init','camera_initRec','camera')})"
grep "15, 20, 3" $O/r4_q_layers.tsv | cut -c1-160
python bench.py $F 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('again', d['value'], d['ms_per_step'])"
