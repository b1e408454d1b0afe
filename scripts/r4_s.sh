#!/bin/bash
O=gpurun_out
python -m pytest tests/test_kernels_gpu.py tests/test_stages_gpu.py tests/test_e2e_gpu.py -x -q -k "branch_tail or posenet or bench_configuration or in_flight or hip_graph" 2>&1 | tail -3
F="--no-cpu-baseline --no-fp32-path --no-other-configs --routing $O/routing_r4.json --steps 40"
for u in 0 1 0 1; do
NOPESAC_BRANCH_TAIL_UNFUSED=$u python bench.py $F 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); bw=d['pose_err_vs_fp32_path']['bench_workload']
print('unfused=$u', d['value'], d['ms_per_step'], 'tape', d['launch_tape']['value'], 'one pair', d['boundary']['one_pair_per_call'], {k:(bw[k]['R_err_deg_mean'], bw[k]['R_err_deg_max']) for k in ('camera_init','camera')})"
done | tee $O/r4_s_branch_tail_ab.txt
