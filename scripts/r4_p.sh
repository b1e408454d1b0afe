#!/bin/bash
O=gpurun_out
python -m pytest tests/test_kernels_gpu.py -x -q -k "stem" 2>&1 | tail -3 | tee $O/r4_p_pytest.log
python scripts/stem_one.py 2>&1 | grep -v amdgpu.ids | tail -3
F="--no-cpu-baseline --no-boundary --no-fp32-path --no-other-configs --no-tape --routing $O/routing_r4.json --steps 40"
for fold in 1 0 1 0; do
  NOPESAC_STEM_FOLDED=$fold python bench.py $F 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); bw=d['pose_err_vs_fp32_path']['bench_workload']
print('folded $fold', d['value'], d['ms_per_step'], {k:(bw[k]['R_err_deg_mean'], bw[k]['R_err_deg_max'], bw[k]['T_err_max']) for k in ('camera_init','camera_initRec','camera')})
print('   loose', {k: v for k, v in d['pose_err_vs_fp32_path'].items() if k != 'bench_workload'})" | cut -c1-900
done | tee $O/r4_p_stem_folded_ab.txt
