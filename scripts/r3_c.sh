#!/bin/bash
O=gpurun_out
python -m pytest tests/test_kernels_gpu.py -x -q -k "bottleneck_tail or backbone_fused or conv3x3_c64 or c64" 2>&1 | tail -5 > $O/r3_c_pytest.log
python -m pytest tests/test_stages_gpu.py -x -q -k "backbone" 2>&1 | tail -5 >> $O/r3_c_pytest.log
{
python scripts/backbone_time.py
NOPESAC_TAIL_NO_RT4=1 NOPESAC_RES3_EDGES_FUSED=1 python scripts/backbone_time.py
NOPESAC_RES3_EDGES_FUSED=1 python scripts/backbone_time.py
NOPESAC_TAIL_RT4_LATE=1 python scripts/backbone_time.py
python scripts/backbone_time.py
for cfg in "64 60 80 128 512 128" "64 120 160 64 256 64" "64 120 160 64 256 128"; do
  echo "== $cfg"; NOPESAC_TAIL_RT4_LATE=1 python scripts/tail_one.py $cfg; python scripts/tail_one.py $cfg
done
echo "== proj res2.0"; NOPESAC_TAIL_NO_RT4=1 python scripts/tail_one.py 64 120 160 64 256 64 64 1;  python scripts/tail_one.py 64 120 160 64 256 64 64 1
python scripts/c64_one.py
} > $O/r3_c_timing.log 2>&1
python bench.py --no-cpu-baseline --no-boundary --no-fp32-path --no-accuracy --no-other-configs > $O/r3_c_bench.json 2> $O/r3_c_bench.err
NOPESAC_TAIL_NO_RT4=1 NOPESAC_RES3_EDGES_FUSED=1 python bench.py --no-cpu-baseline --no-boundary --no-fp32-path --no-accuracy --no-other-configs > $O/r3_c_bench_old_tails.json 2>> $O/r3_c_bench.err
cat $O/r3_c_pytest.log; grep -v amdgpu.ids $O/r3_c_timing.log; for f in $O/r3_c_bench.json $O/r3_c_bench_old_tails.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f', d['value'], d['ms_per_step'], d['config']['host_launch_ms_per_step'])"; done
