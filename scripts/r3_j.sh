#!/bin/bash
O=gpurun_out
python -m pytest tests/test_kernels_gpu.py -x -q -k "bottleneck_tail or backbone_fused" 2>&1 | tail -5 > $O/r3_j_pytest.log
python -m pytest tests/test_stages_gpu.py -x -q -k "backbone" 2>&1 | tail -5 >> $O/r3_j_pytest.log
cat $O/r3_j_pytest.log
{
echo "== res3.3 (CN=256)"; NOPESAC_TAIL_NO_RT8=1 python scripts/tail_one.py 64 60 80 128 512 256; python scripts/tail_one.py 64 60 80 128 512 256
echo "== res3.0 (proj C2=256 s2)"; NOPESAC_TAIL_NO_RT8=1 python scripts/tail_one.py 64 60 80 128 512 128 256 2; python scripts/tail_one.py 64 60 80 128 512 128 256 2
NOPESAC_TAIL_NO_RT8=1 python scripts/backbone_time.py
python scripts/backbone_time.py
NOPESAC_TAIL_NO_RT8=1 python scripts/backbone_time.py
python scripts/backbone_time.py
} 2>&1 | grep -v amdgpu.ids | tee $O/r3_j_timing.log
F="--no-cpu-baseline --no-boundary --no-fp32-path --no-accuracy --no-other-configs --no-tape"
python bench.py $F > $O/r3_j_bench.json 2> $O/r3_j_bench.err
NOPESAC_TAIL_NO_RT8=1 python bench.py $F > $O/r3_j_bench_no_rt8.json 2>> $O/r3_j_bench.err
for f in bench bench_no_rt8; do python -c "
import json; d=json.load(open('$O/r3_j_$f.json')); c=d['config']; print('$f', d['value'], d['ms_per_step'])"; done
