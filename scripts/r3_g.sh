#!/bin/bash
O=gpurun_out
python -m pytest tests/test_kernels_gpu.py -x -q -k "stem" 2>&1 | tail -4 > $O/r3_g_pytest.log; cat $O/r3_g_pytest.log
{ NOPESAC_STEM_SCALAR_LOADS=1 python scripts/stem_one.py; python scripts/stem_one.py; } 2>&1 | grep -v amdgpu > $O/r3_g_stem.log; cat $O/r3_g_stem.log
bash scripts/pmc_one.sh stem_fused scripts/../scripts/stem_one.py > $O/r3_g_pmc_stem.log 2>&1
bash scripts/pmc_one.sh conv3x3_c64 c64_one.py > $O/r3_g_pmc_c64.log 2>&1
grep -E "BANK|IDX_ACTIVE|us " $O/r3_g_pmc_stem.log $O/r3_g_pmc_c64.log
F="--no-cpu-baseline --no-boundary --no-fp32-path --no-accuracy --no-other-configs"
python bench.py $F > $O/r3_g_bench.json 2> $O/r3_g_bench.err
python bench.py $F --single-stream > $O/r3_g_bench_single.json 2>> $O/r3_g_bench.err
for f in bench bench_single; do python -c "
import json; d=json.load(open('$O/r3_g_$f.json')); c=d['config']; print('$f', d['value'], d['ms_per_step'], 'host', c['host_launch_ms_per_step'])"; done
python scripts/small_batch_latency.py 2>&1 | grep -v amdgpu | tee $O/r3_g_latency.log
