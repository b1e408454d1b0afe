#!/bin/bash
O=gpurun_out
python -m pytest tests/test_e2e_gpu.py -x -q -k "hip_graph_mode" 2>&1 | tail -15 > $O/r3_h_pytest.log; cat $O/r3_h_pytest.log
python -m pytest tests/test_kernels_gpu.py -x -q -k "stem" 2>&1 | tail -3
python scripts/small_batch_latency.py 2>&1 | grep -v amdgpu | tee $O/r3_h_latency.log
F="--no-cpu-baseline --no-boundary --no-fp32-path --no-accuracy --no-other-configs"
python bench.py $F > $O/r3_h_bench_eager.json 2> $O/r3_h_bench.err
python bench.py $F --graph > $O/r3_h_bench_tape.json 2>> $O/r3_h_bench.err
for f in eager tape; do python -c "
import json; d=json.load(open('$O/r3_h_bench_$f.json')); c=d['config']; print('$f', d['value'], d['ms_per_step'], 'host', c['host_launch_ms_per_step'], c.get('replay'), c.get('tape_nodes'))"; done
python scripts/boundary_ab.py 2>&1 | grep -v amdgpu.ids | tee $O/r3_h_boundary.log
tail -3 $O/r3_h_bench.err
