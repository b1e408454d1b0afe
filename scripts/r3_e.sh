#!/bin/bash
O=gpurun_out
python -m pytest tests/test_e2e_gpu.py -x -q -k "hip_graph_mode" 2>&1 | tail -15 > $O/r3_e_pytest.log
cat $O/r3_e_pytest.log
F="--no-cpu-baseline --no-boundary --no-fp32-path --no-accuracy --no-other-configs"
python bench.py $F > $O/r3_e_bench_eager.json 2> $O/r3_e_bench.err
python bench.py $F --graph > $O/r3_e_bench_tape.json 2>> $O/r3_e_bench.err
python bench.py $F --graph --whole-graph > $O/r3_e_bench_graph.json 2>> $O/r3_e_bench.err
for f in eager tape graph; do python -c "
import json; d=json.load(open('$O/r3_e_bench_$f.json')); c=d['config']; print('$f', d['value'], d['ms_per_step'], 'host', c['host_launch_ms_per_step'], c.get('replay'), c.get('tape_nodes'))"; done
(time python bench.py) > $O/r3_e_bench_full.json 2> $O/r3_e_bench_full.err
python -c "
import json; d=json.load(open('$O/r3_e_bench_full.json')); print('full', d['value'], d['ms_per_step']); print(json.dumps(d.get('other_configs'))[:1500]); b=d['boundary']; print('boundary', b['value'], b['configuration']); print(json.dumps(b['float32_images'])); print(json.dumps(d.get('pose_err_vs_fp32_path',{}).get('bench_workload')))"
tail -5 $O/r3_e_bench_full.err; tail -3 $O/r3_e_bench.err
