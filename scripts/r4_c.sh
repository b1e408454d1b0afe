#!/bin/bash
# split-rounds route (whole rounds on p8 + the remaining images as a second launch), EPI16 counted drain; grid-cap experiment
O=gpurun_out
python -m pytest tests/test_kernels_gpu.py -x -q -k "p8 or split" 2>&1 | tail -4 | tee $O/r4_c_pytest.log
for m in p832 p838432; do echo "64 30 40 256 256 3 1 $m: $(python scripts/conv_one.py 64 30 40 256 256 3 1 $m | tail -1)"; done 2>&1 | grep -v amdgpu.ids | tee $O/r4_c_cap.txt
F="--no-cpu-baseline --no-boundary --no-fp32-path --no-accuracy --no-other-configs --no-tape"
python bench.py $F --retune --routing $O/routing_r4.json > $O/r4_c_bench_retune.json 2> $O/r4_c_bench.err
python bench.py $F --routing $O/routing_r4.json --layers $O/r4_c_gemm_layers.tsv > $O/r4_c_bench.json 2>> $O/r4_c_bench.err
python -c "
import json
for f in ('r4_c_bench_retune','r4_c_bench'):
    d=json.load(open('$O/'+f+'.json')); r=d['roofline']; print(f, d['value'], d['ms_per_step'], r['frac'], r['by_bound']['mfma_bound_layers'], r['by_bound']['hbm_bound_layers'])
r=json.load(open('$O/routing_r4.json'))['routing']
print({k:v for k,v in r.items() if v==12})
"
grep -c . $O/r4_c_bench.err; tail -3 $O/r4_c_bench.err
