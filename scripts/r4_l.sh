#!/bin/bash
# pose-net layer_3 (3x3, 2048 -> 128 at 15x20, K = 18432): routed kernel vs bfrag tap-major / channel-major; retune with the new bfrag
O=gpurun_out
for m in auto bfrag3 bfrag259 bfrag32 bfrag288; do echo "64 15 20 2048 128 3 1 $m: $(python scripts/conv_one.py 64 15 20 2048 128 3 1 $m | tail -1)"; done 2>&1 | grep -v amdgpu.ids | tee $O/r4_l_layer3.txt
for m in auto bfrag3 bfrag259 bfrag32 bfrag288; do echo "64 30 40 128 128 3 1 $m: $(python scripts/conv_one.py 64 30 40 128 128 3 1 $m | tail -1)"; done 2>&1 | grep -v amdgpu.ids | tee -a $O/r4_l_layer3.txt
for m in auto bfrag3 bfrag259 bfrag32 bfrag288; do echo "32 15 20 304 128 3 1 $m: $(python scripts/conv_one.py 32 15 20 320 128 3 1 $m | tail -1)"; done 2>&1 | grep -v amdgpu.ids | tee -a $O/r4_l_layer3.txt
F="--no-cpu-baseline --no-boundary --no-fp32-path --no-accuracy --no-other-configs --no-tape"
python bench.py $F --retune --routing $O/routing_r4.json > $O/r4_l_bench_retune.json 2> $O/r4_l_bench.err
python bench.py $F --routing $O/routing_r4.json --layers $O/r4_l_gemm_layers.tsv > $O/r4_l_bench.json 2>> $O/r4_l_bench.err
python -c "
import json
for f in ('r4_l_bench_retune','r4_l_bench'):
    d=json.load(open('$O/'+f+'.json')); r=d['roofline']; print(f, d['value'], d['ms_per_step'], r['frac'], r['by_bound']['mfma_bound_layers']['frac_of_mfma_peak'], r['by_bound']['hbm_bound_layers']['frac_of_hbm_peak'])"
grep "2048) w(128, 3, 3" $O/r4_l_gemm_layers.tsv
