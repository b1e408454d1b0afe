#!/bin/bash
O=gpurun_out
for P in 32 27 54; do
python bench.py --pairs $P --no-cpu-baseline --no-boundary --no-fp32-path --no-accuracy --no-other-configs --routing $O/routing_tmp_$P.json > $O/r3_d_bench_p$P.json 2>> $O/r3_d_bench.err
python -c "
import json; d=json.load(open('$O/r3_d_bench_p$P.json')); print('pairs $P:', d['value'], 'pairs/s', d['ms_per_step'], 'ms/step', d['roofline']['by_bound'])"
done
