#!/bin/bash
# Round-4 baseline on this round's boxes: default bench line (short), PMC summary of the dominant kernel on its MFMA-bound shapes.
O=gpurun_out
F="--no-cpu-baseline --no-boundary --no-fp32-path --no-accuracy --no-other-configs --no-tape"
python bench.py $F --layers $O/r4_a_gemm_layers.tsv > $O/r4_a_bench.json 2> $O/r4_a_bench.err
python -c "
import json
d=json.load(open('$O/r4_a_bench.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['by_bound']['mfma_bound_layers'], r['by_bound']['hbm_bound_layers'], r['engine_clock'])"
bash scripts/pmc_summary.sh $O/r4_pmc_conv_p8_res4_3x3.json conv_igemm_p8 conv_one.py 64 30 40 256 256 3 1 p8 > $O/r4_a_pmc1.log 2>&1
bash scripts/pmc_summary.sh $O/r4_pmc_conv_p8_dec_3x3.json conv_igemm_p8 conv_one.py 64 60 80 256 256 3 1 p8 > $O/r4_a_pmc2.log 2>&1
bash scripts/pmc_summary.sh $O/r4_pmc_conv_p8_res4_expand.json conv_igemm_p8 conv_one.py 64 30 40 256 1024 1 1 p8 res > $O/r4_a_pmc3.log 2>&1
tail -30 $O/r4_a_pmc1.log $O/r4_a_pmc2.log $O/r4_a_pmc3.log
