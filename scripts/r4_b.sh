#!/bin/bash
# EPI 4 epilogue with the residual requested one pass ahead: parity tests, A/B against the generic epilogue build (old flow), bench
O=gpurun_out
python -m pytest tests/test_kernels_gpu.py -x -q -k "p8" 2>&1 | tail -4 | tee $O/r4_b_pytest.log
for shp in "64 30 40 256 1024 1 1" "64 15 20 512 2048 1 1" "64 60 80 128 512 1 1"; do
  for m in p80 p864 p832 p896; do echo "$shp $m res: $(python scripts/conv_one.py $shp $m res | tail -1)"; done
done 2>&1 | tee $O/r4_b_ab.txt
F="--no-cpu-baseline --no-boundary --no-fp32-path --no-accuracy --no-other-configs --no-tape"
python bench.py $F --layers $O/r4_b_gemm_layers.tsv > $O/r4_b_bench.json 2> $O/r4_b_bench.err
python -c "
import json
d=json.load(open('$O/r4_b_bench.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['by_bound']['mfma_bound_layers'], r['by_bound']['hbm_bound_layers'])"
bash scripts/pmc_summary.sh $O/r4_pmc_conv_p8_res4_expand_b.json conv_igemm_p8 conv_one.py 64 30 40 256 1024 1 1 p832 res > $O/r4_b_pmc.log 2>&1; tail -12 $O/r4_b_pmc.log
