#!/bin/bash
# Evidence of the final round-3 tree: per-kernel stats of the default benchmark command (4 batches in flight), isolated kernel costs,
# HBM traffic (PMC), the per-layer table.
O=gpurun_out; R=$PWD
F="--no-cpu-baseline --no-boundary --no-fp32-path --no-accuracy --no-other-configs --no-tape"
python bench.py $F --layers $O/r3_s_gemm_layers.tsv > $O/r3_s_bench_layers.json 2> $O/r3_s_bench_layers.err
bash scripts/prof_isolated.sh r3s > $O/r3_s_isolated.log 2>&1
tail -46 $O/r3_s_isolated.log | head -32
cp $O/iso_r3s.tsv $O/r3_s_isolated_kernel_costs.txt
bash scripts/pmc_bench.sh > $O/r3_s_pmc_bench.log 2>&1; tail -4 $O/r3_s_pmc_bench.log
cp $O/pmc_traffic.json $O/r3_s_pmc_traffic.json
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_r3s_stats -o bench -- python $R/bench.py --steps 20 --warmup 5 $F > $R/$O/r3_s_stats.log 2>&1
cd $R
f=$(ls $O/prof_r3s_stats/*/bench_kernel_stats.csv $O/prof_r3s_stats/bench_kernel_stats.csv 2>/dev/null | head -1)
head -25 $f | cut -c1-160
cp $f $O/r3_s_kernel_stats.csv
tail -1 $O/r3_s_stats.log | cut -c1-300
