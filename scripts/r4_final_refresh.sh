O=gpurun_out; T=z
python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee $O/r4_${T}_pytest_gpu.log
python bench.py --gpus 1 --steps 20 --warmup 5 --layers $O/r4_${T}_gemm_layers.tsv > $O/r4_${T}_bench.json 2> $O/r4_${T}_bench.err
python - <<PY
import json
d=json.load(open('$O/r4_${T}_bench.json')); r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], r['by_bound']['mfma_bound_layers'], r['by_bound']['hbm_bound_layers'])
b=d.get('boundary',{}); print('boundary', b.get('value'), 'one pair', b.get('one_pair_per_call'))
print('tape', d.get('launch_tape',{}).get('value'), 'fp32', d.get('fp32_parity_path',{}).get('value'), 'cpu', d.get('cpu_baseline',{}).get('value'))
PY
bash scripts/prof_isolated.sh r4${T} --routing profiles/routing_r4.json > $O/r4_${T}_isolated.log 2>&1; cp $O/iso_r4${T}.tsv $O/r4_${T}_isolated_kernel_costs.txt; head -3 $O/r4_${T}_isolated.log | cut -c1-200
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_r4${T}_stats -o bench -- python $R/bench.py --steps 32 --warmup 5 --no-cpu-baseline --no-accuracy --no-boundary --no-other-configs --no-tape --no-fp32-path --routing $R/profiles/routing_r4.json > $R/$O/r4_${T}_stats.log 2>&1
cd $R; cp $(find $O/prof_r4${T}_stats -name "*kernel_stats.csv" | head -1) $O/r4_${T}_kernel_stats.csv; head -4 $O/r4_${T}_kernel_stats.csv | cut -c1-200
bash scripts/pmc_bench.sh --routing profiles/routing_r4.json > $O/r4_${T}_pmc_bench.log 2>&1; cp $O/pmc_traffic.json $O/r4_${T}_pmc_traffic.json; tail -3 $O/r4_${T}_pmc_bench.log
