#!/bin/bash
# Round-4 evidence run on the final tree: GPU suite, kernel routing re-tuned, default bench line, isolated kernel costs, rocprofv3 kernel
# stats of the default command, HBM traffic, PMC summaries of the dominant kernel and of the kernels the round-3 verdict named.
O=gpurun_out; T=${1:-z}
if [ "${2:-}" != "skip-tests" ]; then python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee $O/r4_${T}_pytest_gpu.log; fi
F="--no-cpu-baseline --no-boundary --no-fp32-path --no-accuracy --no-other-configs --no-tape"
python bench.py $F --retune --routing $O/routing_r4.json > $O/r4_${T}_bench_retune.json 2> $O/r4_${T}_bench.err
python bench.py $F --k 64 --config scannet --steps 8 --warmup 3 --retune --routing $O/routing_r4_scannet_k64.json > /dev/null 2>> $O/r4_${T}_bench.err
python bench.py $F --k 128 --fp8 --steps 8 --warmup 3 --retune --routing $O/routing_r4_fp8_k128.json > /dev/null 2>> $O/r4_${T}_bench.err
cp $O/routing_r4*.json profiles/          # (the box's copy of profiles/: the default bench line below loads them)
python bench.py --gpus 1 --steps 20 --warmup 5 --layers $O/r4_${T}_gemm_layers.tsv > $O/r4_${T}_bench.json 2>> $O/r4_${T}_bench.err
python - <<PY
import json
d=json.load(open('$O/r4_${T}_bench.json')); r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], r['by_bound']['mfma_bound_layers'], r['by_bound']['hbm_bound_layers'], r['engine_clock']['sclk_mhz_under_benchmark_load'])
print('tape', d.get('launch_tape',{}).get('value')); print('pose', {k:(v['R_err_deg_mean'], v['R_err_deg_max']) for k,v in d['pose_err_vs_fp32_path']['bench_workload'].items() if isinstance(v, dict)})
b=d.get('boundary',{}); print('boundary', b.get('value'), 'one pair', b.get('one_pair_per_call'))
print('other', {k:(v.get('value'), v.get('ms_per_step'), v.get('error')) for k,v in d.get('other_configs',{}).items()})
print('cpu', d.get('cpu_baseline'), d.get('fp32_parity_path',{}).get('value'))
PY
bash scripts/prof_isolated.sh r4${T} --routing $O/routing_r4.json > $O/r4_${T}_isolated.log 2>&1; cp $O/iso_r4${T}.tsv $O/r4_${T}_isolated_kernel_costs.txt; head -3 $O/r4_${T}_isolated.log | cut -c1-200
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_r4${T}_stats -o bench -- python $R/bench.py --steps 32 --warmup 5 --no-cpu-baseline --no-accuracy --no-boundary --no-other-configs --no-tape --no-fp32-path --routing $R/$O/routing_r4.json > $R/$O/r4_${T}_stats.log 2>&1
cd $R; cp $(find $O/prof_r4${T}_stats -name "*kernel_stats.csv" | head -1) $O/r4_${T}_kernel_stats.csv; head -4 $O/r4_${T}_kernel_stats.csv | cut -c1-200
bash scripts/pmc_bench.sh --routing $O/routing_r4.json > $O/r4_${T}_pmc_bench.log 2>&1; cp $O/pmc_traffic.json $O/r4_${T}_pmc_traffic.json; tail -3 $O/r4_${T}_pmc_bench.log
bash scripts/pmc_summary.sh $O/r4_pmc_conv_p8.json conv_igemm_p8 conv_one.py 64 60 80 256 256 3 1 p832 > /dev/null 2>&1
bash scripts/pmc_summary.sh $O/r4_pmc_conv_p8_res4_3x3.json conv_igemm_p8 conv_one.py 64 30 40 256 256 3 1 p832 > /dev/null 2>&1
bash scripts/pmc_summary.sh $O/r4_pmc_conv_p8_res4_expand.json conv_igemm_p8 conv_one.py 64 30 40 256 1024 1 1 p832 res > /dev/null 2>&1
bash scripts/pmc_summary.sh $O/r4_pmc_conv_p8_res4_reduce.json conv_igemm_p8 conv_one.py 64 30 40 1024 256 1 1 p832 > /dev/null 2>&1
bash scripts/pmc_summary.sh $O/r4_pmc_res3_tail.json pw_chain tail_one.py 64 60 80 128 512 128 > /dev/null 2>&1
bash scripts/pmc_summary.sh $O/r4_pmc_res2_tail.json pw_chain tail_one.py 64 120 160 64 256 64 > /dev/null 2>&1
bash scripts/pmc_summary.sh $O/r4_pmc_stem.json stem_fused stem_one.py > /dev/null 2>&1
bash scripts/pmc_summary.sh $O/r4_pmc_enc_tail.json enc_tail enc_tail_one.py > /dev/null 2>&1
bash scripts/pmc_summary.sh $O/r4_pmc_sinkhorn.json matcher_sinkhorn sinkhorn_one.py > /dev/null 2>&1
python - <<PY
import json
for f in ('r4_pmc_conv_p8','r4_pmc_conv_p8_res4_3x3','r4_pmc_conv_p8_res4_expand','r4_pmc_conv_p8_res4_reduce','r4_pmc_res3_tail','r4_pmc_res2_tail','r4_pmc_stem','r4_pmc_enc_tail','r4_pmc_sinkhorn'):
    try:
        d=json.load(open('$O/'+f+'.json'))
        for k,v in d['kernels'].items(): print(f, k[:40], d['unprofiled_run'], {a:b for a,b in v.items() if a not in ('counters','wave_cycle_shares')})
    except Exception as e: print(f, 'failed', e)
PY
python scripts/parity_report.py > $O/r4_${T}_parity_report.json 2> $O/r4_${T}_parity.err; tail -c 600 $O/r4_${T}_parity_report.json
