#!/bin/bash
# Evidence run on the current tree: [GPU suite,] [kernel routing re-tuned,] default bench line, isolated kernel costs, rocprofv3 kernel
# stats of the default command, HBM traffic (PMC) over the SAME launch set.  usage: final_profiles.sh <round> <tag> [skip-tests] [retune]
RND=${1:-5}; T=${2:-z}; O=gpurun_out; R=$PWD; P=r${RND}_${T}
mkdir -p $O
ROUT=$R/profiles/routing_r5.json        # (round 6 keeps the round-5 tuning pass; one entry moved by hand: the pose net's first conv on the split-K form, configuration 15)
case " $* " in *" skip-tests "*) ;; *) python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee $O/${P}_pytest_gpu.log;; esac
F="--no-cpu-baseline --no-boundary --no-fp32-path --no-accuracy --no-other-configs --no-tape"
case " $* " in *" retune "*)
  python bench.py $F --retune --routing $O/routing_r${RND}.json > $O/${P}_bench_retune.json 2> $O/${P}_bench.err
  python bench.py $F --k 64 --config scannet --steps 8 --warmup 3 --retune --routing $O/routing_r${RND}_scannet_k64.json > /dev/null 2>> $O/${P}_bench.err
  python bench.py $F --k 128 --fp8 --steps 8 --warmup 3 --retune --routing $O/routing_r${RND}_fp8_k128.json > /dev/null 2>> $O/${P}_bench.err
  cp $O/routing_r${RND}*.json profiles/;;          # (the box's copy of profiles/: everything below loads them; merged back via gpurun_out/)
esac
python bench.py --gpus 1 --steps 20 --warmup 5 --layers $O/${P}_gemm_layers.tsv > $O/${P}_bench.json 2>> $O/${P}_bench.err
python - <<PY
import json
d=json.load(open('$O/${P}_bench.json')); r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], r['by_bound']['mfma_bound_layers'], r['by_bound']['hbm_bound_layers'], (r.get('engine_clock') or {}).get('sclk_mhz_under_benchmark_load'))
print('tape', d.get('launch_tape',{}).get('value')); print('pose', {k:(v['R_err_deg_mean'], v['R_err_deg_max']) for k,v in d.get('pose_err_vs_fp32_path',{}).get('bench_workload',{}).items() if isinstance(v, dict)})
b=d.get('boundary',{}); print('boundary', b.get('value'), 'one pair', b.get('one_pair_per_call'), 'png', b.get('png_decode'))
print('other', {k:(v.get('value'), v.get('ms_per_step'), v.get('error')) for k,v in d.get('other_configs',{}).items()})
print('cpu', d.get('cpu_baseline'), d.get('fp32_parity_path',{}).get('value'))
PY
bash scripts/prof_isolated.sh ${P} --routing $ROUT > $O/${P}_isolated.log 2>&1; cp $O/iso_${P}.tsv $O/${P}_isolated_kernel_costs.txt; head -3 $O/${P}_isolated.log | cut -c1-200
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_${P}_stats -o bench -- python $R/bench.py --steps 32 --warmup 5 --no-cpu-baseline --no-accuracy --no-boundary --no-other-configs --no-tape --no-fp32-path --routing $ROUT > $R/$O/${P}_stats.log 2>&1
cd $R; cp $(find $O/prof_${P}_stats -name "*kernel_stats.csv" | head -1) $O/${P}_kernel_stats.csv; head -4 $O/${P}_kernel_stats.csv | cut -c1-200
bash scripts/pmc_bench.sh --routing $ROUT > $O/${P}_pmc_bench.log 2>&1; cp $O/pmc_traffic.json $O/${P}_pmc_traffic.json; tail -3 $O/${P}_pmc_bench.log
# PMC summaries of the dominant kernel and of this round's new kernels (counters only, separate passes)
bash scripts/pmc_summary.sh $O/r${RND}_pmc_conv_p8.json conv_igemm_p8 conv_one.py 64 60 80 256 256 3 1 p832 > /dev/null 2>&1
bash scripts/pmc_summary.sh $O/r${RND}_pmc_conv_p8_res4_3x3.json conv_igemm_p8 conv_one.py 64 30 40 256 256 3 1 p832 > /dev/null 2>&1

bash scripts/pmc_summary.sh $O/r${RND}_pmc_conv_p8n_res3_3x3.json conv_igemm_p8n conv_one.py 64 60 80 128 128 3 1 p8n0 > /dev/null 2>&1
bash scripts/pmc_summary.sh $O/r${RND}_pmc_conv_p8n_res3_s2.json conv_igemm_p8n conv_one.py 64 120 160 128 128 3 2 p8n32 > /dev/null 2>&1
bash scripts/pmc_summary.sh $O/r${RND}_pmc_conv_p8_res4_expand.json conv_igemm_p8 conv_one.py 64 30 40 256 1024 1 1 p832 res > /dev/null 2>&1
python - <<PY
import json
for f in ('conv_p8','conv_p8_res4_3x3','conv_p8n_res3_3x3','conv_p8n_res3_s2','conv_p8_res4_expand'):
    try:
        d=json.load(open('$O/r${RND}_pmc_'+f+'.json'))
        for k,v in d['kernels'].items(): print(f, k[:44], d['unprofiled_run'], {a:b for a,b in v.items() if a not in ('counters','wave_cycle_shares')})
    except Exception as e: print(f, 'failed', e)
PY
python scripts/parity_report.py > $O/${P}_parity_report.json 2> $O/${P}_parity.err; tail -c 400 $O/${P}_parity_report.json
