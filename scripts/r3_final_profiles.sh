#!/bin/bash
# Round-3 evidence run: new regression tests, kernel routing re-tuned on the final tree, isolated kernel costs, HBM traffic, PMC summaries.
O=gpurun_out
python -m pytest tests/test_e2e_gpu.py tests/test_kernels_gpu.py -x -q -k "in_flight or next_to_mfma or hip_graph_mode or force_k or bottleneck_tail" 2>&1 | tail -4 | tee $O/r3_p_pytest.log
F="--no-cpu-baseline --no-boundary --no-fp32-path --no-accuracy --no-other-configs --no-tape"
python bench.py $F --retune --routing $O/routing_r3.json > $O/r3_p_bench_retune.json 2> $O/r3_p_bench.err
python bench.py $F --routing $O/routing_r3.json --layers $O/r3_p_gemm_layers.tsv > $O/r3_p_bench.json 2>> $O/r3_p_bench.err
python -c "
import json
for f in ('r3_p_bench_retune','r3_p_bench'):
    d=json.load(open('$O/'+f+'.json')); print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['by_bound']['mfma_bound_layers']['TFLOP/s'], d['roofline']['by_bound']['hbm_bound_layers']['TB/s_algorithmic'])"
bash scripts/prof_isolated.sh r3p --routing $O/routing_r3.json --no-boundary --no-other-configs --no-tape --no-fp32-path > $O/r3_p_isolated.log 2>&1
tail -48 $O/r3_p_isolated.log | head -50
bash scripts/pmc_bench.sh --routing $O/routing_r3.json > $O/r3_p_pmc_bench.log 2>&1; tail -5 $O/r3_p_pmc_bench.log
bash scripts/pmc_summary.sh $O/r3_pmc_res3_tail.json pw_chain tail_one.py 64 60 80 128 512 128 > /dev/null 2>&1
bash scripts/pmc_summary.sh $O/r3_pmc_res2_tail.json pw_chain tail_one.py 64 120 160 64 256 64 > /dev/null 2>&1
bash scripts/pmc_summary.sh $O/r3_pmc_res3_edge_cn256.json pw_chain tail_one.py 64 60 80 128 512 256 > /dev/null 2>&1
bash scripts/pmc_summary.sh $O/r3_pmc_c64.json conv3x3_c64 c64_one.py > /dev/null 2>&1
bash scripts/pmc_summary.sh $O/r3_pmc_stem.json stem_fused stem_one.py > /dev/null 2>&1
python -c "
import json
for f in ('r3_pmc_res3_tail','r3_pmc_res2_tail','r3_pmc_res3_edge_cn256','r3_pmc_c64','r3_pmc_stem'):
    d=json.load(open('$O/'+f+'.json'))
    for k,v in d['kernels'].items(): print(f, k[:50], d['unprofiled_run'], {a:b for a,b in v.items() if a!='counters'})"
