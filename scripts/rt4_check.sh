# Tail kernels after an LDS-layout change: parity tests, timings of the res2 / res3 shapes, PMC bank-conflict rates.
O=gpurun_out
timeout 600 python -m pytest tests -q -x -m gpu -k "bottleneck_tail or fused_tail or backbone" 2>&1 | tail -3
for i in 1 2; do
python scripts/tail_one.py 64 120 160 64 256 64
python scripts/tail_one.py 64 60 80 128 512 128
python scripts/tail_one.py 64 60 80 128 512 256
python scripts/tail_one.py 64 60 80 128 512 128 256 2
python scripts/tail_one.py 64 120 160 64 256 64 64 1
done
bash scripts/pmc_summary.sh $O/r4b_pmc_res3_edge_cn256.json pw_chain tail_one.py 64 60 80 128 512 256 > /dev/null 2>&1
bash scripts/pmc_summary.sh $O/r4b_pmc_res3_proj.json pw_chain tail_one.py 64 60 80 128 512 128 256 2 > /dev/null 2>&1
python - <<'P'
import json
for f in ('r4b_pmc_res3_edge_cn256','r4b_pmc_res3_proj'):
    d=json.load(open('gpurun_out/%s.json'%f))
    for k,v in d['kernels'].items():
        c=v['counters']; print(f,k,'bank conflict %.3f'%(c['SQ_LDS_BANK_CONFLICT']/c['SQ_LDS_IDX_ACTIVE']),'mfma busy',v.get('mfma_busy_frac_of_simd_cycles'),'valu/mfma %.1f'%(c['SQ_INSTS_VALU']/c['SQ_INSTS_MFMA']))
P
