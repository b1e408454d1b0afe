# Stem / encoder-tail kernels after an LDS-layout change: parity tests, stand-alone timings, PMC bank-conflict rates.
O=gpurun_out
timeout 900 python -m pytest tests -q -x -m gpu -k "stem or encoder_tail or decoder_tail or transformer_tail or plane_head or backbone" 2>&1 | tail -3
for i in 1 2; do python scripts/stem_one.py; python scripts/enc_tail_one.py; done
bash scripts/pmc_summary.sh $O/r4b_pmc_stem.json stem_fused stem_one.py > /dev/null 2>&1
bash scripts/pmc_summary.sh $O/r4b_pmc_enc_tail.json enc_tail enc_tail_one.py > /dev/null 2>&1
python - <<'P'
import json
for f in ('r4b_pmc_stem','r4b_pmc_enc_tail'):
    d=json.load(open('gpurun_out/%s.json'%f))
    for k,v in d['kernels'].items():
        c=v['counters']; print(f,k,'bank conflict %.3f'%(c['SQ_LDS_BANK_CONFLICT']/c['SQ_LDS_IDX_ACTIVE']),'mfma busy',v.get('mfma_busy_frac_of_simd_cycles'),'valu/mfma %.1f'%(c['SQ_INSTS_VALU']/c['SQ_INSTS_MFMA']))
P
