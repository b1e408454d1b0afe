#!/bin/bash
# Round-6 PMC summaries (counters only, separate passes) of the kernels this round changed, before / after where a switch exists.
O=gpurun_out; mkdir -p $O
NOPESAC_ENC_TAIL_64=1 bash scripts/pmc_summary.sh $O/r6_pmc_enc_tail64_before.json enc_tail enc_tail_proj_one.py > /dev/null 2>&1
bash scripts/pmc_summary.sh $O/r6_pmc_enc_tail96_after.json enc_tail enc_tail_proj_one.py > /dev/null 2>&1
NOPESAC_ENC_TAIL_ROWS=4 bash scripts/pmc_summary.sh $O/r6_pmc_enc_tail128_after.json enc_tail enc_tail_proj_one.py > /dev/null 2>&1
bash scripts/pmc_summary.sh $O/r6_pmc_stem_after.json stem_fused stem_one.py > /dev/null 2>&1
bash scripts/pmc_summary.sh $O/r6_pmc_c64_after.json conv3x3_c64 c64_one.py > /dev/null 2>&1
NOPESAC_AB_LIBRARY=$PWD/scripts/ab/libnopesac_hip_r5.so bash scripts/pmc_summary.sh $O/r6_pmc_stem_before.json stem_fused stem_one.py > /dev/null 2>&1
NOPESAC_AB_LIBRARY=$PWD/scripts/ab/libnopesac_hip_r5.so bash scripts/pmc_summary.sh $O/r6_pmc_c64_before.json conv3x3_c64 c64_one.py > /dev/null 2>&1
python - <<PY
import json
for f in ("enc_tail64_before","enc_tail96_after","enc_tail128_after","stem_before","stem_after","c64_before","c64_after"):
    try:
        d=json.load(open("$O/r6_pmc_"+f+".json"))
        for k,v in d["kernels"].items(): print(f, k[:40], d["unprofiled_run"], {a:b for a,b in v.items() if a not in ("counters",)})
    except Exception as e: print(f, "failed", e)
PY
