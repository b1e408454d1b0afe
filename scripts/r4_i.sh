#!/bin/bash
# bfrag channel-major K order: parity, A/B on the res3 3x3 shapes (time + HBM bytes)
O=gpurun_out
python -m pytest tests/test_kernels_gpu.py -x -q -k "bfrag" 2>&1 | tail -2 | tee $O/r4_i_pytest.log
for shp in "64 60 80 128 128 3 1" "64 120 160 128 128 3 2"; do
  for m in bfrag32 bfrag288 bfrag3 bfrag259; do echo "$shp $m: $(python scripts/conv_one.py $shp $m | tail -1)"; done
done 2>&1 | grep -v amdgpu.ids | tee $O/r4_i_ab.txt
bash scripts/pmc_summary.sh $O/r4_pmc_bfrag_res3_tapmajor.json conv_igemm_bfrag conv_one.py 64 60 80 128 128 3 1 bfrag32 > /dev/null 2>&1
bash scripts/pmc_summary.sh $O/r4_pmc_bfrag_res3_chmajor.json conv_igemm_bfrag conv_one.py 64 60 80 128 128 3 1 bfrag288 > /dev/null 2>&1
python -c "
import json
for f in ('r4_pmc_bfrag_res3_tapmajor','r4_pmc_bfrag_res3_chmajor'):
    d=json.load(open('$O/'+f+'.json'))
    for k,v in d['kernels'].items(): print(f, d['unprofiled_run'], {a:b for a,b in v.items() if a!='counters'})"
