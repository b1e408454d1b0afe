#!/bin/bash
# LDS-DMA conv kernel: channel-major K order on stride-1 KxK layers - parity, pose-net layer_3 A/B (time + HBM bytes), headline A/B
O=gpurun_out
python -m pytest tests/test_kernels_gpu.py -x -q -k "conv2d" 2>&1 | tail -2 | tee $O/r4_m_pytest.log
for km in 1 0; do echo "layer_3 64 15 20 2048 128 3 1 kmajor=$km: $(NOPESAC_GLDS_KMAJOR=$km python scripts/conv_one.py 64 15 20 2048 128 3 1 auto | tail -1)"; done 2>&1 | grep -v amdgpu.ids | tee $O/r4_m_ab.txt
NOPESAC_GLDS_KMAJOR=0 bash scripts/pmc_summary.sh $O/r4_pmc_glds_layer3_tapmajor.json conv_igemm_glds conv_one.py 64 15 20 2048 128 3 1 auto > /dev/null 2>&1
bash scripts/pmc_summary.sh $O/r4_pmc_glds_layer3_chmajor.json conv_igemm_glds conv_one.py 64 15 20 2048 128 3 1 auto > /dev/null 2>&1
python -c "
import json
for f in ('r4_pmc_glds_layer3_tapmajor','r4_pmc_glds_layer3_chmajor'):
    d=json.load(open('$O/'+f+'.json'))
    for k,v in d['kernels'].items(): print(f, d['unprofiled_run'], {a:b for a,b in v.items() if a!='counters'})" | tee -a $O/r4_m_ab.txt
F="--no-cpu-baseline --no-boundary --no-fp32-path --no-accuracy --no-other-configs --no-tape --routing $O/routing_r4.json --steps 40"
for rep in 1 2; do for km in 1 0; do
  NOPESAC_GLDS_KMAJOR=$km python bench.py $F 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('glds kmajor $km', d['value'], d['ms_per_step'])"
done; done | tee -a $O/r4_m_ab.txt
