#!/bin/bash
O=gpurun_out
python -m pytest tests/test_kernels_gpu.py tests/test_stages_gpu.py -x -q -k "c64 or backbone" 2>&1 | tail -3
for u in 0 1 0 1; do echo "conv12 unfused=$u: $(NOPESAC_RES2_CONV12_UNFUSED=$u python scripts/backbone_time.py 2>&1 | tail -1)"; done | tee $O/r4_w_conv12.txt
F="--no-cpu-baseline --no-boundary --no-fp32-path --no-accuracy --no-other-configs --no-tape --routing $O/routing_r4.json --steps 40"
for u in 0 1 0 1; do
NOPESAC_RES2_CONV12_UNFUSED=$u python bench.py $F 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench conv12 unfused=$u', d['value'], d['ms_per_step'])"
done | tee -a $O/r4_w_conv12.txt
