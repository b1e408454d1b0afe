#!/bin/bash
O=gpurun_out
F="--no-cpu-baseline --no-boundary --no-fp32-path --no-accuracy --no-other-configs --no-tape --routing $O/routing_r4.json --steps 40"
for cap in 0 248 240 224 0 192; do
NOPESAC_P8_GRID_CAP=$cap python bench.py $F 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('p8 grid cap $cap', d['value'], d['ms_per_step'])"
done | tee $O/r4_u_p8_grid_cap.txt
