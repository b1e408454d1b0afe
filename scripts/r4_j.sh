#!/bin/bash
# headline A/B: bfrag channel-major K order on / off (alternating runs, one box)
O=gpurun_out
F="--no-cpu-baseline --no-boundary --no-fp32-path --no-accuracy --no-other-configs --no-tape --routing $O/routing_r4.json --steps 40"
for rep in 1 2 3; do
  for km in 1 0; do
    NOPESAC_BFRAG_KMAJOR=$km python bench.py $F 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kmajor $km', d['value'], d['ms_per_step'])"
  done
done | tee $O/r4_j_bfrag_kmajor_ab.txt
