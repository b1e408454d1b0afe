#!/bin/bash
# batches in flight: 4 (default) vs 5 / 6 / 8 on the current tree
O=gpurun_out
F="--no-cpu-baseline --no-boundary --no-fp32-path --no-accuracy --no-other-configs --no-tape --routing $O/routing_r4.json --steps 40"
for n in 4 6 8 4 5; do
  python bench.py $F --inflight $n 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inflight $n', d['value'], d['ms_per_step'], d['config']['streams'])"
done | tee $O/r4_h_inflight.txt
