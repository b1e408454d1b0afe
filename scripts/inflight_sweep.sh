#!/bin/bash
# the default bench line at 3 / 4 / 5 / 6 batches in flight (A B C D, twice)
O=gpurun_out; mkdir -p $O
F="--steps 40 --warmup 8 --no-cpu-baseline --no-boundary --no-fp32-path --no-accuracy --no-other-configs --no-tape"
for rep in 1 2; do for n in 4 3 5 6; do
  python bench.py $F --inflight $n > $O/inflight_${n}_$rep.json 2>> $O/inflight.err
  python - <<PY
import json
d=json.load(open('$O/inflight_${n}_$rep.json')); print('inflight $n rep $rep', d['value'], d['ms_per_step'])
PY
done; done
