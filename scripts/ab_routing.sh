#!/bin/bash
# headline loop on two routing files, interleaved: ab_routing.sh <tag> <routing A> <routing B> [reps]
O=gpurun_out; T=$1; RA=$2; RB=$3; N=${4:-2}
F="--no-cpu-baseline --no-boundary --no-fp32-path --no-accuracy --no-other-configs --no-tape --steps 40 --warmup 8"
for rep in $(seq 1 $N); do for r in A B; do
  RT=$([ $r = A ] && echo $RA || echo $RB)
  python bench.py $F --routing $RT > $O/${T}_$r$rep.json 2>> $O/${T}.err
  python - <<PY
import json
d=json.load(open('$O/${T}_$r$rep.json')); r=d['roofline']; cf=r['conv_family']
print('$r$rep', d['value'], 'pairs/s', d['ms_per_step'], 'ms | conv family', cf['ms'], 'ms | p8 frac', r['frac'], 'routing', d['config']['routing_file'], d['config']['routing_entries_measured_now'])
PY
done; done
