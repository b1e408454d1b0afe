#!/bin/bash
# A B A B of the default bench line under two environments.  usage: bash scripts/ab_env.sh <tag> "<env A>" "<env B>" [reps] [bench args]
T=$1; A=$2; B=$3; N=${4:-2}; shift 4
O=gpurun_out; mkdir -p $O
F="--steps 40 --warmup 8 --no-cpu-baseline --no-boundary --no-fp32-path --no-accuracy --no-other-configs --no-tape"
for i in $(seq 1 $N); do
  for leg in A B; do
    if [ $leg = A ]; then E="$A"; else E="$B"; fi
    env $E python bench.py $F "$@" > $O/ab_${T}_${leg}$i.json 2>> $O/ab_${T}.err
    python - <<PY
import json
d=json.load(open('$O/ab_${T}_${leg}$i.json')); print('$leg$i', '[$E]', d['value'], d['ms_per_step'], (d['roofline'].get('engine_clock') or {}).get('sclk_mhz_under_benchmark_load'))
PY
  done
done
