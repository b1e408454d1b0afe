#!/bin/bash
# A/B on one box: 64-token encoder tail vs the 32-token one (NOPESAC_ENC_TAIL_32=1), alternating runs
for i in 1 2 3; do
for v in 0 1; do
  if [ $v = 1 ]; then export NOPESAC_ENC_TAIL_32=1; else unset NOPESAC_ENC_TAIL_32; fi
  python bench.py --steps 40 --warmup 8 --no-other-configs --no-cpu-baseline --no-fp32-path --no-boundary --no-accuracy --no-tape 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('enc_tail_32=$v', d['value'], d['ms_per_step'])"
done; done
