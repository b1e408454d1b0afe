#!/bin/bash
# does pacing the submissions help the launch tape (batches bunching up in the same phase)?
for pace in 0 2 4 6 8; do
NOPESAC_PACE_MS=$pace python bench.py --steps 40 --warmup 8 --no-other-configs --no-cpu-baseline --no-accuracy --no-fp32-path --no-boundary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pace $pace ms | eager', d['value'], 'tape', (d.get('launch_tape') or {}).get('value'))"
done
