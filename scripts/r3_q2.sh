#!/bin/bash
python -m pytest tests -m gpu -x -q -k "graph or flight or runner or cli or bench_two" 2>&1 | tail -3
run() {
  python bench.py --steps 30 --warmup 6 --no-other-configs --no-cpu-baseline --no-accuracy --no-fp32-path "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d.get('boundary')
f=lambda r:{m:{k:r[m][k]['value'] for k in r[m]} for m in r}
print('$*', '| headline', d['value'], 'tape', (d.get('launch_tape') or {}).get('value'), '| boundary f32', json.dumps(f(b['float32_images'])) if b else None, 'u8', json.dumps(f(b['uint8_images'])) if b else None, b['one_pair_per_call'], d['config']['streams'])"
}
run --streams own
run --streams shift2
run --streams none
python scripts/runner_rate.py 2>&1 | grep "^runner"
python scripts/queue_map.py 0 0 shift0 tape 2>&1 | tail -1
