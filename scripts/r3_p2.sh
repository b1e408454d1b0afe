#!/bin/bash
# headline loop + boundary by stream / hardware-queue policy
run() {
  python bench.py --steps 30 --warmup 6 --no-other-configs --no-cpu-baseline --no-accuracy --no-fp32-path "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d.get('boundary')
f=lambda r:{m:r[m]['four_in_flight']['value'] for m in r}
print('$*', '| headline', d['value'], 'tape', (d.get('launch_tape') or {}).get('value'), '| boundary f32', json.dumps(f(b['float32_images'])) if b else None, 'u8', json.dumps(f(b['uint8_images'])) if b else None, d['config']['streams'].get('side_stream_class'))"
}
run --streams none
run --streams own
run --streams own --single-stream
run --streams shift2
run --streams shift3
GPU_MAX_HW_QUEUES=5 run --streams own
GPU_MAX_HW_QUEUES=5 run --streams shift2
