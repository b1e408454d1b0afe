#!/bin/bash
# headline loop + boundary under different hardware queue counts
for q in 4 5 6 8 12; do
  GPU_MAX_HW_QUEUES=$q python bench.py --steps 30 --warmup 6 --no-other-configs --no-cpu-baseline --no-accuracy --no-fp32-path 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['boundary']
f=lambda r:{m:{k:r[m][k]['value'] for k in r[m]} for m in r}
print('GPU_MAX_HW_QUEUES=$q headline', d['value'], 'tape', d['launch_tape']['value'], 'boundary f32', json.dumps(f(b['float32_images'])), 'u8', json.dumps(f(b['uint8_images'])), b['one_pair_per_call'])"
done
