#!/bin/bash
# what makes bench.py's boundary figure faster than the same loop in a fresh process?
python bench.py --steps 8 --warmup 4 --no-tape --no-other-configs --no-cpu-baseline --no-accuracy --no-fp32-path 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['boundary']
print('bench minimal: headline', d['value'], 'boundary eager4', b['float32_images']['eager']['four_in_flight'], 'tape4', b['float32_images']['hip_graph']['four_in_flight'])"
python scripts/queue_map.py 0 0 2>&1 | grep "^pad"
python scripts/queue_map.py 0 0 mimic_autotune 2>&1 | grep "^pad"
python scripts/queue_map.py 0 0 mimic_resident 2>&1 | grep "^pad"
python scripts/queue_map.py 0 0 mimic_autotune mimic_resident 2>&1 | grep "^pad"
python scripts/queue_map.py 0 0 tape mimic_autotune mimic_resident 2>&1 | grep "^pad"
