#!/bin/bash
for q in 4 5; do for sh in 0 1 2 3; do
  GPU_MAX_HW_QUEUES=$q python scripts/queue_map.py 0 0 shift$sh 2>&1 | grep "^stream set\|^pad" | sed "s/^/Q=$q /"
done; done
