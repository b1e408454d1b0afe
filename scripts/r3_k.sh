#!/bin/bash
# stream creation order / priority vs the boundary rate
for tape in "" tape; do
for prio in 0 -1; do
for pad in 0 1 2 3; do
python scripts/queue_map.py $pad $prio $tape 2>&1 | grep "^pad"
done; done; done
