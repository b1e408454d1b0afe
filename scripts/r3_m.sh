#!/bin/bash
# hardware queues: does giving the null stream's queue a sibling (5 queues) keep the 4 batch streams apart?
R=$PWD
for q in 4 5 6 8; do
  GPU_MAX_HW_QUEUES=$q python scripts/queue_map.py 0 0 2>&1 | tail -1 | sed "s/^/GPU_MAX_HW_QUEUES=$q  /"
done
export TMPDIR=/tmp; cd /tmp
for q in 4 5; do
  GPU_MAX_HW_QUEUES=$q rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_q$q -o q -- python $R/scripts/queue_map.py 0 0 > /dev/null 2>&1
  echo "--- GPU_MAX_HW_QUEUES=$q"; python $R/scripts/queue_ids.py $(ls $R/gpurun_out/prof_q$q/*/q_kernel_trace.csv $R/gpurun_out/prof_q$q/q_kernel_trace.csv 2>/dev/null | head -1)
done
