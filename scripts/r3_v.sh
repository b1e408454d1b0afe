#!/bin/bash
# do the image copies run on the DMA engines or as copy kernels, and does it matter?  (boundary loop, eager / tape)
for sdma in 1 0; do for mode in "" tape; do
  HSA_ENABLE_SDMA=$sdma python scripts/queue_map.py 0 0 shift0 $mode 2>&1 | grep "^pad\|per step" | sed "s/^/HSA_ENABLE_SDMA=$sdma /"
done; done
