#!/bin/bash
O=gpurun_out
for rep in 1 2; do
echo "2-stage: $(python scripts/conv_one.py 64 15 20 2048 128 3 1 auto | tail -1)"
echo "3-stage: $(NOPESAC_GLDS_NARROW3=1 python scripts/conv_one.py 64 15 20 2048 128 3 1 auto | tail -1)"
done 2>&1 | grep -v amdgpu.ids | tee $O/r4_n.txt
