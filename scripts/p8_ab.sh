#!/bin/bash
# p8 A/B on the MFMA-bound shapes: parity tests, time per launch (alternating), cycle stamps.  usage: bash scripts/p8_ab.sh <tag>
O=gpurun_out; T=${1:-x}
python -m pytest tests/test_kernels_gpu.py -x -q -k "p8" 2>&1 | tail -2
for rep in 1 2; do for shp in "64 60 80 256 256 3 1" "64 30 40 256 256 3 1" "64 15 20 512 512 3 1" "64 30 40 1024 256 1 1" "64 60 80 128 256 3 1"; do
  echo "$shp: $(python scripts/conv_one.py $shp p832 | tail -1)"; done; done 2>&1 | grep -v amdgpu | tee $O/p8_ab_$T.txt
python scripts/p8_stamps.py 64 60 80 256 256 3 1 2>&1 | grep -v amdgpu | head -4 | tee -a $O/p8_ab_$T.txt
