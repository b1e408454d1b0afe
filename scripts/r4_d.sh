#!/bin/bash
# EPI 4 with two passes of residual rows in flight; cycle stamps of the current p8 tile (prologue / K loop / epilogue)
O=gpurun_out
python -m pytest tests/test_kernels_gpu.py -x -q -k "p8" 2>&1 | tail -2 | tee $O/r4_d_pytest.log
for shp in "64 30 40 256 1024 1 1" "64 15 20 512 2048 1 1" "64 60 80 128 512 1 1"; do
  for m in p832; do echo "$shp $m res: $(python scripts/conv_one.py $shp $m res | tail -1)"; done
done 2>&1 | grep -v amdgpu.ids | tee $O/r4_d_ab.txt
python scripts/p8_stamps.py 64 30 40 256 256 3 1 2>&1 | grep -v amdgpu.ids | tee $O/r4_d_stamps.txt
python scripts/p8_stamps.py 64 60 80 256 256 3 1 2>&1 | grep -v amdgpu.ids | tee -a $O/r4_d_stamps.txt
python scripts/p8_stamps.py 64 30 40 1024 256 1 1 2>&1 | grep -v amdgpu.ids | tee -a $O/r4_d_stamps.txt
