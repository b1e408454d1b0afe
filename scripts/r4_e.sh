#!/bin/bash
# same-box A/B: EPI 4 with one vs two passes of residual rows in flight (rebuilds the library on the box)
O=gpurun_out
run() { for shp in "64 30 40 256 1024 1 1" "64 15 20 512 2048 1 1" "64 60 80 128 512 1 1"; do echo "$1 $shp: $(python scripts/conv_one.py $shp p832 res | tail -1)"; done; }
for rep in 1 2; do
  python -m nopesac_amd.build > /dev/null; run ahead2
  NOPESAC_HIPCC_EXTRA="-DP8_EPI4_AHEAD=1" python -m nopesac_amd.build > /dev/null; NOPESAC_HIPCC_EXTRA="-DP8_EPI4_AHEAD=1" run ahead1
done 2>&1 | grep -v amdgpu.ids | tee $O/r4_e_ab.txt
