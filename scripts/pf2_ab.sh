#!/bin/bash
# generic 64x64 / BK=128 conv kernel: two tiles of prefetch (PF2) vs one, same box.  usage: bash scripts/pf2_ab.sh <tag>
O=gpurun_out; T=${1:-x}
python -m pytest tests/test_kernels_gpu.py -x -q -k "conv2d or linear" 2>&1 | tail -2 | tee $O/pf2_ab_$T.txt
for rep in 1 2; do
  for flag in "-DNPS_NO_PF2" ""; do
    NOPESAC_HIPCC_EXTRA="$flag" python -m nopesac_amd.build >/dev/null 2>&1
    echo "build flags: '$flag'" | tee -a $O/pf2_ab_$T.txt
    python - <<'PY' 2>/dev/null | tail -1 | tee -a $O/pf2_ab_$T.txt
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
model = bench.build_model(torch.device("cuda:0"), 50, "bfloat16")
r = bench.one_pair_latency(model)
print({k: v["ms_per_call"] for k, v in r.items() if isinstance(v, dict)})
PY
  done
done
