#!/bin/bash
# one pair per call: launch-tape latency of model([pair]) under an environment variable A/B.  usage: bash scripts/one_pair_ab.sh VAR valA valB
V=$1; A=$2; B=$3
for rep in 1 2; do for val in $A $B; do
env $V=$val python - <<'PY' 2>/dev/null | tail -1
import os, sys, json, torch
sys.path.insert(0, os.getcwd())
import bench
dev = torch.device("cuda:0")
model = bench.build_model(dev, 50, "bfloat16")
r = bench.one_pair_latency(model)
print(os.environ.get(sys.argv[0], ""), {k: v["ms_per_call"] for k, v in r.items() if isinstance(v, dict)}, {k: v for k, v in os.environ.items() if k.startswith("NOPESAC_")})
PY
done; done
