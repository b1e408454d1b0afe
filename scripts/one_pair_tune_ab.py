"""One pair per call (the reference's batch-1 harness): kernel routing tuned for the 32-pair shapes only (what bench.py did through
round 5's first half) vs the one-pair shapes tuned as well, without / with the stream-K form on offer."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nopesac_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
model = bench.build_model(dev, 50, "bfloat16")
ops.TUNER.load(os.path.join(ROOT, "profiles", "routing_r5.json"))
print("32-pair routing only      ", bench.one_pair_latency(model), flush=True)
n0 = len(ops.TUNER.log)
model.autotune(1)
print("+ one-pair shapes tuned   ", bench.one_pair_latency(model), "shapes measured", len(ops.TUNER.log) - n0, flush=True)
from collections import Counter
print("   choices", Counter(c for _, c, _ in ops.TUNER.log[n0:]))
ops.P8_SK_TUNABLE[0] = True
keys = [k for k, _, _ in ops.TUNER.log[n0:]]
for k in keys:
    ops.TUNER.best.pop(k, None)
n1 = len(ops.TUNER.log)
model.autotune(1)
print("+ stream-K on offer       ", bench.one_pair_latency(model), flush=True)
print("   choices", Counter(c for _, c, _ in ops.TUNER.log[n1:]))
