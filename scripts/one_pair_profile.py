"""model([pair]) strictly serial through the launch tape, N calls - run under `rocprofv3 --kernel-trace --stats` to see where the
one-pair-per-call latency goes (kernel time vs gaps).  usage: one_pair_profile.py [calls] [tune]"""
import gc
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nopesac_amd import ops  # noqa: E402
from nopesac_amd.synth import synth_pair  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda:0")
model = bench.build_model(dev, 50, "bfloat16")
model.output_rle = True
if len(sys.argv) > 2 and sys.argv[2] == "tune":
    model.autotune(1)
    print("tuned shapes (non-default choices):", sum(1 for v in ops.TUNER.best.values() if v), file=sys.stderr)
gc.collect(); gc.freeze()
inp = [synth_pair(3)]
for v in "01":
    inp[0][v]["image"] = inp[0][v]["image"].pin_memory()
model.use_hip_graph = True
with torch.no_grad():
    for _ in range(6):
        model(inp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(calls):
        model(inp)
    torch.cuda.synchronize()
print("one pair per call: %.2f ms" % (1e3 * (time.perf_counter() - t0) / calls), getattr(model, "tape_counts", None), flush=True)
