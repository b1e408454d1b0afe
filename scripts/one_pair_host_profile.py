"""cProfile of model([pair]) in launch-tape mode (where does the HOST spend a call?).  usage: one_pair_host_profile.py [calls]"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nopesac_amd.synth import synth_pair  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 20
model = bench.build_model(torch.device("cuda:0"), 50, "bfloat16")
model.output_rle, model.use_hip_graph = True, True
inp = [synth_pair(3)]
for v in "01":
    inp[0][v]["image"] = inp[0][v]["image"].pin_memory()
with torch.no_grad():
    for _ in range(6):
        model(inp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(calls):
        model(inp)
    torch.cuda.synchronize()
    print("ms per call: %.2f" % (1e3 * (time.perf_counter() - t0) / calls), getattr(model, "tape_counts", None), getattr(model, "tape_error", None))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(calls):
        model(inp)
    torch.cuda.synchronize()
    pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
