"""GPU box: the drop-in boundary at SMALL batches, strictly serial (the reference's harness calls the model with ONE pair at a time):
model(list[dict]) -> list[dict] with host float32 images, eager launches vs MODEL.AMD.USE_HIP_GRAPH."""
import gc
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nopesac_amd.synth import synth_pair  # noqa: E402

dev = torch.device("cuda:0")
model = bench.build_model(dev, 50, "bfloat16")
model.output_rle = True
gc.collect(); gc.freeze()
for B in (1, 2, 4, 8, 32):
    inputs = [synth_pair(i) for i in range(B)]
    for p in inputs:
        for v in "01":
            p[v]["image"] = p[v]["image"].pin_memory()
    line = "B=%2d" % B
    for mode in ("eager", "hip_graph"):
        model.use_hip_graph = mode == "hip_graph"
        model._graphs = {}
        model.infer_iter = 0
        with torch.no_grad():
            for _ in range(6):
                model(inputs)
            torch.cuda.synchronize()
            n = max(4, 64 // B)
            t0 = time.perf_counter()
            for _ in range(n):
                model(inputs)
            torch.cuda.synchronize()
            el = (time.perf_counter() - t0) / n
        line += "   %s %7.2f ms/call = %6.0f pairs/s" % (mode, 1e3 * el, B / el)
    print(line, flush=True)
