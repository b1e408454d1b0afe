"""GPU box: where the host time of the drop-in boundary goes (model(list[dict]) -> list[dict] with RLE instances): cProfile of a few
serial boundary steps on the benchmark workload."""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nopesac_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B, K = 32, 32
model = bench.build_model(dev, 50, "bfloat16")
ops.TUNER.load(os.path.join(ROOT, "profiles", "routing_r2.json"))
forced = bench.make_forced(B, K, 50, dev, 7)
g = torch.Generator().manual_seed(0)
raw = torch.randint(0, 256, (2 * B, 3, 480, 640), generator=g).float()
host = raw.pin_memory()
inputs = [{"0": {"image": host[i], "image_id": "a%d" % i, "file_name": ""}, "1": {"image": host[B + i], "image_id": "b%d" % i, "file_name": ""}}
          for i in range(B)]
model.output_rle = True


def one(times):
    with torch.no_grad():
        t0 = time.perf_counter()
        imgs = model.stack_images(inputs)
        t1 = time.perf_counter()
        d = model.forward_tensors(None, B, 480, 640, forced=forced, raw_images=imgs)
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        res = model.package(inputs, d)
        t4 = time.perf_counter()
    times.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
    return res


tm = []
for _ in range(3):
    one(tm)
tm = []
pr = cProfile.Profile()
pr.enable()
for _ in range(4):
    one(tm)
pr.disable()
print("ms per step: stack_images(host) %.2f  forward enqueue %.2f  wait for GPU %.2f  package %.2f" % tuple(1e3 * sum(t[i] for t in tm) / len(tm) for i in range(4)))
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
