"""Where does the host spend its time while it submits one 32-pair forward?  cProfile over N forwards of the benchmark configuration
(bf16, K forced), sorted by own time.  usage: python scripts/host_profile.py [--pairs 32] [--n 20] [--top 45]"""
import argparse
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

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=32)
ap.add_argument("--n", type=int, default=20)
ap.add_argument("--top", type=int, default=45)
args = ap.parse_args()
dev = torch.device("cuda:0")
B, K, nq = args.pairs, 32, 50
model = bench.build_model(dev, nq, "bfloat16")
routing = os.path.join(ROOT, "profiles", "routing_r5.json")
if os.path.exists(routing):
    ops.TUNER.load(routing)
raw = torch.randint(0, 256, (2 * B, 3, 480, 640)).float().to(dev)
forced = bench.make_forced(B, K, nq, dev, 7)


def one():
    with torch.no_grad():
        model.forward_tensors(None, B, 480, 640, forced=forced, raw_images=raw)


for _ in range(3):
    one()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.n):
    one()
t_submit = (time.perf_counter() - t0) / args.n
torch.cuda.synchronize()
print("submit time per forward (no profiler): %.2f ms" % (1e3 * t_submit))
pr = cProfile.Profile()
pr.enable()
for _ in range(args.n):
    one()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(args.top)
