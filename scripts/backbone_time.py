"""Single-stream time of the bf16 ResNet-50 of the benchmark batch (64 images, routing file loaded), for A/B runs of the tail kernels
under NOPESAC_TAIL_NO_RT4 / NOPESAC_RES3_EDGES_FUSED / NOPESAC_TAIL_RT4_LATE (read once per process).  usage: backbone_time.py [pairs]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nopesac_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
model = bench.build_model(dev, 50, "bfloat16")
routing = os.path.join(ROOT, "profiles", "routing_r4.json")
if os.path.exists(routing):
    ops.TUNER.load(routing)
raw = torch.randint(0, 256, (2 * B, 3, 480, 640)).float().to(dev)
bb = model.backbone


def one():
    with torch.no_grad():
        return bb(None, raw=(raw, model.pixel_mean, model.pixel_std))


for _ in range(3):
    one()
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        one()
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 10)
env = {k: v for k, v in os.environ.items() if k.startswith("NOPESAC_")}
print("backbone %d images: %.3f ms  env=%s" % (2 * B, best, env))
