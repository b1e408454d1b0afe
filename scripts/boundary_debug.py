"""Where the drop-in boundary loses against the resident-input loop: bench.boundary_rate with its per-step host split (submit / finish)
printed, eager vs launch tape with 4 / 2 / 1 streams per tape.  usage: boundary_debug.py"""
import json
import os
import sys

import torch

os.environ["NOPESAC_BD_DEBUG"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nopesac_amd import ops  # noqa: E402

B = 32
dev = torch.device("cuda:0")
model = bench.build_model(dev, 50, "bfloat16")
routing = os.path.join(ROOT, "profiles", "routing_r5.json")
if os.path.exists(routing):
    ops.TUNER.load(routing)
raw = torch.randint(0, 256, (2 * B, 3, 480, 640)).float().to(dev)
forced = bench.make_forced(B, 32, 50, dev, 7)
streams = [torch.cuda.Stream() for _ in range(4)]
for ts in (4, 2, 1):
    model.tape_streams = ts
    print("=== tape streams", ts, file=sys.stderr, flush=True)
    r = bench.boundary_rate(model, raw, forced, B, streams=streams)
    print(ts, json.dumps({d: {m: r[d][m]["four_in_flight"]["value"] for m in r[d]} for d in ("float32_images", "uint8_images")}),
          getattr(model, "tape_counts", None), flush=True)
