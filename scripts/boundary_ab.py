"""A/B of the drop-in boundary rate (bench.boundary_rate: host images in, result dicts out, 4 batches in flight) under the
submit modes: eager with / without the pose-net side stream, launch tape, whole-graph replay.  usage: boundary_ab.py [pairs]"""
import json
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
routing = os.path.join(ROOT, "profiles", "routing_r3.json")
if os.path.exists(routing):
    ops.TUNER.load(routing)
raw = torch.randint(0, 256, (2 * B, 3, 480, 640)).float().to(dev)
forced = bench.make_forced(B, 32, 50, dev, 7)
streams = [torch.cuda.Stream() for _ in range(4)]
out = {}
for name, two, replay in (("two_streams", True, "launches"), ("one_stream", False, "launches"), ("whole_graph", True, "graph")):
    model.two_streams, model.graph_replay = two, replay
    r = bench.boundary_rate(model, raw, forced, B, streams=streams)["float32_images"]
    out[name] = {m: {k: v["value"] for k, v in r[m].items()} for m in r}
    print(name, json.dumps(out[name]), getattr(model, "tape_counts", None), flush=True)
