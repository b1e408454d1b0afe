"""How the order in which a process creates its HIP streams (= their hardware queues, handed out round robin) moves the drop-in
boundary rate: 4 batch streams, `pad` dummy streams, then the 4 pose-net side streams (optionally high priority), registered with the
model before its first call.  usage: queue_map.py <pad> <side_priority 0|-1> [tape]"""
import gc
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nopesac_amd import ops  # noqa: E402

pad, prio = int(sys.argv[1]), int(sys.argv[2])
use_tape = len(sys.argv) > 3 and sys.argv[3] == "tape"
B = 32
dev = torch.device("cuda:0")
model = bench.build_model(dev, 50, "bfloat16")
ops.TUNER.load(os.path.join(ROOT, "profiles", "routing_r3.json"))
streams = [torch.cuda.Stream() for _ in range(4)]
dummies = [torch.cuda.Stream() for _ in range(pad)]
sides = [torch.cuda.Stream(priority=prio) for _ in range(4)]
model._side_stream = {s.cuda_stream: e for s, e in zip(streams, sides)}
raw = torch.randint(0, 256, (2 * B, 3, 480, 640)).float()
host = raw.pin_memory()
inputs = [{"0": {"image": host[i], "image_id": "a%d" % i, "file_name": ""}, "1": {"image": host[B + i], "image_id": "b%d" % i, "file_name": ""}}
          for i in range(B)]
forced = bench.make_forced(B, 32, 50, dev, 7)
model.output_rle = True
model.use_hip_graph = use_tape
model.graph_slots = 4
gc.collect(); gc.freeze()


def run(n, depth=4):
    def submit(slot):
        with torch.no_grad(), torch.cuda.stream(streams[slot]):
            model.infer_iter += 1
            d = model.forward_device(inputs, forced=forced)
            ev = torch.cuda.Event()
            ev.record()
            return slot, d, ev

    def finish(h):
        h[2].synchronize()
        with torch.no_grad(), torch.cuda.stream(streams[h[0]]):
            return model.package(inputs, h[1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pending = []
    for i in range(n):
        pending.append(submit(i % depth))
        if len(pending) >= depth:
            finish(pending.pop(0))
    while pending:
        finish(pending.pop(0))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


model.infer_iter = 0
run(12)
best = min(run(24) for _ in range(3))
print("pad %d side_priority %d tape %d: %.2f ms/step = %.0f pairs/s  %s" % (pad, prio, use_tape, 1e3 * best, B / best, getattr(model, "tape_counts", "")), flush=True)
