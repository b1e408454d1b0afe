"""GPU box: where the time of the drop-in boundary goes (model(list[dict]) -> list[dict] with RLE instances) with several batches in
flight: host time to submit a batch, host time blocked waiting for the oldest batch's GPU work, host time of the packaging proper."""
import gc
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nopesac_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B, K = 32, 32
DEPTH = int(os.environ.get("DEPTH", "4"))
U8 = os.environ.get("U8", "1") == "1"
GRAPH = os.environ.get("GRAPH", "1") == "1"
model = bench.build_model(dev, 50, "bfloat16")
ops.TUNER.load(os.path.join(ROOT, "profiles", "routing_r5.json"))
forced = bench.make_forced(B, K, 50, dev, 7)
g = torch.Generator().manual_seed(0)
raw = torch.randint(0, 256, (2 * B, 3, 480, 640), generator=g)
host = (raw.to(torch.uint8) if U8 else raw.float()).pin_memory()
inputs = [{"0": {"image": host[i], "image_id": "a%d" % i, "file_name": ""}, "1": {"image": host[B + i], "image_id": "b%d" % i, "file_name": ""}}
          for i in range(B)]
model.output_rle = True
model.use_hip_graph = GRAPH
model.graph_slots = DEPTH
gc.collect(); gc.freeze()
streams = [torch.cuda.Stream() for _ in range(DEPTH)]
t_sub, t_wait, t_pack = [], [], []


def submit(slot):
    t0 = time.perf_counter()
    with torch.no_grad(), torch.cuda.stream(streams[slot]):
        model.infer_iter += 1
        d = model.forward_device(inputs, forced=forced)
        ev = torch.cuda.Event(); ev.record()
    t_sub.append(time.perf_counter() - t0)
    return slot, d, ev


def finish(h):
    t0 = time.perf_counter()
    h[2].synchronize()
    t1 = time.perf_counter()
    with torch.no_grad(), torch.cuda.stream(streams[h[0]]):
        r = model.package(inputs, h[1])
    t_wait.append(t1 - t0); t_pack.append(time.perf_counter() - t1)
    return r


def run(n):
    pending = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        pending.append(submit(i % DEPTH))
        if len(pending) >= DEPTH:
            finish(pending.pop(0))
    while pending:
        finish(pending.pop(0))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


run(3 * DEPTH)
t_sub.clear(); t_wait.clear(); t_pack.clear()
el = run(8 * DEPTH)
ms = lambda v: 1e3 * sum(v) / len(v)
print("depth %d u8 %s graph %s: %.2f ms/step = %.0f pairs/s | host per step: submit %.2f  wait-for-forward %.2f  package %.2f ms"
      % (DEPTH, U8, GRAPH, 1e3 * el, B / el, ms(t_sub), ms(t_wait), ms(t_pack)))
