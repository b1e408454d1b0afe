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
use_tape = "tape" in sys.argv[3:]
mimic = [a for a in sys.argv[3:] if a.startswith("mimic")]
B = 32
dev = torch.device("cuda:0")
model = bench.build_model(dev, 50, "bfloat16")
ops.TUNER.load(os.path.join(ROOT, "profiles", "routing_r5.json"))
if "mimic_autotune" in mimic:
    model.autotune(B)
policy = [a for a in sys.argv[3:] if a.startswith("shift")]
if policy:
    from nopesac_amd.streams import StreamSet
    model.autotune(1)                                   # (library loaded, kernels resident before the probe)
    ss = StreamSet(4, dev, side_shift=int(policy[0][5:])).bind(model)
    streams = ss.mains
    print("stream set:", ss.describe(), flush=True)
else:
    streams = [torch.cuda.Stream() for _ in range(4)]
    dummies = [torch.cuda.Stream() for _ in range(pad)]
    sides = [torch.cuda.Stream(priority=prio) for _ in range(4)]
    model._side_stream = {s.cuda_stream: e for s, e in zip(streams, sides)}
raw = torch.randint(0, 256, (2 * B, 3, 480, 640)).float()
host = raw.pin_memory()
inputs = [{"0": {"image": host[i], "image_id": "a%d" % i, "file_name": ""}, "1": {"image": host[B + i], "image_id": "b%d" % i, "file_name": ""}}
          for i in range(B)]
forced = bench.make_forced(B, 32, 50, dev, 7)
heat = [int(a[4:]) for a in sys.argv[3:] if a.startswith("heat")]
if "mimic_resident" in mimic or heat:               # what bench.py's timed loop does first: forwards on resident inputs
    rawd = raw.to(dev)
    for i in range(heat[0] if heat else 12):
        with torch.no_grad(), torch.cuda.stream(streams[i % 4]):
            model.forward_tensors(None, B, 480, 640, forced=forced, raw_images=rawd)
    torch.cuda.synchronize()
model.output_rle = True
model.use_hip_graph = use_tape
model.graph_slots = 4
gc.collect(); gc.freeze()


T = {"submit": 0.0, "wait": 0.0, "enqueue_fetch": 0.0, "package": 0.0, "gpu_ms": 0.0}


def _timed(name, f):
    def w(*a, **k):
        t = time.perf_counter()
        r = f(*a, **k)
        T[name] += time.perf_counter() - t
        return r
    return w


model._enqueue_fetch = _timed("enqueue_fetch", model._enqueue_fetch)


def run(n, depth=4):
    for k in T:
        T[k] = 0.0

    def submit(slot):
        t = time.perf_counter()
        with torch.no_grad(), torch.cuda.stream(streams[slot]):
            model.infer_iter += 1
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
            d = model.forward_device(inputs, forced=forced)
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
        T["submit"] += time.perf_counter() - t
        return slot, d, ev, e0

    def finish(h):
        t = time.perf_counter()
        h[2].synchronize()
        T["wait"] += time.perf_counter() - t
        T["gpu_ms"] += h[3].elapsed_time(h[2])
        t = time.perf_counter()
        with torch.no_grad(), torch.cuda.stream(streams[h[0]]):
            r = model.package(inputs, h[1])
        T["package"] += time.perf_counter() - t
        return r
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


def mhz():
    t = ops.clock_probe(400000)
    torch.cuda.synchronize()
    t = t.tolist()
    return 100.0 * t[0] / max(t[1], 1)


model.infer_iter = 0
print("engine clock before the loop: %.0f MHz" % mhz(), flush=True)
run(12)
best = min(run(24) for _ in range(3))
print("   per step (last run): " + "  ".join("%s %.2f" % (k, (1e3 if k != "gpu_ms" else 1.0) * v / 24) for k, v in T.items()), flush=True)
print("engine clock after the loop: %.0f MHz" % mhz(), flush=True)
print("pad %d side_priority %d tape %d %s: %.2f ms/step = %.0f pairs/s  %s" % (pad, prio, use_tape, mimic, 1e3 * best, B / best, getattr(model, "tape_counts", "")), flush=True)
