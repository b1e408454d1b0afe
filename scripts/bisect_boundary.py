"""Why is bench.py's boundary figure faster than the same loop in a fresh process?  Runs bench.main() (minimal flags) with
boundary_rate wrapped: the original call (the timed loop's streams), then again with fresh streams, then again after dropping the
allocator cache.  usage: bisect_boundary.py [bench flags...]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

real = bench.boundary_rate


def four(r):
    return {d: {m: r[d][m]["four_in_flight"]["ms_per_step"] for m in r[d]} for d in ("float32_images", "uint8_images")}


def wrapped(model, raw, forced, B, steps=4, streams=None):
    r = real(model, raw, forced, B, steps=steps, streams=streams)
    print("boundary, the timed loop's streams:", json.dumps(four(r)), file=sys.stderr, flush=True)
    r2 = real(model, raw, forced, B, steps=steps, streams=[torch.cuda.Stream() for _ in range(4)])
    print("boundary, fresh streams:           ", json.dumps(four(r2)), file=sys.stderr, flush=True)
    model._side_stream = None
    r3 = real(model, raw, forced, B, steps=steps, streams=[torch.cuda.Stream() for _ in range(4)])
    print("boundary, fresh streams + fresh side streams:", json.dumps(four(r3)), file=sys.stderr, flush=True)
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    r4 = real(model, raw, forced, B, steps=steps, streams=streams)
    print("boundary, loop streams after empty_cache():", json.dumps(four(r4)), file=sys.stderr, flush=True)
    return r


bench.boundary_rate = wrapped
sys.argv = ["bench.py", "--no-tape", "--no-other-configs", "--no-cpu-baseline", "--no-accuracy", "--no-fp32-path"] + sys.argv[1:]
bench.main()
