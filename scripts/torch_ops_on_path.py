"""Which torch (ATen) ops that launch a kernel still run inside one forward of the bench workload, and from which source lines
(TorchDispatchMode + the Python stack).  usage: torch_ops_on_path.py"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nopesac_amd import ops  # noqa: E402

QUIET = ("aten.empty", "aten.view", "aten.as_strided", "aten.slice", "aten.select", "aten.reshape", "aten._unsafe_view", "aten.expand",
         "aten.permute", "aten.transpose", "aten.unsqueeze", "aten.squeeze", "aten.detach", "aten.alias", "aten.t.", "aten.unbind",
         "aten.split", "aten.narrow", "aten.lift_fresh", "aten._local_scalar_dense", "aten.is_pinned", "aten.record_stream", "aten.unflatten",
         "aten.flatten", "aten.chunk", "aten._reshape_alias", "aten.empty_like", "aten.empty_strided", "aten.new_empty")
seen = collections.Counter()


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(QUIET):
            fr = [f for f in traceback.extract_stack() if "/root/repo/" in f.filename or "nopesac_amd" in f.filename]
            fr = [f for f in fr if "torch_ops_on_path" not in f.filename]
            where = "%s:%d" % (fr[-1].filename.split("/root/repo/")[-1], fr[-1].lineno) if fr else "?"
            seen[(name, where)] += 1
        return func(*args, **(kwargs or {}))


B = 32
dev = torch.device("cuda:0")
model = bench.build_model(dev, 50, "bfloat16")
ops.TUNER.load(os.path.join(ROOT, "profiles", "routing_r5.json"))
raw = torch.randint(0, 256, (2 * B, 3, 480, 640)).float().to(dev)
forced = bench.make_forced(B, 32, 50, dev, 7)
with torch.no_grad():
    for _ in range(2):
        model.forward_tensors(None, B, 480, 640, forced=forced, raw_images=raw)
    torch.cuda.synchronize()
    with Log():
        model.forward_tensors(None, B, 480, 640, forced=forced, raw_images=raw)
    torch.cuda.synchronize()
for (name, where), n in sorted(seen.items(), key=lambda kv: kv[0][1]):
    print("%3d x %-34s %s" % (n, name, where))
