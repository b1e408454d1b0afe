"""Which torch (ATen) kernels still run inside one forward of the bench workload, and from which source lines: torch.profiler with
stacks over one eager step.  usage: torch_ops_on_path.py"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nopesac_amd import ops  # noqa: E402

B = 32
dev = torch.device("cuda:0")
model = bench.build_model(dev, 50, "bfloat16")
ops.TUNER.load(os.path.join(ROOT, "profiles", "routing_r3.json"))
raw = torch.randint(0, 256, (2 * B, 3, 480, 640)).float().to(dev)
forced = bench.make_forced(B, 32, 50, dev, 7)
with torch.no_grad():
    for _ in range(2):
        model.forward_tensors(None, B, 480, 640, forced=forced, raw_images=raw)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        model.forward_tensors(None, B, 480, 640, forced=forced, raw_images=raw)
        torch.cuda.synchronize()
import collections
c = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::") or not getattr(ev, "kernels", None):
        continue
    if ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::") and getattr(ev.cpu_parent, "kernels", None):
        continue                                              # count the outermost op that launched something
    st = [f for f in (ev.stack or []) if "/root/repo" in f or "nopesac_amd" in f or "bench.py" in f]
    where = st[0].split("/root/repo/")[-1] if st else (ev.stack[0] if ev.stack else "?")
    c[(ev.name, where, len(ev.kernels))] += 1
for (name, where, nk), n in sorted(c.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print("%3d x %-20s (%d kernel%s)  %s" % (n, name, nk, "" if nk == 1 else "s", where))
print("total torch launches:", sum(n * k[2] for k, n in c.items()))
