"""Run the fused stem a few times (target for rocprofv3 --pmc / timing)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nopesac_amd import ops  # noqa: E402
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = torch.randn(B, 480, 640, 4, device=dev).bfloat16()
w = (torch.randn(64, 224, device=dev) / 12).bfloat16()
sc, bi = torch.ones(64, device=dev), torch.zeros(64, device=dev)
for _ in range(5):
    y = ops.stem_fused(x, w, sc, bi)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    y = ops.stem_fused(x, w, sc, bi)
e1.record()
torch.cuda.synchronize()
print("stem: %.1f us" % (e0.elapsed_time(e1) * 1000 / 20))
