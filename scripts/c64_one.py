"""Run the res2 3x3 halo conv a few times (target for rocprofv3 --pmc / timing)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nopesac_amd import ops  # noqa: E402
dev = torch.device("cuda:0")
x = torch.randn(64, 120, 160, 64, device=dev).bfloat16().relu()
w = (torch.randn(64, 3, 3, 64, device=dev) / 24).bfloat16()
sc, bi = torch.ones(64, device=dev), torch.zeros(64, device=dev)
for _ in range(3):
    y = ops.conv3x3_c64(x, w, sc, bi)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    y = ops.conv3x3_c64(x, w, sc, bi)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 50
print("conv3x3_c64: %.1f us  %.0f TFLOP/s  %.2f TB/s" % (us, 90.6e9 / us / 1e6, 315e6 / us / 1e6))
