"""Run ONE conv shape under one kernel configuration a few times (target for rocprofv3 --pmc)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nopesac_amd import ops  # noqa: E402

B, H, W, Cin, Cout, k, s = [int(v) for v in sys.argv[1:8]]
mode = sys.argv[8] if len(sys.argv) > 8 else ""
if mode and mode != "auto":
    os.environ["NOPESAC_CONV_FORCE"] = mode
dev = torch.device("cuda:0")
x = torch.randn(B, H, W, Cin, device=dev).bfloat16()
w = (torch.randn(Cout, k, k, Cin, device=dev) / (Cin * k * k) ** 0.5).bfloat16()
sc, bi = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
res = None
if len(sys.argv) > 9 and sys.argv[9] == "res":
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    res = torch.randn(B, Ho, Wo, Cout, device=dev).bfloat16()
for _ in range(5):
    y = ops.conv2d(x, w, sc, bi, res, stride=s, pad=k // 2, act=ops.ACT_RELU)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    y = ops.conv2d(x, w, sc, bi, res, stride=s, pad=k // 2, act=ops.ACT_RELU)
e1.record()
torch.cuda.synchronize()
print("done", tuple(y.shape), "us/launch %.1f" % (e0.elapsed_time(e1) * 1000 / 20))
