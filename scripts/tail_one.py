"""Run ONE fused bottleneck-tail configuration a few times (target for rocprofv3 --pmc / timing).
usage: tail_one.py B OH OW C C4 CN [proj_C2 stride]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nopesac_amd import ops  # noqa: E402

B, OH, OW, C, C4, CN = [int(v) for v in sys.argv[1:7]]
C2 = int(sys.argv[7]) if len(sys.argv) > 7 else 0
stride = int(sys.argv[8]) if len(sys.argv) > 8 else 1
dev = torch.device("cuda:0")
rn = lambda *s, k=1.0: (torch.randn(*s, device=dev) * k).bfloat16()
b = rn(B, OH, OW, C)
w3 = ops.mfma_fragment_major(rn(C4, C, k=C ** -0.5))
s3, b3 = torch.ones(C4, device=dev), torch.zeros(C4, device=dev)
kw = {}
if C2:
    kw.update(x2=rn(B, OH * stride, OW * stride, C2), wsc=ops.mfma_fragment_major(rn(C4, C2, k=C2 ** -0.5)), ssc=s3, bsc=b3, stride=stride)
else:
    kw.update(residual=rn(B, OH, OW, C4))
if CN:
    kw.update(w1=ops.mfma_fragment_major(rn(CN, C4, k=C4 ** -0.5)), s1=torch.ones(CN, device=dev), b1=torch.zeros(CN, device=dev))
for _ in range(5):
    y, o = ops.bottleneck_tail(b, w3, s3, b3, **kw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    y, o = ops.bottleneck_tail(b, w3, s3, b3, **kw)
e1.record()
torch.cuda.synchronize()
px = B * OH * OW
fl = 2.0 * px * (C * C4 + C2 * C4 + CN * C4)
by = 2.0 * px * (C + C4 + CN + (C2 / (stride * stride) * 0 + C2 if C2 else C4))
us = e0.elapsed_time(e1) * 1000 / 20
print("tail C=%d C4=%d CN=%d C2=%d: %.1f us  %.0f TFLOP/s  %.0f GB/s (algorithmic)" % (C, C4, CN, C2, us, fl / us / 1e6, by / us / 1e3))
