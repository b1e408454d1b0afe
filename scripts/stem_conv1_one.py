"""GPU box: the raw-image stem alone, the per-layer res2.0 conv1 on its output, and the stem launch that evaluates that conv1 itself (FUSE1)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nopesac_amd import ops  # noqa: E402
dev = torch.device("cuda:0")
B = 64
w = (torch.randn(64, 224, device=dev) / 12).bfloat16()
sc, bi = torch.ones(64, device=dev), torch.zeros(64, device=dev)
img = torch.randint(0, 256, (B, 3, 480, 640), device=dev).float()
pad3 = torch.tensor([123.675, 116.28, 103.53], device=dev) - 128.0
w1 = (torch.randn(64, 1, 1, 64, device=dev) / 8).bfloat16()
w1f = ops.mfma_fragment_major(w1.view(64, 64))


def timed(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


y = ops.stem_fused_raw_shifted(img, pad3, w, sc, bi)
for rep in range(2):
    t_stem = timed(lambda: ops.stem_fused_raw_shifted(img, pad3, w, sc, bi))
    t_c1 = timed(lambda: ops.conv2d(y, w1, sc, bi, act=ops.ACT_RELU))
    t_f = timed(lambda: ops.stem_fused_raw_shifted_conv1(img, pad3, w, sc, bi, w1f, sc, bi))
    print("stem %.1f us + conv1 %.1f us = %.1f us; stem with conv1 fused: %.1f us" % (t_stem, t_c1, t_stem + t_c1, t_f))
