"""HBM-bound p8 launches under a grid cap (persistent workgroups < CUs): is the launch time flat while CUs are handed back?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nopesac_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [(64, 30, 40, 256, 1024, 1, 1, True), (64, 30, 40, 1024, 256, 1, 1, False), (64, 15, 20, 512, 2048, 1, 1, True), (64, 15, 20, 2048, 512, 1, 1, False),
          (64, 60, 80, 512, 1024, 1, 2, True), (64, 30, 40, 1024, 2048, 1, 2, True), (64, 30, 40, 1024, 512, 1, 1, False), (64, 60, 80, 512, 256, 1, 1, False),
          (64, 60, 80, 256, 256, 1, 1, False), (64, 30, 40, 256, 256, 3, 1, False)]
CAPS = [0, 224, 192, 160, 128, 96, 64]
lib = _lib.load()
for (B, H, W, Cin, Cout, k, s, res) in SHAPES:
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    x = torch.randn(B, H, W, Cin, device=dev).bfloat16()
    w = (torch.randn(Cout, k, k, Cin, device=dev) / (Cin * k * k) ** 0.5).bfloat16()
    sc, bi = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
    r = torch.randn(B, Ho, Wo, Cout, device=dev).bfloat16() if res else None
    y = torch.empty(B, Ho, Wo, Cout, device=dev, dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    row = []
    for cap in CAPS:
        args = (x.data_ptr(), w.data_ptr(), sc.data_ptr(), bi.data_ptr(), r.data_ptr() if res else None, y.data_ptr(), B, H, W, Cin, Cout, k, k, s, pad,
                Cin, Cout, Cout if res else 0, ops.ACT_RELU, 1, 32 | (cap << 8))
        best = 1e9
        for rnd in range(3):
            for _ in range(3):
                assert lib.nopesac_conv2d_nhwc_p8(*args, st) == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                lib.nopesac_conv2d_nhwc_p8(*args, st)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1000 / 20)
        row.append(best)
    mb = (x.numel() + w.numel() + y.numel() * (2 if res else 1)) * 2 / 1e6
    print("x(%d,%d,%d,%d) w(%d,%d,%d) s%d res=%d %5.0f MB | " % (B, H, W, Cin, Cout, k, k, s, res, mb) +
          "  ".join("%s %6.1f us (%.2f TB/s)" % ("all" if c == 0 else str(c), t, mb / t / 1e6 * 1e6 / 1e6) for c, t in zip(CAPS, row)))
