"""Stream-K vs plain p8 on the res4 / res5 / pose-net shapes (isolated launches, bf16 output, ReLU): us per launch and TFLOP/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nopesac_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [(64, 30, 40, 256, 256, 3, 1, False), (64, 60, 80, 256, 256, 3, 2, False), (64, 15, 20, 512, 512, 3, 1, False), (64, 30, 40, 512, 512, 3, 2, False),
          (64, 30, 40, 1024, 256, 1, 1, False), (64, 15, 20, 2048, 512, 1, 1, False), (64, 30, 40, 256, 1024, 1, 1, True),
          (64, 15, 20, 512, 2048, 1, 1, True), (64, 30, 40, 1024, 512, 1, 1, False), (64, 60, 80, 256, 256, 3, 1, False), (64, 60, 80, 512, 256, 1, 1, False)]
lib = _lib.load()
ws = ops.p8_sk_workspace(dev)
for (B, H, W, Cin, Cout, k, s, res) in SHAPES:
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    x = torch.randn(B, H, W, Cin, device=dev).bfloat16()
    w = (torch.randn(Cout, k, k, Cin, device=dev) / (Cin * k * k) ** 0.5).bfloat16()
    sc, bi = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
    r = torch.randn(B, Ho, Wo, Cout, device=dev).bfloat16() if res else None
    y = torch.empty(B, Ho, Wo, Cout, device=dev, dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    args = (x.data_ptr(), w.data_ptr(), sc.data_ptr(), bi.data_ptr(), r.data_ptr() if res else None, y.data_ptr(), B, H, W, Cin, Cout, k, k, s, pad,
            Cin, Cout, Cout if res else 0, ops.ACT_RELU, 1, 32)
    out = {}
    for name, fn in (("plain", lambda: lib.nopesac_conv2d_nhwc_p8(*args, st)), ("sk", lambda: lib.nopesac_conv2d_nhwc_p8_sk(*args, ws.data_ptr(), ws.numel(), st))):
        best = 1e9
        for rnd in range(3):
            for _ in range(3):
                assert fn() == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1000 / 20)
        out[name] = best
    fl = 2.0 * B * Ho * Wo * Cout * k * k * Cin
    tiles = -(-(B * Ho * Wo) // 256) * (Cout // 256)
    print("x(%d,%d,%d,%d) w(%d,%d,%d) s%d res=%d tiles %4d nk %3d | plain %7.1f us %6.0f TF | sk %7.1f us %6.0f TF | %.2fx" % (
        B, H, W, Cin, Cout, k, k, s, res, tiles, k * k * Cin // 64, out["plain"], fl / out["plain"] / 1e6, out["sk"], fl / out["sk"] / 1e6, out["plain"] / out["sk"]))
