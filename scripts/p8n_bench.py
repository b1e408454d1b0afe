"""256x128-tile kernel vs the round-4 routes on the Cout = 128 layers (isolated launches): us per launch and TFLOP/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nopesac_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [(64, 60, 80, 128, 128, 3, 1), (64, 120, 160, 128, 128, 3, 2), (64, 30, 40, 128, 128, 3, 1), (64, 15, 20, 2048, 128, 3, 1),
          (64, 60, 80, 512, 128, 1, 1), (64, 30, 40, 1024, 128, 1, 1), (64, 60, 80, 256, 256, 3, 1), (64, 30, 40, 256, 256, 3, 1)]
lib = _lib.load()
for (B, H, W, Cin, Cout, k, s) in SHAPES:
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    x = torch.randn(B, H, W, Cin, device=dev).bfloat16()
    w = (torch.randn(Cout, k, k, Cin, device=dev) / (Cin * k * k) ** 0.5).bfloat16()
    wf = ops._frag_weights(w)
    sc, bi = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
    y = torch.empty(B, Ho, Wo, Cout, device=dev, dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    cands = {"p8n": lambda: lib.nopesac_conv2d_nhwc_p8n(x.data_ptr(), w.data_ptr(), sc.data_ptr(), bi.data_ptr(), y.data_ptr(), B, H, W, Cin, Cout, k, k, s, pad,
                                                         Cin, Cout, ops.ACT_RELU, 32, st),
             "p8n-tap": lambda: lib.nopesac_conv2d_nhwc_p8n(x.data_ptr(), w.data_ptr(), sc.data_ptr(), bi.data_ptr(), y.data_ptr(), B, H, W, Cin, Cout, k, k, s, pad,
                                                             Cin, Cout, ops.ACT_RELU, 0, st)}
    for name, var in (("bfrag<4,32>", 32), ("bfrag<3,64>", 3)):
        v = var + (256 if (k > 1 and s == 1) else 0)
        cands[name] = (lambda v=v: lib.nopesac_conv2d_nhwc_bfrag(x.data_ptr(), wf.data_ptr(), sc.data_ptr(), bi.data_ptr(), None, y.data_ptr(), B, H, W, Cin, Cout,
                                                                  k, k, s, pad, Cin, Cout, 0, ops.ACT_RELU, 1, v, st))
    for cfg, name in ((3, "glds64"), (4, "glds32")):
        cands[name] = (lambda cfg=cfg: lib.nopesac_conv2d_nhwc_ex(x.data_ptr(), w.data_ptr(), sc.data_ptr(), bi.data_ptr(), None, y.data_ptr(), B, H, W, Cin, Cout,
                                                                  k, k, s, pad, Cin, Cout, 0, 0, ops.ACT_RELU, 1, 1, cfg, st))
    if Cout % 256 == 0:
        cands["p8"] = lambda: lib.nopesac_conv2d_nhwc_p8(x.data_ptr(), w.data_ptr(), sc.data_ptr(), bi.data_ptr(), None, y.data_ptr(), B, H, W, Cin, Cout, k, k, s, pad,
                                                          Cin, Cout, 0, ops.ACT_RELU, 1, 32, st)
    out = {}
    for name, fn in cands.items():
        best = 1e9
        try:
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
        except AssertionError:
            out[name] = float("nan")
    fl = 2.0 * B * Ho * Wo * Cout * k * k * Cin
    print("x(%d,%d,%d,%d) w(%d,%d,%d) s%d tiles %4d nk %3d | " % (B, H, W, Cin, Cout, k, k, s, -(-(B * Ho * Wo) // 256) * (Cout // 128), k * k * Cin // 64) +
          "  ".join("%s %6.1f us %5.0f TF" % (n, t, fl / t / 1e6) for n, t in out.items()))
