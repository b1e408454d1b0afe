"""Kernel configurations of ops.conv2d side by side on the backbone's 3x3 layers: time per launch, TFLOP/s, equality with the heuristic
kernel's result.  usage: conv_variant_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nopesac_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
NAMES = dict(ops.CONV_CFG_KERNEL)
NAMES[0] = "heuristic"


def run(B, H, W, Cin, Cout, k, stride, cfgs, residual=False):
    x = (torch.randn(B, H, W, Cin, device=dev) * 0.5).to(torch.bfloat16)
    w = (torch.randn(Cout, k, k, Cin, device=dev) * (1.0 / (k * k * Cin) ** 0.5)).to(torch.bfloat16)
    sc = torch.rand(Cout, device=dev) + 0.5
    bs = torch.randn(Cout, device=dev) * 0.1
    pad = k // 2
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = torch.randn(B, OH, OW, Cout, device=dev).to(torch.bfloat16) if residual else None
    flops = 2.0 * B * OH * OW * Cout * k * k * Cin
    ops.TUNER.loaded = {"_": 0}
    ref = None
    print("x(%d,%d,%d,%d) w(%d,%dx%d) s%d%s" % (B, H, W, Cin, Cout, k, k, stride, " +res" if residual else ""))
    for cfg in cfgs:
        ops.TUNER.choose = lambda key, launch, extra=(), c=cfg: c
        try:
            y = ops.conv2d(x, w, sc, bs, res, stride=stride, pad=pad, act=ops.ACT_RELU)
        except Exception as e:
            print("   %-44s rejected: %s" % (NAMES[cfg], str(e)[:80]))
            continue
        torch.cuda.synchronize()
        if ref is None:
            ref = y.float()
        err = float((y.float() - ref).abs().max())
        for _ in range(3):
            ops.conv2d(x, w, sc, bs, res, stride=stride, pad=pad, act=ops.ACT_RELU)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for rep in range(3):
            e0.record()
            for _ in range(10):
                ops.conv2d(x, w, sc, bs, res, stride=stride, pad=pad, act=ops.ACT_RELU)
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        print("   %-44s %7.1f us  %6.0f TFLOP/s   max|y - y_heuristic| %.3g" % (NAMES[cfg], 1e3 * best, flops / best / 1e9, err))


run(64, 60, 80, 128, 128, 3, 1, (0, 8, 7))
run(64, 120, 160, 128, 128, 3, 2, (0, 8))
run(64, 120, 160, 64, 128, 1, 1, (0, 8))
run(64, 30, 40, 256, 256, 3, 1, (0, 11, 8))
run(64, 15, 20, 512, 512, 3, 1, (0, 11))
run(64, 15, 20, 2048, 128, 3, 1, (0, 3, 8))
run(64, 60, 80, 256, 128, 1, 1, (0, 8))
