"""GPU box: the fused bottleneck tails (pw_chain kernels) against the same work as separate p8 launches (expand conv + residual,
then the next block's reduce conv; plus the projection shortcut where there is one)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nopesac_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.load()
st = torch.cuda.current_stream().cuda_stream
CASES = [  # B, H, W, C, C4, CN, proj (C2, stride)
    (64, 120, 160, 64, 256, 64, None), (64, 120, 160, 64, 256, 128, None), (64, 120, 160, 64, 256, 64, (64, 1)),
    (64, 60, 80, 128, 512, 128, None), (64, 60, 80, 128, 512, 256, None), (64, 60, 80, 128, 512, 128, (256, 2)),
    (64, 30, 40, 256, 1024, 256, None), (64, 30, 40, 256, 1024, 512, None), (64, 30, 40, 256, 1024, 256, (512, 2)),
]


def timeit(fn, n=10, rounds=3):
    best = 1e9
    for _ in range(rounds):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best


def p8(x, w, sc, bi, res, y, B, H, W, Cin, Cout, stride=1):
    rc = L.nopesac_conv2d_nhwc_p8(x.data_ptr(), w.data_ptr(), sc.data_ptr(), bi.data_ptr(), res.data_ptr() if res is not None else None, y.data_ptr(),
                                  B, H, W, Cin, Cout, 1, 1, stride, 0, Cin, Cout, Cout if res is not None else 0, ops.ACT_RELU if True else 0, 1, 32, st)
    assert rc == 0


for (B, H, W, C, C4, CN, proj) in CASES:
    g = lambda *s: torch.randn(*s, device=dev).bfloat16()
    b = g(B, H, W, C).relu()
    w3, w1 = g(C4, 1, 1, C) / C ** 0.5, g(CN, 1, 1, C4) / C4 ** 0.5
    s3, b3, s1, b1 = torch.ones(C4, device=dev), torch.zeros(C4, device=dev), torch.ones(CN, device=dev), torch.zeros(CN, device=dev)
    res = g(B, H, W, C4).relu()
    y = torch.empty(B, H, W, C4, device=dev, dtype=torch.bfloat16)
    o = torch.empty(B, H, W, CN, device=dev, dtype=torch.bfloat16)
    kw = {}
    line = f"{B}x{H}x{W} C={C} C4={C4} CN={CN} proj={proj}".ljust(46)
    if proj:
        C2, s = proj
        x2 = g(B, H * s, W * s, C2).relu()
        wsc = g(C4, 1, 1, C2) / C2 ** 0.5
        ssc, bsc = torch.ones(C4, device=dev), torch.zeros(C4, device=dev)
        sc_out = torch.empty(B, H, W, C4, device=dev, dtype=torch.bfloat16)
        fused = lambda: ops.bottleneck_tail(b, ops.mfma_fragment_major(w3.view(C4, C)), s3, b3, x2=x2, wsc=ops.mfma_fragment_major(wsc.view(C4, C2)),
                                            ssc=ssc, bsc=bsc, stride=s, w1=ops.mfma_fragment_major(w1.view(CN, C4)), s1=s1, b1=b1)
        wf = [ops.mfma_fragment_major(w3.view(C4, C)), ops.mfma_fragment_major(wsc.view(C4, C2)), ops.mfma_fragment_major(w1.view(CN, C4))]
        fused = lambda: ops.bottleneck_tail(b, wf[0], s3, b3, x2=x2, wsc=wf[1], ssc=ssc, bsc=bsc, stride=s, w1=wf[2], s1=s1, b1=b1)
    else:
        wf = [ops.mfma_fragment_major(w3.view(C4, C)), ops.mfma_fragment_major(w1.view(CN, C4))]
        fused = lambda: ops.bottleneck_tail(b, wf[0], s3, b3, residual=res, w1=wf[1], s1=s1, b1=b1)
    t_f = timeit(fused)
    line += f" fused {t_f * 1e3:6.1f}us"
    ok3 = C % 64 == 0 and C4 % 256 == 0
    ok1 = C4 % 64 == 0 and CN % 256 == 0
    t3 = t1 = tsc = None
    if ok3:
        if proj:
            tsc = timeit(lambda: L.nopesac_conv2d_nhwc_p8(x2.data_ptr(), wsc.data_ptr(), ssc.data_ptr(), bsc.data_ptr(), None, sc_out.data_ptr(), B, H * s, W * s,
                                                           C2, C4, 1, 1, s, 0, C2, C4, 0, ops.ACT_NONE, 1, 32, st)) if C2 % 64 == 0 else None
            t3 = timeit(lambda: p8(b, w3, s3, b3, sc_out, y, B, H, W, C, C4))
        else:
            t3 = timeit(lambda: p8(b, w3, s3, b3, res, y, B, H, W, C, C4))
    if ok1:
        t1 = timeit(lambda: p8(y, w1, s1, b1, None, o, B, H, W, C4, CN))
    line += "  p8: conv3+res %s  conv1' %s  shortcut %s" % tuple("%6.1fus" % (t * 1e3) if t else "  n/a  " for t in (t3, t1, tsc))
    if t3 and t1 and (tsc or not proj):
        line += "  sum %6.1fus" % ((t3 + t1 + (tsc or 0)) * 1e3)
    print(line, flush=True)
