"""Run ONE conv shape under one kernel configuration a few times (target for rocprofv3 --pmc)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nopesac_amd import ops  # noqa: E402

B, H, W, Cin, Cout, k, s = [int(v) for v in sys.argv[1:8]]
mode = sys.argv[8] if len(sys.argv) > 8 else ""
if mode and mode != "auto" and not mode.startswith("bfrag") and not mode.startswith("p8"):     # (p8<variant>, p8n<variant>, p8sk<variant>: below)
    os.environ["NOPESAC_CONV_FORCE"] = mode
dev = torch.device("cuda:0")
x = torch.randn(B, H, W, Cin, device=dev).bfloat16()
w = (torch.randn(Cout, k, k, Cin, device=dev) / (Cin * k * k) ** 0.5).bfloat16()
sc, bi = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
res = None
if len(sys.argv) > 9 and sys.argv[9] == "res":
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    res = torch.randn(B, Ho, Wo, Cout, device=dev).bfloat16()
if mode.startswith("bfrag"):
    from nopesac_amd import _lib
    wf = ops._frag_weights(w)
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    yb = torch.empty(B, Ho, Wo, Cout, device=dev, dtype=torch.bfloat16)
    _conv2d = ops.conv2d

    def _bfrag(x, w, sc, bi, res, stride=1, pad=0, act=0):
        rc = _lib.load().nopesac_conv2d_nhwc_bfrag(x.data_ptr(), wf.data_ptr(), sc.data_ptr(), bi.data_ptr(), res.data_ptr() if res is not None else None,
                                                    yb.data_ptr(), B, H, W, Cin, Cout, k, k, stride, pad, Cin, Cout, Cout if res is not None else 0,
                                                    act, 1, int(mode[5:]), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        return yb
    ops.conv2d = _bfrag
if mode.startswith("p8n") or mode.startswith("p8sk"):
    from nopesac_amd import _lib
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    yb = torch.empty(B, Ho, Wo, Cout, device=dev, dtype=torch.bfloat16)
    ws = ops.p8_sk_workspace(dev) if mode.startswith("p8sk") else None

    def _p8x(x, w, sc, bi, res, stride=1, pad=0, act=0):
        if mode.startswith("p8n"):
            rc = _lib.load().nopesac_conv2d_nhwc_p8n(x.data_ptr(), w.data_ptr(), sc.data_ptr(), bi.data_ptr(), yb.data_ptr(), B, H, W, Cin, Cout, k, k, stride, pad,
                                                      Cin, Cout, act, int(mode[3:] or 0), torch.cuda.current_stream().cuda_stream)
        else:
            rc = _lib.load().nopesac_conv2d_nhwc_p8_sk(x.data_ptr(), w.data_ptr(), sc.data_ptr(), bi.data_ptr(), res.data_ptr() if res is not None else None,
                                                        yb.data_ptr(), B, H, W, Cin, Cout, k, k, stride, pad, Cin, Cout, Cout if res is not None else 0,
                                                        act, 1, int(mode[4:] or 0), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        return yb
    ops.conv2d = _p8x
elif mode.startswith("p8"):
    from nopesac_amd import _lib
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    yb = torch.empty(B, Ho, Wo, Cout, device=dev, dtype=torch.bfloat16)
    if "zero" in sys.argv:
        x.zero_(); w.zero_()

    def _p8(x, w, sc, bi, res, stride=1, pad=0, act=0):
        rc = _lib.load().nopesac_conv2d_nhwc_p8(x.data_ptr(), w.data_ptr(), sc.data_ptr(), bi.data_ptr(), res.data_ptr() if res is not None else None,
                                                 yb.data_ptr(), B, H, W, Cin, Cout, k, k, stride, pad, Cin, Cout, Cout if res is not None else 0,
                                                 act, 1, int(mode[2:] or 0), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        return yb
    ops.conv2d = _p8
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
