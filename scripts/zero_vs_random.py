"""MFMA throughput of the conv kernels on zero-filled vs random operands (is the ~900 TFLOP/s plateau a power / clock limit?)."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nopesac_amd import ops, _lib
dev = torch.device("cuda:0")
def t(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
B, H, W, Cin, Cout = 64, 60, 80, 256, 256
st = torch.cuda.current_stream().cuda_stream
for name in ("random", "zeros", "ones"):
    if name == "random":
        x = torch.randn(B, H, W, Cin, device=dev).bfloat16(); w = (torch.randn(Cout, 3, 3, Cin, device=dev) / math.sqrt(9 * Cin)).bfloat16()
    elif name == "zeros":
        x = torch.zeros(B, H, W, Cin, device=dev, dtype=torch.bfloat16); w = torch.zeros(Cout, 3, 3, Cin, device=dev, dtype=torch.bfloat16)
    else:
        x = torch.ones(B, H, W, Cin, device=dev, dtype=torch.bfloat16); w = torch.ones(Cout, 3, 3, Cin, device=dev, dtype=torch.bfloat16)
    s, b = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
    y = torch.empty(B, H, W, Cout, device=dev, dtype=torch.bfloat16)
    wf = ops.mfma_fragment_major(w.reshape(Cout, -1))
    fl = 2.0 * B * H * W * Cout * Cin * 9
    ms = t(lambda: _lib.load().nopesac_conv2d_nhwc_bfrag(x.data_ptr(), wf.data_ptr(), s.data_ptr(), b.data_ptr(), None, y.data_ptr(), B, H, W, Cin, Cout, 3, 3, 1, 1, Cin, Cout, 0, 1, 1, 3, st))
    os.environ["NOPESAC_CONV_FORCE"] = "glds"
    ms2 = t(lambda: ops.conv2d(x, w, s, b, stride=1, pad=1, act=ops.ACT_RELU, out=y))
    print("%-7s bfrag3 %.3f ms %4.0f TF | glds %.3f ms %4.0f TF" % (name, ms, fl / ms / 1e9, ms2, fl / ms2 / 1e9))
