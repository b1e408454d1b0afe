"""GPU box: the pose net's first conv (3x3, 2048 -> 128 at 15 x 20 x 64 images) on the kernels that can take it: the tuner's round-5 choice
(LDS-DMA kernel, 300 workgroups), the 256x128-tile kernel (75 tiles) and its split-K form (75 tiles x 3 slices).  us per call, interleaved."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nopesac_amd import ops  # noqa: E402
dev = torch.device("cuda:0")
B, H, W, Cin, Cout = 64, 15, 20, 2048, 128
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(B, H, W, Cin, device=dev, generator=g).bfloat16()
w = (torch.randn(Cout, 3, 3, Cin, device=dev, generator=g) / math.sqrt(9 * Cin)).bfloat16()
ops.TUNER.measuring = True
res = {}
for name, cfg in (("glds<BK=32> (300 workgroups)", 4), ("p8n (75 tiles)", ops.CFG_P8N), ("p8n split-K (75 x 3)", ops.CFG_P8N_SPLIT)) * 3:
    ops.TUNER.choose = lambda key, launch, extra=(), c=cfg: c
    for _ in range(3):
        y = ops.conv2d(x, w, None, None, pad=1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y = ops.conv2d(x, w, None, None, pad=1)
    e1.record()
    torch.cuda.synchronize()
    assert ops.LAST_CONV_CFG[0] == cfg
    res.setdefault(name, []).append(e0.elapsed_time(e1) * 50)
    if name.startswith("glds"):
        ref = y.float()
    else:
        res.setdefault(name + " max |diff| vs glds", []).append(float((y.float() - ref).abs().max()))
for k, v in res.items():
    print("%-45s %s" % (k, " ".join("%.4g" % t for t in v)))
