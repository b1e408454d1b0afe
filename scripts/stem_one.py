"""Run the fused stem a few times (target for rocprofv3 --pmc / timing).  usage: stem_one.py [images] [raw|nhwc]
raw (default) = the product path: f32 NCHW images, normalisation folded into the weights (the patch holds v - 128);
raw_norm = the round-3 form that normalises while staging; nhwc = bf16 NHWC input."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nopesac_amd import ops  # noqa: E402
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mode = sys.argv[2] if len(sys.argv) > 2 else "raw"
raw = mode in ("raw", "raw_norm")
w = (torch.randn(64, 224, device=dev) / 12).bfloat16()
sc, bi = torch.ones(64, device=dev), torch.zeros(64, device=dev)
if raw:
    img = torch.randint(0, 256, (B, 3, 480, 640), device=dev).float()
    mean, std = torch.tensor([123.675, 116.28, 103.53], device=dev), torch.tensor([58.395, 57.12, 57.375], device=dev)
    pad3 = mean - 128.0
    run = (lambda: ops.stem_fused_raw_shifted(img, pad3, w, sc, bi)) if mode == "raw" else (lambda: ops.stem_fused_raw(img, mean, std, w, sc, bi))
    nbytes = img.numel() * 4 + B * 120 * 160 * 64 * 2
else:
    x = torch.randn(B, 480, 640, 4, device=dev).bfloat16()
    run = lambda: ops.stem_fused(x, w, sc, bi)
    nbytes = x.numel() * 2 + B * 120 * 160 * 64 * 2
for _ in range(5):
    y = run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    y = run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1000 / 20
print("stem (%s): %.1f us  %.2f TB/s (algorithmic)  env=%s" % (mode, us, nbytes / us / 1e6,
                                                          {k: v for k, v in os.environ.items() if k.startswith("NOPESAC_")}))
