"""GPU box: the fused encoder-layer tail (out-proj + LN1 + FFN + LN2) on the benchmark's shape (64 images x 300 tokens)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nopesac_amd import ops  # noqa: E402
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 19200
g = torch.Generator(device=dev).manual_seed(0)
attn = torch.randn(M, 256, device=dev, generator=g).bfloat16()
src = torch.randn(M, 256, device=dev, generator=g)
fm = lambda n, k: ops.mfma_fragment_major((torch.randn(n, k, device=dev, generator=g) / k ** 0.5).bfloat16())
v = lambda n: 0.1 * torch.randn(n, device=dev, generator=g)
W = {"wo": fm(256, 256), "bo": v(256), "g1": 1 + v(256), "be1": v(256), "w1": fm(1024, 256), "b1": v(1024), "w2": fm(256, 1024), "b2": v(256),
     "g2": 1 + v(256), "be2": v(256)}
pos = torch.randn(300, 256, device=dev, generator=g)
f = lambda: ops.encoder_tail(attn, src, W, pos=pos)
f(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    f()
e1.record(); e1.synchronize()
print("enc_tail M=%d: %.1f us" % (M, 1e3 * e0.elapsed_time(e1) / 10))
