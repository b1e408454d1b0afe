"""GPU box: the encoder tail + chained q|k / v projections on the benchmark's shape (19200 tokens), whichever kernel the environment
selects (default: the 96-token form of enc_tail128_kernel; NOPESAC_ENC_TAIL_ROWS=4: 128 tokens; NOPESAC_ENC_TAIL_64=1: the round-3 kernel).
Target of scripts/pmc_summary.sh."""
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
W = {"wo": fm(256, 256), "bo": v(256), "ga": 1 + v(256), "bea": v(256), "w1": fm(1024, 256), "b1": v(1024), "w2": fm(256, 1024), "b2": v(256),
     "gb": 1 + v(256), "beb": v(256)}
pos = torch.randn(300, 256, device=dev, generator=g)
pp, pj = (fm(512, 256), v(512), 512), (fm(256, 256), v(256), 256)
f = lambda: ops.transformer_tail(attn, src, W, pre_norm=False, pos=pos, want=("y",), proj_pos=pp, proj=pj)
f(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    f()
e1.record(); e1.synchronize()
print("encoder tail + projections M=%d: %.1f us  (30.2 GFLOP -> %.0f TFLOP/s)  env=%s" % (M, 1e3 * e0.elapsed_time(e1) / 10, 30.2e3 / (1e3 * e0.elapsed_time(e1) / 10),
                                                                                     {k: v for k, v in os.environ.items() if k.startswith("NOPESAC_")}))
