"""GPU box: the encoder tail + chained projections on the benchmark's shape (64 images x 300 tokens), three kernels A/B/C in one process:
128-token (round 6), 64-token (round 3), 32-token.  Prints us per launch; `NOPESAC_ENC_TAIL_*` is read per launch."""
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


def timed(n=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


for rep in range(3):
    for name, env in (("128", {"NOPESAC_ENC_TAIL_ROWS": "4"}), ("96", {"NOPESAC_ENC_TAIL_ROWS": "3"}), ("64", {"NOPESAC_ENC_TAIL_64": "1"}), ("32", {"NOPESAC_ENC_TAIL_32": "1"})):
        for k in ("NOPESAC_ENC_TAIL_64", "NOPESAC_ENC_TAIL_32", "NOPESAC_ENC_TAIL_ROWS"):
            os.environ.pop(k, None)
        os.environ.update(env)
        print("enc tail %s-token M=%d: %.1f us" % (name, M, timed()), flush=True)
