"""Run the fused mask head a few times (target for rocprofv3 --pmc / timing)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nopesac_amd import ops  # noqa: E402
dev = torch.device("cuda:0")
B, nq = 64, 50
c1 = (0.5 * torch.randn(B, 120, 160, 256, device=dev)).bfloat16()
t1 = (0.5 * torch.randn(B, 60, 80, 256, device=dev)).bfloat16()
wl = ops.mfma_fragment_major((torch.randn(256, 256, device=dev) / 16).bfloat16())
sc, bi = torch.ones(256, device=dev), torch.zeros(256, device=dev)
mw, mb = torch.randn(B, nq, 256, device=dev) / 16, torch.randn(B, nq, device=dev)
for pipe in (True, False):
    for _ in range(3):
        y = ops.mask_head(c1, t1, wl, sc, bi, mw, mb, pipe=pipe)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        y = ops.mask_head(c1, t1, wl, sc, bi, mw, mb, pipe=pipe)
    e1.record()
    torch.cuda.synchronize()
    print("mask_head (%s): %.1f us" % ("persistent pipeline" if pipe else "two workgroups per CU", e0.elapsed_time(e1) * 1000 / 10))
