"""GPU box: the Sinkhorn / assignment kernel at B = 32 pairs for nq = 50 / 64 / 128 full plane sets."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nopesac_amd import ops  # noqa: E402
dev = torch.device("cuda:0")
B = 32
for nq in (50, 63, 64, 100, 128):
    g = torch.Generator(device=dev).manual_seed(nq)
    dots = torch.randn(B, nq, nq, device=dev, generator=g)
    p1 = torch.randn(B, nq, 3, device=dev, generator=g); p2 = torch.randn(B, nq, 3, device=dev, generator=g)
    cam = torch.randn(B, 7, device=dev, generator=g); cam[:, 3:] = torch.nn.functional.normalize(cam[:, 3:], dim=1)
    n = torch.full((B,), nq, device=dev, dtype=torch.int32)
    bin_score = torch.ones(1, device=dev)
    f = lambda: ops.matcher_sinkhorn(dots, p1, p2, cam, n, n, bin_score, 1.0, 20.0, 200, 0.2)
    ls, A = f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        f()
    e1.record(); e1.synchronize()
    t_new = 1e3 * e0.elapsed_time(e1) / 5
    os.environ["NOPESAC_SINKHORN_NO_WG"] = "1"          # the 1024-thread kernel (nq >= 64)
    f(); torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        f()
    e1.record(); e1.synchronize()
    os.environ.pop("NOPESAC_SINKHORN_NO_WG")
    print("sinkhorn nq=%3d: %8.1f us (1024-thread kernel where it differs: %8.1f us)   checksum %.6f  matches %d" % (
        nq, t_new, 1e3 * e0.elapsed_time(e1) / 5, float(ls[:, :nq, :nq].double().sum()), int(A.sum())))
