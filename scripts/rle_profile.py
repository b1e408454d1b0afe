"""GPU box: time the pieces of rle.encode_views on the benchmark's winner maps (64 views, K = 32 kept planes each)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nopesac_amd import ops, rle  # noqa: E402

dev = torch.device("cuda:0")
B, K = 32, 32
model = bench.build_model(dev, 50, "bfloat16")
ops.TUNER.load(os.path.join(ROOT, "profiles", "routing_r3.json"))
forced = bench.make_forced(B, K, 50, dev, 7)
g = torch.Generator().manual_seed(0)
raw = torch.randint(0, 256, (2 * B, 3, 480, 640), generator=g).float().to(dev)
model.output_rle = True
with torch.no_grad():
    d = model.forward_tensors(None, B, 480, 640, forced=forced, raw_images=raw)
sel = d["sel"]
winner, kept_idx, n_kept, flags = sel["winner"], sel["kept_idx"], sel["n_kept"], sel["flags"]
torch.cuda.synchronize()
V, H, W = winner.shape
nq = kept_idx.shape[1]


def T(fn, name, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    print("%-28s %.2f ms" % (name, 1e3 * (time.perf_counter() - t0) / n))
    return r


labels = T(lambda: ops.rle_labels(winner, kept_idx, n_kept, flags), "rle_labels")
counts = T(lambda: ops.rle_transitions(labels, n_kept, nq), "rle_transitions(count)")
c64 = counts.view(-1).to(torch.int64)
ends = torch.cumsum(c64, 0)
offsets = (ends - c64).contiguous()
total = int(ends[-1].item())
print("n_kept", n_kept.tolist()[:4], "flip positions total", total, "per mask", total / max(int(n_kept.sum()), 1))
pos = torch.empty(max(total, 1), device=dev, dtype=torch.int32)
T(lambda: ops.rle_transitions(labels, n_kept, nq, offsets=offsets.view(V, nq), positions=pos), "rle_transitions(fill)")
res = T(lambda: ops.rle_compress(pos, offsets, counts.view(-1).contiguous(), H, W), "rle_compress (2 passes)")
print("string bytes", res[0].numel())
T(lambda: (res[0].cpu(), res[1].cpu(), res[2].cpu(), res[3].cpu()), "D2H of the results")
T(lambda: rle.encode_views(winner, kept_idx, n_kept, flags), "encode_views (all)")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(3):
    model.package([{"0": {"image_id": "a", "file_name": ""}, "1": {"image_id": "b", "file_name": ""}}] * B, d)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)

inp = [{"0": {"image_id": "a", "file_name": ""}, "1": {"image_id": "b", "file_name": ""}}] * B
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = rle.encode_views(winner, kept_idx, n_kept, flags)
    t1 = time.perf_counter()
    model.package(inp, d)
    t2 = time.perf_counter()
    model.output_rle = False
    model.package(inp, d)
    t3 = time.perf_counter()
    model.output_rle = True
    print("encode_views %.2f ms   package(with rle) %.2f ms   package(no rle) %.2f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2)))
