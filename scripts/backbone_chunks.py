"""GPU box: does the bf16 backbone run faster on batch CHUNKS whose activations fit the 256 MB Infinity Cache?  64 images as 1 x 64, 2 x 32,
4 x 16, 8 x 8 (routing file loaded; shapes it does not list run on the library's heuristics - tuned per chunk size with --tune)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nopesac_amd import ops  # noqa: E402
dev = torch.device("cuda:0")
model = bench.build_model(dev, 50, "bfloat16")
routing = os.path.join(ROOT, "gpurun_out", "routing_r4.json")
if os.path.exists(routing):
    ops.TUNER.load(routing)
raw = torch.randint(0, 256, (64, 3, 480, 640)).float().to(dev)
bb = model.backbone
stop = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "--tune" else None
for nchunk in (1, 2, 4, 8):
    parts = raw.chunk(nchunk)

    def one():
        with torch.no_grad():
            return [bb(None, raw=(p, model.pixel_mean, model.pixel_std), stop_after=stop) for p in parts]

    if "--tune" in sys.argv:
        ops.TUNER.measuring = True
        one()
        ops.TUNER.measuring = False
    for _ in range(3):
        one()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            one()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    print("backbone%s 64 images as %d x %d: %.3f ms" % ("" if stop is None else " through " + stop, nchunk, 64 // nchunk, best), flush=True)
