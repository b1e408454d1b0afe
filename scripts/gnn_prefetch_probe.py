"""GPU box: the one-pair GNN chain (27 launches) with and without the next-layer weight prefetch (NOPESAC_GNN_PREFETCH=0 / 1), from
flushed caches (a 1 GB copy before every call: the live pipeline moves ~100 MB of other weights and activations between two uses of
a layer's weights) and with nothing else running (the weights stay in the Infinity Cache)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import make_model  # noqa: E402
dev = torch.device("cuda:0")
B, nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1, 50
mh = make_model(dev, dtype="bfloat16", nq=nq).matching_head
g = torch.Generator().manual_seed(3)
app = torch.randn(2 * B, nq, 256, generator=g).to(dev)
n_all = torch.full((2 * B,), nq, dtype=torch.int32, device=dev)
f = lambda: mh.descriptors(app, n_all, B)
d0, d1 = f(); torch.cuda.synchronize()
flush_a, flush_b = torch.empty(1 << 28, device=dev, dtype=torch.float32), torch.empty(1 << 28, device=dev, dtype=torch.float32)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(label, flush, reps=12):
    tot = 0.0
    for _ in range(reps):
        if flush:
            flush_a.copy_(flush_b)
        torch.cuda.synchronize()
        e0.record(); f(); e1.record(); e1.synchronize()
        tot += e0.elapsed_time(e1)
    print("prefetch=%s B=%d %-62s %.1f us per call" % (os.environ.get("NOPESAC_GNN_PREFETCH", "1"), B, label, 1e3 * tot / reps), flush=True)


timed("caches flushed before every call", True)
timed("nothing else between calls", False)
print("   checksum %.6f" % float(d0.double().abs().mean() + d1.double().abs().mean()))
