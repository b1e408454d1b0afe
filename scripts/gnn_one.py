"""GPU box: the 18 GNN layers of the plane matcher (27 launches of gnn_layer_kernel) at B pairs, nq = 50 full plane sets."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import make_model  # noqa: E402
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
nq = 50
mh = make_model(dev, dtype="bfloat16", nq=nq).matching_head
g = torch.Generator().manual_seed(3)
app = torch.randn(2 * B, nq, 256, generator=g).to(dev)
n_all = torch.full((2 * B,), nq, dtype=torch.int32, device=dev)
f = lambda: mh.descriptors(app, n_all, B)
for _ in range(3):
    d0, d1 = f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    f()
e1.record(); e1.synchronize()
print("GNN descriptors B=%d: %.1f us per call (27 layer launches + projections)  checksum %.5f" % (B, 100 * e0.elapsed_time(e1), float(d0.double().abs().mean())))

# warm-L2 experiment: the same chain with ONE layer's weights for all 27 launches (the 1.28 MB stay in the XCD's L2 between launches)
orig = mh._fused_weights
mh._fused_weights = lambda i: orig(0)
for _ in range(3):
    f()
torch.cuda.synchronize()
e0.record()
for _ in range(10):
    f()
e1.record(); e1.synchronize()
print("   same weights in every layer (L2-warm): %.1f us per call" % (100 * e0.elapsed_time(e1)))
