"""GPU box: the chained-MLP kernel on the shapes of the camera head's RANSAC stage (B = 32 pairs, nq = 50: 1600 rows), timed against
one launch per layer (ops.linear, the same bf16 weights).  Single-kernel harness for scripts/pmc_summary.sh."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nopesac_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1600
STACKS = {  # name: (K0, [(N, act)])
    "geo_encoder+geo_proj_s1": (8, [(1024, 1)] * 5 + [(1024, 0)] + [(1024, 1)] * 2 + [(1024, 0)]),
    "decoder_rot": (1024, [(512, 1)] * 5 + [(256, 0)]),
    "geo_proj_s2+decoder_tran": (1280, [(1024, 1)] * 2 + [(1024, 0)] + [(512, 1)] * 5 + [(256, 0)]),
    "decoder_rot2+rots": (512, [(512, 1), (512, 1), (256, 1), (4, 0)]),
}
g = torch.Generator(device=dev).manual_seed(0)
tot_chain = tot_layer = 0.0
for name, (k0, spec) in STACKS.items():
    x = torch.randn(rows, k0, device=dev, generator=g)
    layers, ws, k = [], [], k0
    for n, act in spec:
        w = torch.randn(n, k, device=dev, generator=g) * (1.4 / k ** 0.5)
        b = 0.1 * torch.randn(n, device=dev, generator=g)
        layers.append(ops.MlpLayer(w, b)); ws.append((w.bfloat16().contiguous(), b)); k = n
    acts = [a for _, a in spec]
    out = torch.empty(rows, spec[-1][0], device=dev)
    outs = [None] * (len(spec) - 1) + [out]

    def chain():
        ops.mlp_chain(x, layers, acts, outs)

    def per_layer():
        a = x
        for (w, b), act in zip(ws, acts):
            a = ops.linear(a, w, b, act=act)
        return a

    res = {}
    for fn in (chain, per_layer):
        fn(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        res[fn.__name__] = best
    flops = 2.0 * rows * sum(n * kk for (n, _), kk in zip(spec, [k0] + [n for n, _ in spec[:-1]]))
    tot_chain += res["chain"]; tot_layer += res["per_layer"]
    print("%-28s %2d layers  chain %7.1f us (%5.1f TFLOP/s on %d workgroups)   one launch per layer %7.1f us" %
          (name, len(spec), 1e3 * res["chain"], flops / res["chain"] / 1e9, -(-rows // 32), 1e3 * res["per_layer"]))
print("mlp_chain rows %d: 4 stacks chained %.1f us, per-layer launches %.1f us" % (rows, 1e3 * tot_chain, 1e3 * tot_layer))
