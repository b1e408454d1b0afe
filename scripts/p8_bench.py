"""GPU box: the 256x256 phase-interleaved conv kernel (nopesac_conv2d_nhwc_p8, variants 0/1/2) against the routed kernels
(bfrag K-tile 64 / 32, halo) on the MFMA-bound layer shapes of the benchmark step; interleaved rounds, random data."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nopesac_amd import _lib, ops  # noqa: E402

SHAPES = [  # B,H,W,Cin,Cout,k,s
    (64, 60, 80, 256, 256, 3, 1),
    (64, 60, 80, 128, 256, 3, 1),
    (64, 30, 40, 256, 256, 3, 1),
    (64, 15, 20, 512, 512, 3, 1),
    (64, 60, 80, 512, 256, 1, 1),
    (64, 60, 80, 256, 256, 1, 1),
    (64, 30, 40, 1024, 256, 1, 1),
    (64, 15, 20, 512, 2048, 1, 1),
    (64, 15, 20, 2048, 512, 1, 1),
    (64, 30, 40, 1024, 2048, 1, 2),
    (64, 60, 80, 256, 256, 3, 2),
    (64, 30, 40, 512, 512, 3, 2),
]


def main():
    dev = torch.device("cuda:0")
    L = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    rounds = int(os.environ.get("ROUNDS", "3"))
    for (B, H, W, Cin, Cout, k, s) in SHAPES:
        x = torch.randn(B, H, W, Cin, device=dev).bfloat16()
        w = (torch.randn(Cout, k, k, Cin, device=dev) / (Cin * k * k) ** 0.5).bfloat16()
        wf = ops._frag_weights(w)
        sc, bi = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
        pad = k // 2
        OH, OW = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
        y = torch.empty(B, OH, OW, Cout, device=dev, dtype=torch.bfloat16)
        flops = 2.0 * B * OH * OW * Cout * Cin * k * k

        def p8(v):
            return lambda: L.nopesac_conv2d_nhwc_p8(x.data_ptr(), w.data_ptr(), sc.data_ptr(), bi.data_ptr(), None, y.data_ptr(), B, H, W, Cin, Cout,
                                                    k, k, s, pad, Cin, Cout, 0, ops.ACT_RELU, 1, v, st)

        def bfrag(n):
            return lambda: L.nopesac_conv2d_nhwc_bfrag(x.data_ptr(), wf.data_ptr(), sc.data_ptr(), bi.data_ptr(), None, y.data_ptr(), B, H, W, Cin,
                                                       Cout, k, k, s, pad, Cin, Cout, 0, ops.ACT_RELU, 1, n, st)

        cands = {"bfrag64": bfrag(3), "bfrag32": bfrag(32), "p8/0": p8(0), "p8/cm": p8(32)}
        if k == 3 and s == 1:
            cands["halo16"] = lambda: L.nopesac_conv3x3_halo_bf16(x.data_ptr(), wf.data_ptr(), sc.data_ptr(), bi.data_ptr(), y.data_ptr(), B, H, W, Cin,
                                                                 Cout, ops.ACT_RELU, 0, st)
        outs, best = {}, {}
        for name, fn in cands.items():
            assert fn() == 0, name
            outs[name] = y.float().clone()
        torch.cuda.synchronize()
        for _ in range(rounds):
            for name, fn in cands.items():
                fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    fn()
                e1.record()
                e1.synchronize()
                t = e0.elapsed_time(e1) / 10
                best[name] = min(best.get(name, t), t)
        ref = outs["bfrag64"]
        line = f"{B}x{H}x{W}x{Cin}->{Cout} k{k} s{s}".ljust(34)
        for name in cands:
            err = float((outs[name] - ref).abs().max())
            line += f"  {name} {best[name] * 1e3:6.1f}us {flops / best[name] / 1e9:5.0f}TF" + ("" if err < 0.05 else f" ERR{err:.2g}")
        print(line, flush=True)


if __name__ == "__main__":
    main()
