"""Time individual conv shapes under each kernel configuration (NOPESAC_CONV_FORCE) on the GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nopesac_amd import ops  # noqa: E402

SHAPES = [  # B,H,W,Cin,Cout,k,s, residual
    (64, 120, 160, 64, 256, 1, 1, True),
    (64, 120, 160, 256, 64, 1, 1, False),
    (64, 120, 160, 256, 256, 1, 1, True),
    (64, 60, 80, 128, 512, 1, 1, True),
    (64, 60, 80, 512, 128, 1, 1, False),
    (64, 30, 40, 256, 1024, 1, 1, True),
    (64, 30, 40, 1024, 256, 1, 1, False),
    (64, 15, 20, 2048, 512, 1, 1, False),
    (64, 15, 20, 512, 2048, 1, 1, True),
    (64, 120, 160, 64, 64, 3, 1, False),
    (64, 60, 80, 128, 128, 3, 1, False),
    (64, 30, 40, 256, 256, 3, 1, False),
    (64, 15, 20, 512, 512, 3, 1, False),
    (64, 60, 80, 256, 256, 3, 1, False),
    (64, 60, 80, 512, 256, 1, 1, False),
]


def main():
    dev = torch.device("cuda:0")
    modes = sys.argv[1:] or ["", "t128", "t64", "glds", "glds3", "glds4"]
    modes = ["" if m == "auto" else m for m in modes]
    print("shape".ljust(44) + "".join(m.rjust(26) if m else "auto".rjust(26) for m in modes))
    for (B, H, W, Cin, Cout, k, s, res) in SHAPES:
        x = torch.randn(B, H, W, Cin, device=dev).bfloat16()
        w = (torch.randn(Cout, k, k, Cin, device=dev) / (Cin * k * k) ** 0.5).bfloat16()
        sc, bi = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
        OH = (H + 2 * (k // 2) - k) // s + 1
        OW = (W + 2 * (k // 2) - k) // s + 1
        r = torch.randn(B, OH, OW, Cout, device=dev).bfloat16() if res else None
        y = torch.empty(B, OH, OW, Cout, device=dev, dtype=torch.bfloat16)
        flops = 2.0 * B * OH * OW * Cout * Cin * k * k
        nbytes = 2 * (x.numel() + w.numel() + y.numel() * (2 if res else 1))
        line = f"{B}x{H}x{W}x{Cin}->{Cout} k{k} s{s}{' +res' if res else ''}".ljust(44)
        os.environ["NOPESAC_CONV_FORCE"] = "t128"
        ref = ops.conv2d(x, w, sc, bi, r, stride=s, pad=k // 2, act=ops.ACT_RELU).float()
        for m in modes:
            if m.startswith("bfrag"):
                if Cin % 64 or Cout % 128 or (m == 'bfrag256' and Cout % 256):
                    line += "n/a".rjust(26)
                    continue
                from nopesac_amd import _lib
                wf = ops._frag_weights(w)
                st = torch.cuda.current_stream().cuda_stream

                def call():
                    rc = _lib.load().nopesac_conv2d_nhwc_bfrag(x.data_ptr(), wf.data_ptr(), sc.data_ptr(), bi.data_ptr(), r.data_ptr() if res else None,
                                                                y.data_ptr(), B, H, W, Cin, Cout, k, k, s, k // 2, Cin, Cout, Cout if res else 0,
                                                                ops.ACT_RELU, 1, int(m[5:]), st)
                    assert rc == 0
            else:
                if m:
                    os.environ["NOPESAC_CONV_FORCE"] = m
                else:
                    os.environ.pop("NOPESAC_CONV_FORCE", None)

                def call():
                    ops.conv2d(x, w, sc, bi, r, stride=s, pad=k // 2, act=ops.ACT_RELU, out=y)
            for _ in range(3):
                call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            n = 20
            for _ in range(n):
                call()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            err = float((y.float() - ref).abs().max())
            line += f"{ms:7.3f}ms {flops / ms / 1e9:5.0f}TF {nbytes / ms / 1e6:5.0f}GB/s{'' if err < 1e-6 else ' ERR%.2g' % err}".rjust(26)
        print(line)


if __name__ == "__main__":
    main()
