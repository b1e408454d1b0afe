"""GPU box: layers 1..5 of the two pose-net branches - one fused launch (csrc/posenet_branch.hip) vs ten ops.conv2d launches."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nopesac_amd import ops  # noqa: E402
from nopesac_amd.modeling.params import ConvW  # noqa: E402
dev = torch.device("cuda:0")
for B in (1, 32):
    g = torch.Generator().manual_seed(1)
    convs = [[ConvW((torch.randn(128, 128, 3, 3, generator=g) / math.sqrt(9 * 64)).to(dev), torch.ones(128, device=dev), torch.zeros(128, device=dev))
              for _ in range(5)] for _ in range(2)]
    xs = [torch.randn(B, 15, 20, 128, generator=g).to(dev, torch.bfloat16) for _ in range(2)]
    packed = ops.PoseBranchTail(convs[0], convs[1])

    def fused():
        return ops.posenet_branch_tail(xs[0], xs[1], packed)

    def layers():
        out = []
        for br in range(2):
            t = xs[br]
            for i, c in enumerate(convs[br]):
                t = ops.conv2d(t, c.w(torch.bfloat16), c.scale, c.bias, stride=2 if i % 2 == 0 else 1, pad=1, act=ops.ACT_LEAKY,
                               out_dtype=torch.float32 if i == 4 else None)
            out.append(t)
        return out

    for name, f in (("fused", fused), ("10 launches", layers)):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record(); e1.synchronize()
        print("B=%d %-12s %.1f us" % (B, name, 50 * e0.elapsed_time(e1)))
