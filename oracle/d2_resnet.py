"""detectron2==0.4 `build_resnet_backbone` semantics, restated as an nn.Module so the reference
meta-arch can be built in this container (oracle/ref_shim.py injects it as `build_backbone`).

TEST INFRASTRUCTURE ONLY (never imported by nopesac_amd/).

Third-party arithmetic: detectron2 is NOT under /root/reference (pinned ==0.4, README.md:28;
selected by configs/Base.yaml:2-12 `MODEL.BACKBONE.NAME: build_resnet_backbone`,
`RESNETS.DEPTH 50, STRIDE_IN_1X1 False, OUT_FEATURES res2..res5`; call site
NopeSAC_Net/modeling/meta_arch/siamese_planeTR.py:62,456).  What is restated (SURVEY.md
Appendix A): BasicStem = 7x7/s2/p3 conv (no bias) + FrozenBN + ReLU + maxpool 3x3/s2/p1;
stages res2..res5 of [3,4,6,3] BottleneckBlocks, bottleneck widths 64/128/256/512, outputs
256/512/1024/2048, first block of res3..5 has stride 2 placed on the 3x3 conv, projection
shortcut (1x1 conv stride s + FrozenBN) when in != out, every conv bias-free + FrozenBN(eps 1e-5).
State-dict names follow d2: stem.conv1.{weight,norm.*}, res{k}.{i}.{conv1,conv2,conv3,shortcut}.*
"""
import torch
from torch import nn
from torch.nn import functional as F


class _FrozenBN(nn.Module):
    def __init__(self, c, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.register_buffer("weight", torch.ones(c))
        self.register_buffer("bias", torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))

    def forward(self, x):
        scale = self.weight * (self.running_var + self.eps).rsqrt()
        shift = self.bias - self.running_mean * scale
        return x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)


class _ConvBN(nn.Module):
    def __init__(self, cin, cout, k, stride=1, pad=0):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(cout, cin, k, k))
        self.norm = _FrozenBN(cout)
        self.stride, self.pad = stride, pad

    def forward(self, x):
        return self.norm(F.conv2d(x, self.weight, None, self.stride, self.pad))


class _Stem(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = _ConvBN(3, 64, 7, 2, 3)

    def forward(self, x):
        return F.max_pool2d(F.relu(self.conv1(x)), kernel_size=3, stride=2, padding=1)


class _Bottleneck(nn.Module):
    def __init__(self, cin, cmid, cout, stride):
        super().__init__()
        self.conv1 = _ConvBN(cin, cmid, 1)
        self.conv2 = _ConvBN(cmid, cmid, 3, stride, 1)  # STRIDE_IN_1X1: False
        self.conv3 = _ConvBN(cmid, cout, 1)
        self.shortcut = _ConvBN(cin, cout, 1, stride) if cin != cout else None

    def forward(self, x):
        y = F.relu(self.conv1(x))
        y = F.relu(self.conv2(y))
        y = self.conv3(y)
        s = x if self.shortcut is None else self.shortcut(x)
        return F.relu(y + s)


class _Shape:
    def __init__(self, channels, stride):
        self.channels, self.stride = channels, stride
        self.height = self.width = None


class D2ResNet50(nn.Module):
    size_divisibility = 0

    def __init__(self):
        super().__init__()
        self.stem = _Stem()
        cin = 64
        for name, n, cmid, cout, stride in (("res2", 3, 64, 256, 1), ("res3", 4, 128, 512, 2),
                                             ("res4", 6, 256, 1024, 2), ("res5", 3, 512, 2048, 2)):
            blocks = []
            for i in range(n):
                blocks.append(_Bottleneck(cin, cmid, cout, stride if i == 0 else 1))
                cin = cout
            setattr(self, name, nn.Sequential(*blocks))

    def output_shape(self):
        return {"res2": _Shape(256, 4), "res3": _Shape(512, 8), "res4": _Shape(1024, 16),
                "res5": _Shape(2048, 32)}

    def forward(self, x):
        out = {}
        x = self.stem(x)
        for name in ("res2", "res3", "res4", "res5"):
            x = getattr(self, name)(x)
            out[name] = x
        return out


def build_resnet50_backbone(cfg, input_shape=None):
    return D2ResNet50()
