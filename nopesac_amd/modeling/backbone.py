"""ResNet-50 backbone on the HIP implicit-GEMM conv (detectron2 `build_resnet_backbone` semantics:
configs/Base.yaml:2-12; call site meta_arch/siamese_planeTR.py:62,456).

NHWC end to end; every conv carries its FrozenBN scale/shift, the residual add and the ReLU in the GEMM
epilogue, so a bottleneck is 3 (4 with projection shortcut) kernel launches and no elementwise passes.
"""
from __future__ import annotations

import os

import torch

from .. import ops
from ..config import amd_options
from ..registry import BACKBONE_REGISTRY
from ..synth import RES_STAGES, state_dict_spec
from .params import DERIVED_EPOCH, ParamModule, conv_bn


class ShapeSpec:
    def __init__(self, channels=None, height=None, width=None, stride=None):
        self.channels, self.height, self.width, self.stride = channels, height, width, stride


class HipResNet50(ParamModule):
    size_divisibility = 0
    STEM_CIN_PAD = 4   # RGB padded to 4 channels so the stem's im2col vectors are 8/16-byte aligned

    def __init__(self, cfg=None):
        spec = {k[len("backbone."):]: v for k, v in state_dict_spec(50).items() if k.startswith("backbone.")}
        super().__init__(spec)
        if cfg is not None:
            r = cfg.MODEL.RESNETS
            assert r.DEPTH == 50 and not r.STRIDE_IN_1X1 and r.NORM == "FrozenBN" and r.NUM_GROUPS == 1, \
                "only the ResNet-50 / FrozenBN / stride-in-3x3 configuration of configs/Base.yaml is implemented"
            self.out_features = list(r.OUT_FEATURES)
        else:
            self.out_features = ["res2", "res3", "res4", "res5"]
        self.fused_stem = True
        # raw-image stem with the normalisation folded into its weights (exact bf16 input operand; NOPESAC_STEM_FOLDED=0: the
        # round-3 form that normalises while staging, bit-identical to preprocess + fused stem)
        self.stem_folded_norm = os.environ.get("NOPESAC_STEM_FOLDED", "1") != "0"
        self.fused_tail = True
        self.halo_conv2 = not os.environ.get("NOPESAC_NO_HALO_CONV2")
        # fp8 mode (MODEL.AMD.BACKBONE_FP8): the 3x3 conv of every res3 / res4 / res5 bottleneck runs on the fp8 MFMA; its input (the block's conv1 output)
        # is written as e4m3fn by the producing kernel.  act_scale[block] = static scale of that input (x ~= x8 * scale).
        self.fp8_conv2 = bool(amd_options(cfg).BACKBONE_FP8) if cfg is not None else False
        self.act_scale: dict = {}
        self._calib: dict | None = None
        self._q8: dict = {}
        self.unfused_wide_tails = os.environ.get("NOPESAC_TAIL_RES4_FUSED", "0") != "1"
        # res3's edge blocks un-fused onto the 128-pixel identity tail: measured 5.39 ms vs 5.35 ms for the whole backbone with the edge
        # blocks left on the 32-pixel fused kernel (profiles/r3_c_tail_ab.txt) - off; NOPESAC_RES3_EDGES_UNFUSED=1 for A/B runs
        self.rt4_res3_edges = os.environ.get("NOPESAC_RES3_EDGES_UNFUSED", "0") == "1"

    def output_shape(self):
        full = {"res2": ShapeSpec(256, stride=4), "res3": ShapeSpec(512, stride=8), "res4": ShapeSpec(1024, stride=16),
                "res5": ShapeSpec(2048, stride=32)}
        return {k: full[k] for k in self.out_features}

    def pack(self) -> dict:
        P = {"stem": conv_bn(self, "stem.conv1.weight", "stem.conv1.norm", 1e-5, cin_pad=self.STEM_CIN_PAD)}
        # fused bf16 stem: [64][kh 7][kw 8 (zero padded)][c 4] -> 224 K per output channel
        w = self.raw("stem.conv1.weight").float().permute(0, 2, 3, 1)                   # [64,7,7,3]
        w8 = w.new_zeros(64, 7, 8, 4)
        w8[:, :, :7, :3] = w
        P["stem_fused_w"] = w8.reshape(64, 224).to(torch.bfloat16).contiguous()
        cin = 64
        for name, n, cmid, cout in RES_STAGES:
            for i in range(n):
                p = f"{name}.{i}"
                for c in ("conv1", "conv2", "conv3") + (("shortcut",) if cin != cout else ()):
                    P[f"{p}.{c}"] = conv_bn(self, f"{p}.{c}.weight", f"{p}.{c}.norm", 1e-5)
                cin = cout
        return P

    def _apply(self, fn, *a, **k):
        self._q8 = {}
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):
        self._q8 = {}
        return super()._load_from_state_dict(*a, **k)

    def calibrate_fp8(self, x: torch.Tensor, headroom: float = 2.0) -> dict:
        """One bf16 forward over representative (normalised NHWC) inputs that records the largest magnitude of every 3x3 conv's
        input and fixes the static activation scales of the fp8 mode so that this maximum maps to 448 / headroom."""
        was, self.fp8_conv2, self._calib = self.fp8_conv2, False, {}
        try:
            with torch.no_grad():
                self.forward(x)
            amax = {k: float(v) for k, v in self._calib.items()}
        finally:
            self.fp8_conv2, self._calib = was, None
        self.act_scale = {k: max(v, 1e-6) * headroom / ops.FP8_MAX for k, v in amax.items()}
        self._q8 = {}
        DERIVED_EPOCH[0] += 1
        return dict(self.act_scale)

    def _quant(self, p: str) -> dict:
        """fp8 operands of block p: fragment-major e4m3fn 3x3 weights, the de-quantising epilogue scale of conv2, and the conv1
        scale / shift with 1 / act_scale folded in (ReLU commutes with a positive factor)."""
        q = self._q8.get(p)
        if q is None:
            P = self.packed
            c1, c2 = P[p + ".conv1"], P[p + ".conv2"]
            xs = float(self.act_scale.get(p, 1.0))
            w8, wsc = ops.quantize_weights_fp8(c2.w(torch.float32))
            q = {"w8": w8, "s2": (c2.scale * wsc * xs).contiguous(), "s1": (c1.scale / xs).contiguous(), "b1": (c1.bias / xs).contiguous()}
            self._q8[p] = q
        return q

    def forward(self, x: torch.Tensor, raw=None, stop_after: str = None, resume=None) -> dict:
        """x: NHWC [B,H,W,4] (normalised, channel-padded) in the compute dtype -> {res2..res5} NHWC.
        bf16 mode only: `raw` = (images f32 NCHW [B,3,H,W], mean f32[3], std f32[3]) may be given INSTEAD of x - the fused stem then
        normalises while it stages its input patches (no NHWC copy of the batch is made).
        Diagnostics (scripts/bf16_attribution.py): `stop_after` = "stem" / "res2" ... returns {"x": the activation after that stage}
        as soon as it exists; `resume` = (stage name, x) continues behind that stage (x in this model's compute dtype)."""
        P = self.packed
        dt = x.dtype if x is not None else torch.bfloat16
        assert x is not None or resume is not None or (raw is not None and self.fused_stem), "backbone: raw images need the fused bf16 stem"
        assert not self.fp8_conv2 or (dt == torch.bfloat16 and self.fused_tail), "MODEL.AMD.BACKBONE_FP8 needs MODEL.AMD.COMPUTE_DTYPE bfloat16"

        def cv(t, key, stride=1, pad=0, act=ops.ACT_RELU, residual=None):
            c = P[key]
            return ops.conv2d(t, c.w(dt), c.scale, c.bias, residual, stride=stride, pad=pad, act=act)

        stages = ["stem"] + [name for name, _, _, _ in RES_STAGES]
        skip_until = None
        if resume is not None:
            skip_until, x = resume
            dt = x.dtype
        elif x is None:
            if self.stem_folded_norm:
                # normalisation folded into the stem's weights / shift, the patch holds v - 128 (exact in bf16): no input rounding
                key = ("stem_shift", raw[1].data_ptr(), raw[2].data_ptr())
                if key not in P:
                    P[key] = ops.fold_stem_normalisation(self.raw("stem.conv1.weight").float().permute(0, 2, 3, 1), P["stem"].scale,
                                                         P["stem"].bias, raw[1], raw[2])
                pad3, wsh, bsh = P[key]
                x = ops.stem_fused_raw_shifted(raw[0], pad3, wsh, P["stem"].scale, bsh)
            else:
                x = ops.stem_fused_raw(raw[0], raw[1], raw[2], P["stem_fused_w"], P["stem"].scale, P["stem"].bias)
        elif dt == torch.bfloat16 and self.fused_stem:
            x = ops.stem_fused(x, P["stem_fused_w"], P["stem"].scale, P["stem"].bias)
        else:
            x = cv(x, "stem", 2, 3)
            x = ops.maxpool(x, 3, 2, 1)
        if stop_after == "stem":
            return {"x": x}
        out = {}
        cin = 64
        fuse = dt == torch.bfloat16 and self.fused_tail
        blocks = [(name, i, n, cmid, cout) for name, n, cmid, cout in RES_STAGES for i in range(n)]
        a_pre = None                                  # conv1 output of the current block, when the previous tail produced it
        for bi, (name, i, n, cmid, cout) in enumerate(blocks):
            if skip_until is not None and stages.index(name) <= stages.index(skip_until):
                cin = cout
                continue
            p = f"{name}.{i}"
            stride = 2 if (i == 0 and name != "res2") else 1
            proj = cin != cout
            fp8 = self.fp8_conv2 and cmid % 128 == 0           # res3-res5 (the fp8 kernel's tiles are 128 output channels wide)
            if fp8:
                q = self._quant(p)
                if a_pre is None:
                    y = ops.conv2d(x, P[p + ".conv1"].w(dt), q["s1"], q["b1"], act=ops.ACT_RELU, out_dtype=torch.float8_e4m3fn)
                else:
                    y = a_pre
            else:
                y = a_pre if a_pre is not None else cv(x, p + ".conv1")
            if self._calib is not None:
                self._calib[p] = torch.maximum(self._calib.get(p, y.new_zeros((), dtype=torch.float32)), y.float().abs().amax())
            if fp8:
                c2 = P[p + ".conv2"]
                y = ops.conv2d_fp8(y, q["w8"], q["s2"], c2.bias, ksize=3, stride=stride, pad=1, act=ops.ACT_RELU)
            elif fuse and cmid == 64 and stride == 1 and self.halo_conv2:
                c2 = P[p + ".conv2"]           # res2: 3x3 out of an LDS halo tile (csrc/conv3x3_c64.hip)
                y = ops.conv3x3_c64(y, c2.w(dt), c2.scale, c2.bias)
            else:
                y = cv(y, p + ".conv2", stride, 1)
            a_pre = None
            nxt = None
            if bi + 1 < len(blocks):
                nn_, ni = blocks[bi + 1][0], blocks[bi + 1][1]
                nxt = P[f"{nn_}.{ni}.conv1"]
            c3 = P[p + ".conv3"]
            cfg_next = (cmid, cout, nxt.cout if nxt is not None else 0, cin if proj else 0)
            cfg_solo = (cmid, cout, 0, cin if proj else 0)
            # res4 (C4 = 1024): the fused tail re-streams 1 MB of weights from L2 per 32-64 pixels and is L2-bound (2.3 TB/s of HBM
            # traffic); as separate launches on the 256x256 persistent conv kernel the expand conv + residual streams at ~4.8 TB/s
            # and the whole tail is 6-24 % faster although y is read once more (scripts/tail_vs_p8.py).  fp8 mode keeps the fused
            # form (its conv1 output is written as fp8 by the tail).
            wide_unfused = self.unfused_wide_tails and cmid >= 256 and not self.fp8_conv2
            if fuse and not wide_unfused and (cfg_next in ops.BOTTLENECK_TAIL_CONFIGS or cfg_solo in ops.BOTTLENECK_TAIL_CONFIGS):
                # conv3 + shortcut + ReLU (+ the next block's conv1) in one launch (csrc/pwchain.hip)
                use_next = cfg_next in ops.BOTTLENECK_TAIL_CONFIGS and nxt is not None
                kw = {}
                # res3's edge blocks on the 128-pixel identity tail (csrc/pwchain.hip: pw_chain_rt4_kernel, round 3): the projection
                # shortcut of res3.0 (K = 256, stride 2) runs as its own launch and enters the tail as a plain residual (the fused
                # projection form needs a 68 KB second operand tile: one workgroup per CU, 32 pixels per weight pass); res3.3 leaves the
                # 256-wide conv1 of res4.0 to the conv kernel (three 32-channel column tiles x four row tiles of accumulators per wave
                # do not fit next to the tail's own).  Off by default (see __init__)
                rt4_edges = self.rt4_res3_edges and cmid == 128 and not self.fp8_conv2
                if rt4_edges and use_next and nxt.cout > 128:
                    use_next = False
                if proj and rt4_edges:
                    kw.update(residual=cv(x, p + ".shortcut", stride, 0, ops.ACT_NONE))
                elif proj:
                    sc = P[p + ".shortcut"]
                    kw.update(x2=x, wsc=sc.wfrag(dt), ssc=sc.scale, bsc=sc.bias, stride=stride)
                else:
                    kw.update(residual=x)
                if use_next and self.fp8_conv2 and blocks[bi + 1][3] % 128 == 0:
                    qn = self._quant(f"{blocks[bi + 1][0]}.{blocks[bi + 1][1]}")
                    kw.update(w1=nxt.wfrag(dt), s1=qn["s1"], b1=qn["b1"], o_fp8=True)
                elif use_next:
                    kw.update(w1=nxt.wfrag(dt), s1=nxt.scale, b1=nxt.bias)
                x, a_pre = ops.bottleneck_tail(y, c3.wfrag(dt), c3.scale, c3.bias, **kw)
            else:
                sc = cv(x, p + ".shortcut", stride, 0, ops.ACT_NONE) if proj else x
                x = cv(y, p + ".conv3", residual=sc)
            cin = cout
            if i == n - 1 and name in self.out_features:
                out[name] = x
            if i == n - 1 and stop_after == name:
                return {"x": x}
        return out


@BACKBONE_REGISTRY.register()
def build_resnet_backbone(cfg, input_shape=None):
    return HipResNet50(cfg)


def build_backbone(cfg):
    return BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)(cfg)
