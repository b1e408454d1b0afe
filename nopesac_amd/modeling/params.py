"""Parameter containers: tensors live under the REFERENCE's state-dict names so
`load_state_dict` / `state_dict` / `.to(device)` interoperate with reference checkpoints, while the
kernels consume *packed* copies (NHWC-ordered conv weights, folded BatchNorm scale/shift, permuted FC
columns) built lazily on the device.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import nn

_BUFFER_LEAVES = ("running_mean", "running_var", "num_batches_tracked")
# bumped whenever packed / derived weights of ANY ParamModule are dropped, fp8 scales change or the kernel routing gains a decision:
# PlaneTR_NopeSAC re-captures its hipGraphs when the epoch they were captured under is no longer current
DERIVED_EPOCH = [0]


class _Node(nn.Module):
    pass


class ParamModule(nn.Module):
    """nn.Module whose parameters/buffers are declared by a {dotted_name: shape} spec."""

    def __init__(self, spec: Dict[str, tuple]):
        super().__init__()
        self._spec_keys = list(spec)
        for key, shape in spec.items():
            parts = key.split(".")
            mod = self
            for p in parts[:-1]:
                if not hasattr(mod, p):
                    mod.add_module(p, _Node())
                mod = getattr(mod, p)
            leaf = parts[-1]
            if leaf == "num_batches_tracked":
                mod.register_buffer(leaf, torch.zeros((), dtype=torch.int64))
            elif leaf in _BUFFER_LEAVES:
                mod.register_buffer(leaf, torch.ones(shape) if leaf == "running_var" else torch.zeros(shape))
            else:
                mod.register_parameter(leaf, nn.Parameter(torch.zeros(shape), requires_grad=False))
        self._packed: Optional[dict] = None
        # dtype of the weight operand of head GEMMs whose activations are f32 (bf16 => mixed MFMA mode)
        self.gemm_dtype = torch.float32

    # raw access by reference name
    def raw(self, key: str) -> torch.Tensor:
        mod = self
        parts = key.split(".")
        for p in parts[:-1]:
            mod = getattr(mod, p)
        return getattr(mod, parts[-1])

    def has(self, key: str) -> bool:
        try:
            self.raw(key)
            return True
        except AttributeError:
            return False

    # packed-weight cache invalidation: `_packed` and every derived-weight cache a subclass keeps in its __dict__
    # (fragment-major bf16 copies for the fused kernels) are dropped whenever the raw tensors can have changed
    _DERIVED_CACHES = ("_fused_w", "_enc_tail_w", "_dec_tail_w", "_tail_w", "_mlp_chain_w")

    def _drop_derived(self):
        DERIVED_EPOCH[0] += 1          # anything that holds addresses of packed tensors (captured hipGraphs) is stale now
        self._packed = None
        for name in self._DERIVED_CACHES:
            self.__dict__.pop(name, None)

    def _apply(self, fn, *a, **k):
        self._drop_derived()
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):
        self._drop_derived()
        return super()._load_from_state_dict(*a, **k)

    def invalidate(self):
        self._drop_derived()
        for m in self.children():
            if isinstance(m, ParamModule):
                m.invalidate()

    @property
    def packed(self) -> dict:
        if self._packed is None:
            with torch.no_grad():
                self._packed = self.pack()
        return self._packed

    def pack(self) -> dict:  # pragma: no cover - overridden
        raise NotImplementedError


class ConvW:
    """Packed conv / linear weights: w [Cout,KH,KW,Cin] (f32 master + per-dtype cache), f32 scale/bias."""

    def __init__(self, w_oihw: torch.Tensor, scale=None, bias=None, cin_pad: int = 0):
        w = w_oihw.detach().float()
        if w.dim() == 2:
            w = w[:, :, None, None]
        if w.dim() == 3:     # Conv1d k=1
            w = w[:, :, :, None]
        w = w.permute(0, 2, 3, 1).contiguous()
        if cin_pad and cin_pad > w.shape[-1]:
            w = torch.cat([w, w.new_zeros(*w.shape[:-1], cin_pad - w.shape[-1])], -1).contiguous()
        self._w = {torch.float32: w}
        self.scale = None if scale is None else scale.detach().float().contiguous()
        self.bias = None if bias is None else bias.detach().float().contiguous()
        self.cout, self.kh, self.kw, self.cin = w.shape

    def w(self, dtype=torch.float32) -> torch.Tensor:
        if dtype not in self._w:
            self._w[dtype] = self._w[torch.float32].to(dtype).contiguous()
        return self._w[dtype]

    def w2d(self, dtype=torch.float32) -> torch.Tensor:
        return self.w(dtype).view(self.cout, -1)

    def chain(self):
        """This Linear packed for the chained-MLP kernel (ops.MlpLayer: bf16, K / N zero padded, MFMA fragment-major)."""
        if "chain" not in self._w:
            from .. import ops
            assert self.kh == 1 and self.kw == 1
            self._w["chain"] = ops.MlpLayer(self.w2d(torch.float32), self.bias)
        return self._w["chain"]

    def wfrag(self, dtype=torch.bfloat16) -> torch.Tensor:
        """1x1 weights in MFMA fragment-major order (ops.mfma_fragment_major), for the fused bottleneck tail."""
        key = ("frag", dtype)
        if key not in self._w:
            from .. import ops
            assert self.kh == 1 and self.kw == 1
            self._w[key] = ops.mfma_fragment_major(self.w2d(dtype))
        return self._w[key]


def fold_bn(pm: ParamModule, prefix: str, eps: float):
    """(scale, shift) of an inference-mode BatchNorm: y = x*scale + shift."""
    w, b = pm.raw(prefix + ".weight").float(), pm.raw(prefix + ".bias").float()
    mean, var = pm.raw(prefix + ".running_mean").float(), pm.raw(prefix + ".running_var").float()
    scale = w * torch.rsqrt(var + eps)
    return scale, b - mean * scale


def conv_bn(pm: ParamModule, conv_key: str, bn_prefix: str, eps: float, cin_pad: int = 0) -> ConvW:
    s, b = fold_bn(pm, bn_prefix, eps)
    return ConvW(pm.raw(conv_key), s, b, cin_pad)


def conv_bias(pm: ParamModule, prefix: str) -> ConvW:
    return ConvW(pm.raw(prefix + ".weight"), None, pm.raw(prefix + ".bias") if pm.has(prefix + ".bias") else None)


def mlp_layers(pm: ParamModule, prefix: str) -> list:
    out, i = [], 0
    while pm.has(f"{prefix}.layers.{i}.weight"):
        out.append(conv_bias(pm, f"{prefix}.layers.{i}"))
        i += 1
    return out
