"""Plane matching head on HIP kernels (matching_net/matching_head.py:43-133; transformer/gnn.py).

Ragged plane sets are kept padded to nq rows per image with int32 length vectors on the device; the
attention kernel masks keys >= n and the Sinkhorn kernel only reads the n1 x n2 block, so padded rows
never influence a live row (and no host synchronisation is needed to learn n1, n2).
"""
from __future__ import annotations

import torch

from .. import ops
from ..registry import MATCHING_HEAD_REGISTRY
from ..synth import state_dict_spec
from .params import ConvW, ParamModule, conv_bias


@MATCHING_HEAD_REGISTRY.register()
class MatchingHead(ParamModule):
    def __init__(self, cfg):
        self.offset_multiplier = float(cfg.MODEL.MATCHING_HEAD.OFFSET_MULTIPLIER)
        self.normal_multiplier = float(cfg.MODEL.MATCHING_HEAD.NORMAL_MULTIPLIER)
        self.sinkhorn_iterations = 200                                    # matching_head.py:38
        self.num_queries = cfg.MODEL.SEM_SEG_HEAD.NUM_OBJECT_QUERIES
        self.fused_gnn = True
        spec = {k[len("matching_head."):]: v for k, v in state_dict_spec(self.num_queries).items()
                if k.startswith("matching_head.")}
        super().__init__(spec)

    def pack(self) -> dict:
        P = {"app": conv_bias(self, "planeApp_proj"), "desc": conv_bias(self, "planeDesc_proj"), "layers": []}
        for i in range(18):
            p = f"gnn.layers.{i}"
            w0 = self.raw(p + ".mlp.0.weight").float()
            P["layers"].append({
                "q": ConvW(self.raw(p + ".q_proj.weight")),
                "kv": ConvW(torch.cat([self.raw(p + ".k_proj.weight").float(), self.raw(p + ".v_proj.weight").float()], 0)),
                "qkv": ConvW(torch.cat([self.raw(p + ".q_proj.weight").float(), self.raw(p + ".k_proj.weight").float(),
                                        self.raw(p + ".v_proj.weight").float()], 0)),
                "merge": ConvW(self.raw(p + ".merge.weight")),
                "mlp0_x": ConvW(w0[:, :256].contiguous()), "mlp0_m": ConvW(w0[:, 256:].contiguous()),
                "mlp2": ConvW(self.raw(p + ".mlp.2.weight")), "prefix": p})
        dev = self.raw("bin_score").device
        P["dot_scale"] = torch.full((self.num_queries,), 1.0 / 256 ** 0.5, device=dev, dtype=torch.float32)
        return P

    def _gnn_layer(self, W, x, src, nb, qlen, klen):
        """x [nb*nq,256], src [nb*nq,256] -> x + LN(mlp(cat[x, LN(merge(attn))]))   (gnn.py:73-96)."""
        nq, gd = self.num_queries, self.gemm_dtype
        if x is src:                                   # self layer: one fused q|k|v projection
            qkv = ops.linear(x, W["qkv"].w2d(gd))
            q, k, v = qkv[:, :256], qkv[:, 256:512], qkv[:, 512:]
        else:
            q = ops.linear(x, W["q"].w2d(gd))
            kv = ops.linear(src, W["kv"].w2d(gd))
            k, v = kv[:, :256], kv[:, 256:]
        msg = ops.attention(q, k, v, nb, nq, nq, 8, 32 ** -0.5, qlen, klen, mfma_bf16=(gd == torch.bfloat16))
        p = W["prefix"]
        msg = ops.layernorm(ops.linear(msg, W["merge"].w2d(gd)), self.raw(p + ".norm1.weight"), self.raw(p + ".norm1.bias"))
        h = ops.linear(x, W["mlp0_x"].w2d(gd))
        h = ops.linear(msg, W["mlp0_m"].w2d(gd), residual=h, act=ops.ACT_RELU)
        out = ops.linear(h, W["mlp2"].w2d(gd))
        _, y = ops.layernorm(out, self.raw(p + ".norm2.weight"), self.raw(p + ".norm2.bias"), addend=x)
        return y

    def _fused_weights(self, i: int) -> dict:
        """Layer i's weights for the fused kernel: bf16, MFMA fragment-major, 1/sqrt(32) folded into Wq."""
        cache = self.__dict__.setdefault("_fused_w", {})
        if i not in cache:
            p = f"gnn.layers.{i}"
            fm = lambda w: ops.mfma_fragment_major(w.float().to(torch.bfloat16).contiguous())
            cache[i] = {"wq": fm(self.raw(p + ".q_proj.weight").float() * (32 ** -0.5)), "wk": fm(self.raw(p + ".k_proj.weight")),
                        "wv": fm(self.raw(p + ".v_proj.weight")), "wm": fm(self.raw(p + ".merge.weight")),
                        "w0": fm(self.raw(p + ".mlp.0.weight")), "w2": fm(self.raw(p + ".mlp.2.weight")),
                        "g1": self.raw(p + ".norm1.weight").float().contiguous(), "b1": self.raw(p + ".norm1.bias").float().contiguous(),
                        "g2": self.raw(p + ".norm2.weight").float().contiguous(), "b2": self.raw(p + ".norm2.bias").float().contiguous()}
        return cache[i]

    def descriptors(self, app: torch.Tensor, n_all: torch.Tensor, B: int):
        """app [2B,nq,256] (view-1 sets first), n_all int32[2B] -> GNN descriptors d0, d1 [B,nq,256]."""
        P, nq, gd = self.packed, self.num_queries, self.gemm_dtype
        f = ops.linear(app.reshape(2 * B * nq, 256), P["app"].w2d(gd), P["app"].bias)
        if gd == torch.bfloat16 and nq <= 128 and self.fused_gnn:
            # one launch per layer step, one workgroup per plane set (csrc/gnn_layer.hip): 27 launches instead of 234
            cur, nxt = f.view(2 * B, nq, 256), torch.empty(2 * B, nq, 256, device=f.device, dtype=torch.float32)
            for i in range(18):
                W = self._fused_weights(i)
                # (few plane sets: the last launch that uses a layer's weights also pulls the next layer's into L2 - ops.gnn_layer)
                Wn = self._fused_weights(i + 1) if i + 1 < 18 and 2 * B <= 16 else None
                if i % 2 == 0:
                    ops.gnn_layer(cur, 0, cur, 0, nxt, 0, 2 * B, n_all, W, Wn, B)
                else:                            # feat1 attends to the UPDATED feat0 (gnn.py:131-133)
                    ops.gnn_layer(cur, 0, cur, B, nxt, 0, B, n_all, W)
                    ops.gnn_layer(cur, B, nxt, 0, nxt, B, B, n_all, W, Wn, 2 * B)
                cur, nxt = nxt, cur
            d = ops.linear(cur.view(2 * B * nq, 256), P["desc"].w2d(gd), P["desc"].bias)
            return d[:B * nq].view(B, nq, 256), d[B * nq:].view(B, nq, 256)
        n1, n2 = n_all[:B], n_all[B:]
        for i, W in enumerate(P["layers"]):
            if i % 2 == 0:                       # 'self' (gnn.py:128-130): both sets in one launch
                f = self._gnn_layer(W, f, f, 2 * B, n_all, n_all)
            else:                                # 'cross' (gnn.py:131-133): feat1 attends to the UPDATED feat0
                f0 = self._gnn_layer(W, f[:B * nq], f[B * nq:], B, n1, n2)
                f1 = self._gnn_layer(W, f[B * nq:], f0, B, n2, n1)
                f = torch.cat([f0, f1], 0)
        d = ops.linear(f, P["desc"].w2d(gd), P["desc"].bias)
        return d[:B * nq].view(B, nq, 256), d[B * nq:].view(B, nq, 256)

    def forward(self, app: torch.Tensor, n_all: torch.Tensor, cam7: torch.Tensor, planes1: torch.Tensor,
                planes2: torch.Tensor, match_thr: float):
        """-> (log_scores_padded [B,nq+1,nq+1] with the dustbin at index nq, assignment [B,nq,nq])."""
        P, nq = self.packed, self.num_queries
        B = cam7.shape[0]
        d0, d1 = self.descriptors(app, n_all, B)
        dots = ops.conv2d(d0.view(B, 1, nq, 256), d1.reshape(B, nq, 1, 1, 256), P["dot_scale"], batched_weights=True)
        return ops.matcher_sinkhorn(dots.view(B, nq, nq), planes1, planes2, cam7, n_all[:B].contiguous(), n_all[B:].contiguous(),
                                    self.raw("bin_score").reshape(1), self.offset_multiplier, self.normal_multiplier,
                                    self.sinkhorn_iterations, float(match_thr))


def build_matching_head(cfg):
    return MATCHING_HEAD_REGISTRY.get("MatchingHead")(cfg)
