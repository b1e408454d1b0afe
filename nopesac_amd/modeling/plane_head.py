"""PlaneTR head on HIP kernels (planeTR_net/planeTR_head.py:116-192; transformer/transformer.py;
transformer/position_encoding.py) + fused plane post-selection (meta_arch/siamese_planeTR.py:625-803).

Token matrices are batch-major [B*L, 256] fp32.  Per encoder layer: 4 GEMMs (fused q|k projection, v,
out-proj with the residual in the epilogue, 2 FFN GEMMs with ReLU / residual epilogues), 1 attention
launch, 2 LayerNorm launches (which also emit `x + pos` for the next layer's q/k).  Only the last decoder
layer's heads are evaluated (the reference's deep-supervision outputs are training-only).
"""
from __future__ import annotations

import os

import math

import torch

from .. import _lib, ops
from ..registry import SEM_SEG_HEADS_REGISTRY
from ..synth import state_dict_spec
from .params import ConvW, ParamModule, conv_bias, conv_bn, mlp_layers


def sine_position_embedding(h: int, w: int, num_pos_feats: int = 128) -> torch.Tensor:
    """[h*w, 2*num_pos_feats] table of transformer/position_encoding.py:29-52 (normalize=True, no mask);
    input independent, built once on the host."""
    eps, scale, temperature = 1e-6, 2 * math.pi, 10000.0
    y = torch.arange(1, h + 1, dtype=torch.float32).view(h, 1).expand(h, w)
    x = torch.arange(1, w + 1, dtype=torch.float32).view(1, w).expand(h, w)
    y = y / (y[-1:, :] + eps) * scale
    x = x / (x[:, -1:] + eps) * scale
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / num_pos_feats)
    px, py = x[:, :, None] / dim_t, y[:, :, None] / dim_t
    px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), dim=3).flatten(2)
    py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), dim=3).flatten(2)
    return torch.cat((py, px), dim=2).reshape(h * w, 2 * num_pos_feats).contiguous()


def run_mlp(x, layers, out=None, final_act=ops.ACT_NONE, gd=torch.float32):
    if gd == torch.bfloat16 and x.dtype == torch.float32 and x.dim() == 2:
        return run_stacks(x, [(layers, final_act, out)], gd)[0]
    for i, l in enumerate(layers):
        last = i == len(layers) - 1
        x = ops.linear(x, l.w2d(gd), l.bias, act=final_act if last else ops.ACT_RELU, out=out if last else None)
    return x


CHAIN_MIN_ROWS = int(os.environ.get("NOPESAC_CHAIN_MIN_ROWS", "64"))     # up to this many rows big stacks run one launch per layer
CHAIN_SMALL_PARAMS = 1 << 20                                              # ... when they hold at least this many weights (2 MB in bf16)


def run_stacks(x, stacks, gd, x_bcast=None, rows_per=1, parallel=False):
    """Consecutive MLP stacks (each: ReLU between its layers, `final_act` after its last one) applied to the rows of x [rows, K] f32;
    stacks = [(layers: [ConvW], final_act, out)], out = an f32 [rows, N] tensor (may be a column slice) that receives the stack's
    output, True to allocate one, None if only the next stack consumes it (the last stack always returns its output).
    x_bcast [P, Kb]: optional input prefix shared by `rows_per` consecutive rows.  Returns one entry per stack (tensor or None).
    parallel = True, or a list with one flag per stack: a flagged stack reads x instead of the previous stack's output (independent
    stacks over the same rows; still one launch in bf16 GEMM mode: NOPESAC_MLP_RESTART).
    bf16 GEMM mode: ONE launch for everything (csrc/mlp_chain.hip, activations stay in LDS); fp32 mode: one launch per layer."""
    rows = x.shape[0]
    par = list(parallel) if isinstance(parallel, (list, tuple)) else [bool(parallel)] * len(stacks)
    flat, last_of, restarts = [], [], []
    for si, (layers, final_act, out) in enumerate(stacks):
        for i, l in enumerate(layers):
            last = i == len(layers) - 1
            flat.append((l, final_act if last else ops.ACT_RELU))
            last_of.append(si if last else -1)
            restarts.append(par[si] and si > 0 and i == 0)
    want = [o for (_, _, o) in stacks]
    for si in range(len(stacks)):                             # a stack nothing in the chain consumes always returns its output
        if want[si] is None and (si == len(stacks) - 1 or par[si + 1]):
            want[si] = True
    results = [None] * len(stacks)
    for si, o in enumerate(want):
        if o is True:
            results[si] = torch.empty(rows, stacks[si][0][-1].cout, device=x.device, dtype=torch.float32)
        elif o is not None:
            results[si] = o
    k0 = x.shape[1] + (0 if x_bcast is None else x_bcast.shape[1])
    # One pair per call (round 4): with <= 64 rows the chained kernel is ONE or TWO workgroups streaming the whole stack's weights
    # through one CU's L2 port (the refine stacks: 10 MB, 263 us for 33-64 rows); one launch per layer spreads every weight matrix
    # over N / 64 workgroups (~12 us per layer).  Same rounding points (f32 activations, bf16 operands), another summation order.
    few_rows = rows <= CHAIN_MIN_ROWS and sum(l.cout * l.cin for l, _ in flat) >= CHAIN_SMALL_PARAMS
    if (gd == torch.bfloat16 and x.dtype == torch.float32 and len(flat) <= _lib.MLP_MAX_LAYERS and k0 <= _lib.MLP_MAX_IN
            and all(l.cout <= _lib.MLP_MAX_WIDTH for l, _ in flat) and not few_rows):
        ops.mlp_chain(x, [l.chain() for l, _ in flat], [a for _, a in flat], [results[si] if si >= 0 else None for si in last_of],
                      x_bcast=x_bcast, rows_per=rows_per, restarts=restarts)
        return results
    if x_bcast is not None:                       # per-layer path: materialise the concatenated input once
        assert rows == x_bcast.shape[0] * rows_per, "run_stacks: x_bcast rows x rows_per must equal the rows of x"
        full = torch.empty(rows, k0, device=x.device, dtype=torch.float32)
        full.view(-1, rows_per, k0)[:, :, :x_bcast.shape[1]] = x_bcast[:, None, :]
        full[:, x_bcast.shape[1]:] = x
        x = full
    x_in = x
    for (l, a), si, rs in zip(flat, last_of, restarts):
        # bf16 GEMM mode: a layer output only the NEXT layer reads is stored as bf16 - the value the consumer's MFMA operand would be
        # rounded to anyway (same rounding point as the chained kernel and as the f32 -> bf16 staging of the mixed GEMM kernel), half the
        # activation traffic, and the all-bf16 small-grid kernel with two tiles of prefetch instead of the mixed one
        inner = gd == torch.bfloat16 and (si < 0 or results[si] is None)
        x = ops.linear(x_in if rs else x, l.w2d(gd), l.bias, act=a, out=results[si] if si >= 0 else None,
                       out_dtype=torch.bfloat16 if inner else None)
    return results


@SEM_SEG_HEADS_REGISTRY.register()
class PlaneTRHead(ParamModule):
    def __init__(self, cfg, input_shape=None):
        H = cfg.MODEL.SEM_SEG_HEAD
        self.num_queries = H.NUM_OBJECT_QUERIES
        self.nheads = H.NHEADS
        assert H.HIDDEN_DIM == 256 and H.MASK_DIM == 256 and H.NHEADS == 8 and H.ENC_LAYERS == 6 and H.DEC_LAYERS == 6, \
            "PlaneTRHead kernels are specialised to hidden 256 / 8 heads / 6+6 layers (config/config.py:46-51)"
        assert H.PARAM_ON and H.CENTER_ON and not cfg.MODEL.DEPTH_ON, "inference configs set PARAM_ON/CENTER_ON, no depth"
        spec = {k[len("sem_seg_head."):]: v for k, v in state_dict_spec(self.num_queries).items()
                if k.startswith("sem_seg_head.")}
        super().__init__(spec)
        self._pos_cache = {}
        self.fused_encoder_tail = True
        self.fused_mask_head = True
        self.fused_decoder_tail = True
        self.chain_projections = True           # the tails also compute the next attention's input projections (one launch per half layer)

    # ---------------------------------------------------------------- packing
    def _mha(self, prefix: str, fuse_qk: bool):
        w, b = self.raw(prefix + ".in_proj_weight").float(), self.raw(prefix + ".in_proj_bias").float()
        E = 256
        d = {"o": conv_bias(self, prefix + ".out_proj"), "v": ConvW(w[2 * E:], None, b[2 * E:])}
        if fuse_qk:
            d["qk"] = ConvW(w[:2 * E], None, b[:2 * E])
        else:
            d["q"], d["k"] = ConvW(w[:E], None, b[:E]), ConvW(w[E:2 * E], None, b[E:2 * E])
        return d

    def pack(self) -> dict:
        P = {"input_proj": conv_bias(self, "input_proj")}
        for i in range(6):
            p = f"context_SA.layers.{i}"
            P[p] = {"attn": self._mha(p + ".self_attn", True), "l1": conv_bias(self, p + ".linear1"),
                    "l2": conv_bias(self, p + ".linear2")}
            p = f"context2plane_decoder.layers.{i}"
            P[p] = {"self": self._mha(p + ".self_attn", True), "cross": self._mha(p + ".multihead_attn", False),
                    "l1": conv_bias(self, p + ".linear1"), "l2": conv_bias(self, p + ".linear2")}
        # decoder cross-attention K / V projections of all six layers act on the same encoder memory: one GEMM each
        # (N = 6 x 256) instead of twelve M = B*L, N = 256 launches
        wk = [self.raw(f"context2plane_decoder.layers.{i}.multihead_attn.in_proj_weight").float()[256:512] for i in range(6)]
        bk = [self.raw(f"context2plane_decoder.layers.{i}.multihead_attn.in_proj_bias").float()[256:512] for i in range(6)]
        wv = [self.raw(f"context2plane_decoder.layers.{i}.multihead_attn.in_proj_weight").float()[512:768] for i in range(6)]
        bv = [self.raw(f"context2plane_decoder.layers.{i}.multihead_attn.in_proj_bias").float()[512:768] for i in range(6)]
        P["cross_k_all"] = ConvW(torch.cat(wk, 0), None, torch.cat(bk, 0))
        P["cross_v_all"] = ConvW(torch.cat(wv, 0), None, torch.cat(bv, 0))
        for nm in ("up_conv3", "up_conv2", "up_conv1", "c4_conv", "c3_conv", "c2_conv", "c1_conv", "m_conv_dict.m4"):
            P[nm] = conv_bn(self, f"top_down.{nm}.0.weight", f"top_down.{nm}.1", 1e-5)
        for nm in ("plane_embedding", "plane_param", "plane_center"):
            P[nm] = mlp_layers(self, nm)
        P["pixel_embedding"] = conv_bias(self, "pixel_embedding")
        # mask logit = plane_emb . (W_pe p1 + b_pe) = (W_pe^T plane_emb) . p1 + plane_emb . b_pe: the 256x256 pixel-embedding conv
        # over the full 120x160 map folds into the per-image 50x256 mask weights (rows 0..255 = W_pe^T, row 256 = b_pe)
        w_pe = self.raw("pixel_embedding.weight").float().reshape(256, 256)
        fold = torch.zeros(264, 256, device=w_pe.device)
        fold[:256] = w_pe.t()
        fold[256] = self.raw("pixel_embedding.bias").float()
        P["pe_fold"] = ConvW(fold)
        P["pixel_plane_center"] = conv_bias(self, "pixel_plane_center")
        P["plane_prob"] = conv_bias(self, "plane_prob")
        return P

    def _ln(self, x, prefix, **kw):
        return ops.layernorm(x, self.raw(prefix + ".weight"), self.raw(prefix + ".bias"), **kw)

    def _pos(self, h, w, device):
        key = (h, w, str(device))
        if key not in self._pos_cache:
            self._pos_cache[key] = sine_position_embedding(h, w).to(device)
        return self._pos_cache[key]

    # ---------------------------------------------------------------- forward
    def forward(self, features: dict, want_logits: bool = False):
        """features: NHWC res2..res5 (compute dtype).  Returns (outputs, query_feat [B,nq,256]) where outputs
        holds pred_logits [B,nq,2], pred_params [B,nq,3], pred_centers [B,nq,2], mask_prob [B,h,w,nq]
        (= sigmoid(pred_mask_logits), NHWC) and, if `want_logits`, pred_mask_logits / pixel_centers."""
        P, gd = self.packed, self.gemm_dtype
        c1, c2, c3, c4 = features["res2"], features["res3"], features["res4"], features["res5"]
        cd = c4.dtype
        B, hc, wc, _ = c4.shape
        L, nq, nh = hc * wc, self.num_queries, self.nheads
        scale = 32 ** -0.5
        mf = gd == torch.bfloat16          # bf16 mode: MFMA attention kernel
        pos = self._pos(hc, wc, c4.device)
        ip = P["input_proj"]
        src = ops.conv2d(c4, ip.w(cd), None, ip.bias, out_dtype=torch.float32).view(B * L, 256)
        mark = getattr(self, "mark", None) or (lambda name: None)
        mark("ph.input_proj")
        if mf:
            hs, memory = self._transformer_bf16(src, pos, B, L, nq, nh, scale, mark)
            mark("ph.decoder")
            return self._heads(features, hs, memory, B, hc, wc, nq, want_logits, mark)
        q_in = ops.add_rows(src, pos)
        # ---- encoder (post-norm; transformer.py:183-199)
        for i in range(6):
            p = f"context_SA.layers.{i}"
            W = P[p]
            qk = ops.linear(q_in, W["attn"]["qk"].w2d(gd), W["attn"]["qk"].bias)
            v = ops.linear(src, W["attn"]["v"].w2d(gd), W["attn"]["v"].bias)
            o = ops.attention(qk[:, :256], qk[:, 256:], v, B, L, L, nh, scale, mfma_bf16=mf)
            s = ops.linear(o, W["attn"]["o"].w2d(gd), W["attn"]["o"].bias, residual=src)
            src = self._ln(s, p + ".norm1")
            hdn = ops.linear(src, W["l1"].w2d(gd), W["l1"].bias, act=ops.ACT_RELU)
            s = ops.linear(hdn, W["l2"].w2d(gd), W["l2"].bias, residual=src)
            src, q_in = self._ln(s, p + ".norm2", addend=pos)
        memory, mem_k = self._ln(src, "context_SA.norm", addend=pos)
        mark("ph.encoder")
        # ---- decoder (pre-norm; transformer.py:293-322), only hs[-1] is needed at inference
        qpos = self.raw("query_embed.weight")
        tgt = torch.zeros(B * nq, 256, device=c4.device, dtype=torch.float32)
        for i in range(6):
            p = f"context2plane_decoder.layers.{i}"
            W = P[p]
            t2, q_in = self._ln(tgt, p + ".norm1", addend=qpos)
            qk = ops.linear(q_in, W["self"]["qk"].w2d(gd), W["self"]["qk"].bias)
            v = ops.linear(t2, W["self"]["v"].w2d(gd), W["self"]["v"].bias)
            o = ops.attention(qk[:, :256], qk[:, 256:], v, B, nq, nq, nh, scale, mfma_bf16=mf)
            tgt = ops.linear(o, W["self"]["o"].w2d(gd), W["self"]["o"].bias, residual=tgt)
            t2, q_in = self._ln(tgt, p + ".norm2", addend=qpos)
            q = ops.linear(q_in, W["cross"]["q"].w2d(gd), W["cross"]["q"].bias)
            k = ops.linear(mem_k, W["cross"]["k"].w2d(gd), W["cross"]["k"].bias)
            v = ops.linear(memory, W["cross"]["v"].w2d(gd), W["cross"]["v"].bias)
            o = ops.attention(q, k, v, B, nq, L, nh, scale, mfma_bf16=mf)
            tgt = ops.linear(o, W["cross"]["o"].w2d(gd), W["cross"]["o"].bias, residual=tgt)
            t2 = self._ln(tgt, p + ".norm3")
            hdn = ops.linear(t2, W["l1"].w2d(gd), W["l1"].bias, act=ops.ACT_RELU)
            tgt = ops.linear(hdn, W["l2"].w2d(gd), W["l2"].bias, residual=tgt)
        hs = self._ln(tgt, "context2plane_decoder.norm")             # [B*nq, 256]
        mark("ph.decoder")
        return self._heads(features, hs, memory, B, hc, wc, nq, want_logits, mark)

    def _enc_tail_weights(self, i: int) -> dict:
        cache = self.__dict__.setdefault("_enc_tail_w", {})
        if i not in cache:
            p, W = f"context_SA.layers.{i}", self.packed[f"context_SA.layers.{i}"]
            f = lambda k: self.raw(p + k).float().contiguous()
            cache[i] = {"wo": W["attn"]["o"].wfrag(torch.bfloat16), "bo": W["attn"]["o"].bias, "w1": W["l1"].wfrag(torch.bfloat16),
                        "b1": W["l1"].bias, "w2": W["l2"].wfrag(torch.bfloat16), "b2": W["l2"].bias,
                        "g1": f(".norm1.weight"), "be1": f(".norm1.bias"), "g2": f(".norm2.weight"), "be2": f(".norm2.bias")}
        return cache[i]

    def _transformer_bf16(self, src, pos, B, L, nq, nh, scale, mark):
        """bf16-mode encoder/decoder.  Same arithmetic as the fp32-activation path above run in mixed mode - every GEMM / attention
        operand was rounded to bf16 when staged there - but the tensors that only feed GEMMs or attention (q|k, v, attention
        output, FFN hidden, the LayerNorm outputs) are now WRITTEN as bf16, halving their HBM traffic; the residual stream, the
        LayerNorm statistics and all accumulation stay f32.  The decoder's cross-attention K / V of all six layers come from two
        N = 1536 GEMMs over the encoder memory."""
        P, bf, f32 = self.packed, torch.bfloat16, torch.float32
        lin = ops.linear

        def ln(x, prefix, addend=None, want=("y",)):
            return ops.layernorm_ex(x, self.raw(prefix + ".weight"), self.raw(prefix + ".bias"), addend=addend, want=want)

        src16, q_in16 = ops.add_rows_bf16(src, pos)            # src and src + pos as bf16 GEMM operands, one launch
        chain = self.fused_encoder_tail and self.fused_decoder_tail and self.chain_projections
        wg_enc, wg_dec = (B * L + 31) // 32, (B * nq + 31) // 32

        def tail_pf(kind, i):
            """(weight tensors, workgroups) of the tail launch (kind, i): what the launch before it prefetches (ops.transformer_tail)"""
            Wt = self._tail_weights(kind, i)
            tens = [Wt["wo"], Wt.get("w1"), Wt.get("w2")]
            if kind == "enc" and i < 5:
                a = P[f"context_SA.layers.{i + 1}"]["attn"]
                tens += [a["qk"].wfrag(bf), a["v"].wfrag(bf)]
            elif kind == "dec_self":
                tens += [P[f"context2plane_decoder.layers.{i}"]["cross"]["q"].wfrag(bf)]
            elif kind == "dec" and i < 5:
                a = P[f"context2plane_decoder.layers.{i + 1}"]["self"]
                tens += [a["qk"].wfrag(bf), a["v"].wfrag(bf)]
            return tens, (wg_enc if kind == "enc" else wg_dec)
        qk = v = None
        for i in range(6):
            p = f"context_SA.layers.{i}"
            W = P[p]
            if qk is None:
                qk = lin(q_in16, W["attn"]["qk"].w2d(bf), W["attn"]["qk"].bias, out_dtype=bf)
                v = lin(src16, W["attn"]["v"].w2d(bf), W["attn"]["v"].bias, out_dtype=bf)
            o = ops.attention(qk[:, :256], qk[:, 256:], v, B, L, L, nh, scale, mfma_bf16=True)
            qk = v = None
            if chain:      # ... + the NEXT layer's q|k and v projections from the tiles that are still on the chip (one launch per layer)
                Wn = P[f"context_SA.layers.{i + 1}"]["attn"] if i < 5 else None
                r = ops.transformer_tail(o, src, self._tail_weights("enc", i), pre_norm=False, pos=pos, want=("y",),
                                         proj_pos=(Wn["qk"].wfrag(bf), Wn["qk"].bias, 512) if Wn else None,
                                         proj=(Wn["v"].wfrag(bf), Wn["v"].bias, 256) if Wn else None,
                                         prefetch=tail_pf("enc", i + 1) if i < 5 else tail_pf("dec_self", 0))
                src = r["y"]
                if Wn:
                    qk, v = r["proj_pos"], r["proj"]
                continue
            if self.fused_encoder_tail:      # out-proj + LN1 + FFN + LN2 in one launch (csrc/enc_tail.hip)
                r = ops.encoder_tail(o, src, self._enc_tail_weights(i), pos=pos)
                src, src16, q_in16 = r["y"], r["y16"], r["ypos16"]
                continue
            s = lin(o, W["attn"]["o"].w2d(bf), W["attn"]["o"].bias, residual=src, out_dtype=f32)
            r = ln(s, p + ".norm1", want=("y", "y16"))
            src, src16 = r["y"], r["y16"]
            hdn = lin(src16, W["l1"].w2d(bf), W["l1"].bias, act=ops.ACT_RELU, out_dtype=bf)
            s = lin(hdn, W["l2"].w2d(bf), W["l2"].bias, residual=src, out_dtype=f32)
            r = ln(s, p + ".norm2", addend=pos, want=("y", "y16", "y2_16"))
            src, src16, q_in16 = r["y"], r["y16"], r["y2_16"]
        r = ln(src, "context_SA.norm", addend=pos, want=("y", "y16", "y2_16"))
        memory, mem16, memk16 = r["y"], r["y16"], r["y2_16"]
        mark("ph.encoder")
        k_all = lin(memk16, P["cross_k_all"].w2d(bf), P["cross_k_all"].bias, out_dtype=bf)          # [B*L, 6*256]
        v_all = lin(mem16, P["cross_v_all"].w2d(bf), P["cross_v_all"].bias, out_dtype=bf)
        qpos = self.raw("query_embed.weight")
        # the decoder's all-zero start (read-only: every layer writes a new tensor)
        tgt = ops.cached_constant(self.__dict__.setdefault("_zero_tgt", {}), (B * nq, src.device),
                                  lambda: torch.zeros(B * nq, 256, device=src.device, dtype=f32))
        nin = ln(tgt, "context2plane_decoder.layers.0.norm1", addend=qpos, want=("y16", "y2_16"))
        n16, npos16, hs = nin["y16"], nin["y2_16"], None
        qk = v = None
        for i in range(6):
            p = f"context2plane_decoder.layers.{i}"
            W = P[p]
            if qk is None:
                qk = lin(npos16, W["self"]["qk"].w2d(bf), W["self"]["qk"].bias, out_dtype=bf)
                v = lin(n16, W["self"]["v"].w2d(bf), W["self"]["v"].bias, out_dtype=bf)
            o = ops.attention(qk[:, :256], qk[:, 256:], v, B, nq, nq, nh, scale, mfma_bf16=True)
            qk = v = None
            if chain:      # self out-proj + residual + norm2 + the cross-attention's q projection: one launch
                r = ops.transformer_tail(o, tgt, self._tail_weights("dec_self", i), pre_norm=True, skip_ffn=True, pos=qpos, want=("y",),
                                         proj_pos=(W["cross"]["q"].wfrag(bf), W["cross"]["q"].bias, 256), prefetch=tail_pf("dec", i))
                tgt, q = r["y"], r["proj_pos"]
            else:
                tgt = lin(o, W["self"]["o"].w2d(bf), W["self"]["o"].bias, residual=tgt, out_dtype=f32)
                r = ln(tgt, p + ".norm2", addend=qpos, want=("y2_16",))
                q = lin(r["y2_16"], W["cross"]["q"].w2d(bf), W["cross"]["q"].bias, out_dtype=bf)
            o = ops.attention(q, k_all[:, 256 * i:256 * (i + 1)], v_all[:, 256 * i:256 * (i + 1)], B, nq, L, nh, scale, mfma_bf16=True)
            nxt = f"context2plane_decoder.layers.{i + 1}.norm1" if i < 5 else "context2plane_decoder.norm"
            if chain and i < 5:      # cross out-proj + LN3 + FFN + the next norm + the next layer's self q|k and v projections
                Wn = P[f"context2plane_decoder.layers.{i + 1}"]["self"]
                r = ops.transformer_tail(o, tgt, self._tail_weights("dec", i), pre_norm=True, pos=qpos, want=("y",),
                                         proj_pos=(Wn["qk"].wfrag(bf), Wn["qk"].bias, 512), proj=(Wn["v"].wfrag(bf), Wn["v"].bias, 256),
                                         prefetch=tail_pf("dec_self", i + 1))
                tgt, qk, v = r["y"], r["proj_pos"], r["proj"]
                continue
            if self.fused_decoder_tail:      # cross out-proj + LN3 + FFN + the next norm in one launch (csrc/enc_tail.hip)
                r = ops.decoder_tail(o, tgt, self._dec_tail_weights(i, nxt), pos=qpos,
                                     want=("y", "y16", "ypos16") if i < 5 else ("yn",))
                if i < 5:
                    tgt, n16, npos16 = r["y"], r["y16"], r["ypos16"]
                else:
                    hs = r["yn"]
                continue
            tgt = lin(o, W["cross"]["o"].w2d(bf), W["cross"]["o"].bias, residual=tgt, out_dtype=f32)
            r = ln(tgt, p + ".norm3", want=("y16",))
            hdn = lin(r["y16"], W["l1"].w2d(bf), W["l1"].bias, act=ops.ACT_RELU, out_dtype=bf)
            tgt = lin(hdn, W["l2"].w2d(bf), W["l2"].bias, residual=tgt, out_dtype=f32)
            if i < 5:
                nin = ln(tgt, nxt, addend=qpos, want=("y16", "y2_16"))
                n16, npos16 = nin["y16"], nin["y2_16"]
        if hs is None:
            hs = self._ln(tgt, "context2plane_decoder.norm")
        return hs, memory

    def _tail_weights(self, kind: str, i: int) -> dict:
        """Operands of ops.transformer_tail: "enc" = encoder layer i, "dec" = decoder layer i after the cross-attention (second norm = the
        next layer's norm1), "dec_self" = decoder layer i after the self-attention (out-proj + norm2 only)."""
        cache = self.__dict__.setdefault("_tail_w", {})
        if (kind, i) not in cache:
            bf = torch.bfloat16
            f = lambda k: self.raw(k).float().contiguous()
            if kind == "enc":
                e = self._enc_tail_weights(i)
                cache[(kind, i)] = dict(e, ga=e["g1"], bea=e["be1"], gb=e["g2"], beb=e["be2"])
            elif kind == "dec":
                d = self._dec_tail_weights(i, f"context2plane_decoder.layers.{i + 1}.norm1" if i < 5 else "context2plane_decoder.norm")
                cache[(kind, i)] = dict(d, ga=d["g3"], bea=d["be3"], gb=d["gn"], beb=d["ben"])
            else:
                p, W = f"context2plane_decoder.layers.{i}", self.packed[f"context2plane_decoder.layers.{i}"]
                cache[(kind, i)] = {"wo": W["self"]["o"].wfrag(bf), "bo": W["self"]["o"].bias, "ga": f(p + ".norm2.weight"), "bea": f(p + ".norm2.bias")}
        return cache[(kind, i)]

    def _dec_tail_weights(self, i: int, next_norm: str) -> dict:
        cache = self.__dict__.setdefault("_dec_tail_w", {})
        if i not in cache:
            p, W = f"context2plane_decoder.layers.{i}", self.packed[f"context2plane_decoder.layers.{i}"]
            f = lambda k: self.raw(k).float().contiguous()
            cache[i] = {"wo": W["cross"]["o"].wfrag(torch.bfloat16), "bo": W["cross"]["o"].bias, "w1": W["l1"].wfrag(torch.bfloat16),
                        "b1": W["l1"].bias, "w2": W["l2"].wfrag(torch.bfloat16), "b2": W["l2"].bias,
                        "g3": f(p + ".norm3.weight"), "be3": f(p + ".norm3.bias"),
                        "gn": f(next_norm + ".weight"), "ben": f(next_norm + ".bias")}
        return cache[i]

    def _heads(self, features, hs, memory, B, hc, wc, nq, want_logits, mark):
        P, gd = self.packed, self.gemm_dtype
        c1, c2, c3, c4 = features["res2"], features["res3"], features["res4"], features["res5"]
        cd = c4.dtype
        # ---- top-down pyramid (planeTR_head.py:241-252): lateral + relu(bn(conv(up(.))))
        RA = ops.ACT_RELU | ops.ACT_RES_AFTER

        def cbr(x, nm, residual=None, out_dtype=None):
            c = P[nm]
            return ops.conv2d(x, c.w(gd if x.dtype == torch.float32 else x.dtype), c.scale, c.bias, residual,
                              act=RA if residual is not None else ops.ACT_RELU, out_dtype=out_dtype)

        p4 = cbr(memory.view(B, hc, wc, 256), "m_conv_dict.m4", residual=cbr(c4, "c4_conv"), out_dtype=cd)

        def up_stage(x, nm, lateral):
            """relu(bn(conv1x1(up2(x)))) + lateral, evaluated as relu(up2(bn(conv1x1(x)))) + lateral: the 1x1 conv and
            the BN affine are linear and the bilinear weights sum to 1, so they commute with the up-sampling
            exactly in real arithmetic (fp rounding only) - 4x fewer conv FLOPs and bytes (planeTR_head.py:246-250)."""
            c = P[nm]
            t = ops.conv2d(x, c.w(x.dtype), c.scale, c.bias, act=ops.ACT_NONE)
            return ops.upsample2x_bilinear(t, lateral, act=ops.ACT_RELU)

        p3 = up_stage(p4, "up_conv3", cbr(c3, "c3_conv"))
        p2 = up_stage(p3, "up_conv2", cbr(c2, "c2_conv"))
        # plane embedding MLP [B*nq, 256] -> folded mask-head operands [B*nq, 264]: mask weights | bias | pad
        # ... and the three small heads on the same rows: five stacks, one launch in bf16 GEMM mode
        if gd == torch.bfloat16 and hs.dtype == torch.float32:
            r = run_stacks(hs, [(P["plane_embedding"], ops.ACT_NONE, None), ([P["pe_fold"]], ops.ACT_NONE, True),
                                ([P["plane_prob"]], ops.ACT_NONE, True), (P["plane_param"], ops.ACT_NONE, True),
                                (P["plane_center"], ops.ACT_SIGMOID, True)], gd, parallel=[False, False, True, True, True])
            fold = r[1]
            heads = {"pred_logits": r[2].view(B, nq, 2), "pred_params": r[3].view(B, nq, 3), "pred_centers": r[4].view(B, nq, 2)}
        else:
            fold = run_stacks(hs, [(P["plane_embedding"], ops.ACT_NONE, None), ([P["pe_fold"]], ops.ACT_NONE, True)], gd)[1]
            heads = {
                "pred_logits": run_mlp(hs, [P["plane_prob"]], gd=gd).view(B, nq, 2),
                "pred_params": run_mlp(hs, P["plane_param"], gd=gd).view(B, nq, 3),
                "pred_centers": run_mlp(hs, P["plane_center"], final_act=ops.ACT_SIGMOID, gd=gd).view(B, nq, 2),
            }
        h1, w1 = c1.shape[1], c1.shape[2]
        if (self.fused_mask_head and cd == torch.bfloat16 and not want_logits and nq <= 128 and nq % 2 == 0 and (h1 * w1) % 128 == 0):
            # finest lateral conv + bilinear add + mask GEMM in one launch: p1 never goes to HBM (csrc/mask_head.hip)
            c = P["up_conv1"]
            t1 = ops.conv2d(p2, c.w(cd), c.scale, c.bias, act=ops.ACT_NONE)
            l = P["c1_conv"]
            mark("ph.top_down")
            # (a planar [B,nq,h,w] output + valid-planes-only fetch in the post-selection is implemented and tested, but measured
            # slower - 581 vs 437 us: that kernel is bound by its per-(pixel, valid query) instruction stream, not by bytes)
            heads["mask_prob"] = ops.mask_head(c1, t1, l.wfrag(cd), l.scale, l.bias, None, None, fold=fold)
            return heads, hs.view(B, nq, 256)
        p1 = up_stage(p2, "up_conv1", cbr(c1, "c1_conv"))
        mark("ph.top_down")
        # ---- instance heads
        mw = fold[:, :256].to(cd).contiguous().view(B, nq, 1, 1, 256)
        mb = fold[:, 256].contiguous().view(B, nq)
        out = dict(heads)
        out["mask_prob"] = ops.conv2d(p1, mw, None, mb, batched_weights=True, act=ops.ACT_SIGMOID, out_dtype=torch.float32)
        if want_logits:
            out["pred_mask_logits"] = ops.conv2d(p1, mw, None, mb, batched_weights=True, out_dtype=torch.float32)
            pc = P["pixel_plane_center"]
            out["pixel_centers"] = ops.conv2d(p1, pc.w(cd), None, pc.bias, act=ops.ACT_SIGMOID, out_dtype=torch.float32)
        return out, hs.view(B, nq, 256)


def build_planeTR_head(cfg, input_shape=None):
    return SEM_SEG_HEADS_REGISTRY.get(cfg.MODEL.SEM_SEG_HEAD.NAME)(cfg, input_shape)


def post_select(head_out: dict, query_feat: torch.Tensor, height: int, width: int, cfg) -> dict:
    """Fused plane post-selection for a batch of images (siamese_planeTR.py:625-803); thresholds from
    cfg.TEST (config/config.py:92-94).  Everything stays on the device (no per-plane host sync)."""
    return ops.postselect_planes(head_out["pred_logits"], head_out["mask_prob"], head_out["pred_params"], query_feat,
                                 height, width, float(cfg.TEST.PLANE_SCORE_THRESHOLD), float(cfg.TEST.MASK_PROB_THRESHOLD),
                                 float(cfg.TEST.OVERLAP_THRESHOLD), planar=bool(head_out.get("mask_prob_planar", False)))
