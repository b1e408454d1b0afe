"""Training step of the camera head's refinement stage on MI355X (reference: `__forward_PlaneCamRefHead`,
camera_net/camera_head.py:737-923; losses camera_modules.py:355-365; optimiser groups train_NopeSAC.py:88-169) - round 5.

Forward = the kernels of the training-side twin (ops.geo_sequence, ops.ransac_score_maps, ops.ransac_soft_vote mode | 16,
ops.plane_cam_ref_losses) with the MLP stacks as one f32 GEMM launch per layer (every layer output is kept for the backward pass);
backward = the hand-written vector-Jacobian kernels of csrc/refine_bwd.hip (losses, scoring + aggregation + pose heads, hypothesis x
plane geometry) and, for every Linear layer, dgrad = dY W and wgrad = dY^T X on the library's f32 GEMM kernel + column sums for the
bias + the ReLU mask.  `torch.autograd.Function` only does the bookkeeping (which gradient goes where); concatenations and the
broadcast of the initial-pose features are torch views / a [B, nq, 256] sum.  Everything is f32 and deterministic.

What this is NOT: a trainer for the whole network.  The backbone, the plane head, the matcher and the pixel pose net have no backward
kernels here - their outputs (the initial pose, its 256-d features, the plane sets and the assignment) are inputs of this stage, as they
are of the reference function, and receive no gradient beyond `input_grads`.  Gated against torch.autograd on the oracle
(tests/test_training_gpu.py)."""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from . import _lib, ops

PREFIX = "camera_head_list.0."
MLPS = ("geo_encoder", "geo_proj_s1", "decoder_rot", "geo_proj_s2", "decoder_tran", "decoder_rot2", "decoder_tran2", "normal_score_proj",
        "param_score_proj")
LINEARS = ("rots", "trans", "rot_score_reg", "trans_score_reg")


def _L():
    return _lib.load()


def _p(t):
    return None if t is None else t.data_ptr()


def _st():
    return torch.cuda.current_stream().cuda_stream


def transpose(x: torch.Tensor) -> torch.Tensor:
    """[rows, cols] f32 (rows may be strided) -> contiguous [cols, rows]."""
    assert x.dim() == 2 and x.dtype == torch.float32 and x.stride(1) == 1
    y = torch.empty(x.shape[1], x.shape[0], device=x.device, dtype=torch.float32)
    _lib.check(_L().nopesac_transpose_f32(_p(x), x.shape[0], x.shape[1], x.stride(0), _p(y), _st()), "nopesac_transpose_f32")
    return y


def col_sum(x: torch.Tensor) -> torch.Tensor:
    assert x.dim() == 2 and x.dtype == torch.float32 and x.stride(1) == 1
    out = torch.empty(x.shape[1], device=x.device, dtype=torch.float32)
    _lib.check(_L().nopesac_col_sum_f32(_p(x), x.shape[0], x.shape[1], x.stride(0), _p(out), _st()), "nopesac_col_sum_f32")
    return out


class _Linear(torch.autograd.Function):
    """y = act(x W^T + b) on the exact-f32 GEMM kernel; backward: dX = dY W, dW = dY^T X (the same kernel), db = column sums."""

    @staticmethod
    def forward(ctx, x, w, b, relu: bool):
        x = x.contiguous()
        y = ops.linear(x, w, b, act=ops.ACT_RELU if relu else ops.ACT_NONE)
        ctx.relu = relu
        ctx.save_for_backward(x, w, y)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w, y = ctx.saved_tensors
        g = g.contiguous()
        if ctx.relu:
            gm = torch.empty_like(g)
            _lib.check(_L().nopesac_relu_backward_f32(_p(g), _p(y), g.numel(), _p(gm), _st()), "nopesac_relu_backward_f32")
            g = gm
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = ops.linear(g, transpose(w))                      # [rows, N] x [N, K]
        if ctx.needs_input_grad[1]:
            gw = ops.linear(transpose(g), transpose(x))           # [N, rows] x [rows, K]
        if ctx.needs_input_grad[2]:
            gb = col_sum(g)
        return gx, gw, gb, None


class _ScoreMaps(torch.autograd.Function):
    @staticmethod
    def forward(ctx, geo_local, rot_raw, trans_raw, init_rot, init_trans, m):
        out = ops.ransac_score_maps(geo_local, rot_raw.contiguous(), trans_raw.contiguous(), init_rot, init_trans, m, diagnostics=True)
        ctx.save_for_backward(geo_local, rot_raw, trans_raw, init_rot, init_trans, m)
        # (rots_all / trans_all only pick the hypothesis of the index losses; the row sums feed the inference-only 'min-cost' mode)
        ctx.mark_non_differentiable(out["rots_all"], out["trans_all"], out["dn_sum"], out["dl2_sum"])
        return out["normal_score"], out["param_score"], out["l2_dist"], out["rots_all"], out["trans_all"], out["dn_sum"], out["dl2_sum"]

    @staticmethod
    def backward(ctx, g_ns, g_ps, g_l2, _a, _b, _c, _d):
        geo_local, rot_raw, trans_raw, init_rot, init_trans, m = ctx.saved_tensors
        B, nq, _ = geo_local.shape
        z = lambda t, ref: torch.zeros_like(ref) if t is None else t.contiguous()
        ref = torch.empty(B, nq + 1, nq, device=geo_local.device, dtype=torch.float32)
        g_ns, g_ps, g_l2 = (z(t, ref) for t in (g_ns, g_ps, g_l2))
        g_rot, g_tr = torch.empty_like(rot_raw), torch.empty_like(trans_raw)
        g_ir, g_it = torch.empty_like(init_rot), torch.empty_like(init_trans)
        rc = _L().nopesac_refine_score_maps_backward(_p(geo_local), _p(rot_raw.contiguous()), _p(trans_raw.contiguous()), _p(init_rot), _p(init_trans),
                                                     _p(m), B, nq, _p(g_ns), _p(g_ps), _p(g_l2), _p(g_rot), _p(g_tr), _p(g_ir), _p(g_it), _st())
        _lib.check(rc, "nopesac_refine_score_maps_backward")
        return None, g_rot, g_tr, g_ir, g_it, None


class _Vote(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sf_rot, sf_trans, reg_rot_w, reg_rot_b, reg_trans_w, reg_trans_b, init_rot_feat, init_trans_feat, fused_rot, fused_trans,
                rots_w, rots_b, trans_w, trans_b, rots_all, trans_all, dn_sum, dl2_sum, init_rot, init_trans, m):
        args = [t.contiguous() for t in (sf_rot, sf_trans, reg_rot_w, reg_rot_b, reg_trans_w, reg_trans_b, init_rot_feat, init_trans_feat, fused_rot,
                                         fused_trans, rots_w, rots_b, trans_w, trans_b)]
        maps = {"rots_all": rots_all, "trans_all": trans_all, "dn_sum": dn_sum, "dl2_sum": dl2_sum}
        out = ops.ransac_soft_vote(*args, maps, init_rot, init_trans, m, 16)
        ctx.save_for_backward(*args, m)
        return out["pred_rot"], out["pred_trans"], out["avg_rot"], out["avg_trans"], out["score_rot"], out["score_trans"]

    @staticmethod
    def backward(ctx, g_pr, g_pt, g_ar, g_at, g_sr, g_st):
        (sf_rot, sf_trans, reg_rot_w, reg_rot_b, reg_trans_w, reg_trans_b, init_rot_feat, init_trans_feat, fused_rot, fused_trans, rots_w, rots_b,
         trans_w, trans_b, m) = ctx.saved_tensors
        B, NH, _ = sf_rot.shape
        nq = NH - 1
        dev = sf_rot.device
        f32 = dict(device=dev, dtype=torch.float32)
        zl = lambda t, shape: torch.zeros(shape, **f32) if t is None else t.contiguous()
        g_pr, g_ar = zl(g_pr, (B, 4)), zl(g_ar, (B, 4))
        g_pt, g_at = zl(g_pt, (B, 3)), zl(g_at, (B, 3))
        g_sr, g_st = zl(g_sr, (B, NH)), zl(g_st, (B, NH))
        o = {"g_sf_rot": torch.empty_like(sf_rot), "g_sf_trans": torch.empty_like(sf_trans), "g_irf": torch.empty_like(init_rot_feat),
             "g_itf": torch.empty_like(init_trans_feat), "g_fr": torch.empty_like(fused_rot), "g_ft": torch.empty_like(fused_trans),
             "pb_rw": torch.empty(B, 4 * 256, **f32), "pb_rb": torch.empty(B, 4, **f32), "pb_tw": torch.empty(B, 3 * 256, **f32),
             "pb_tb": torch.empty(B, 3, **f32), "pb_rrw": torch.empty(B, 64, **f32), "pb_rrb": torch.empty(B, 1, **f32),
             "pb_rtw": torch.empty(B, 64, **f32), "pb_rtb": torch.empty(B, 1, **f32)}
        rc = _L().nopesac_refine_vote_backward(
            _p(sf_rot), _p(sf_trans), _p(reg_rot_w), _p(reg_rot_b), _p(reg_trans_w), _p(reg_trans_b), _p(init_rot_feat), _p(init_trans_feat), _p(fused_rot),
            _p(fused_trans), _p(rots_w), _p(rots_b), _p(trans_w), _p(trans_b), _p(m), B, nq, _p(g_pr), _p(g_pt), _p(g_ar), _p(g_at), _p(g_sr), _p(g_st),
            _p(o["g_sf_rot"]), _p(o["g_sf_trans"]), _p(o["g_irf"]), _p(o["g_itf"]), _p(o["g_fr"]), _p(o["g_ft"]), _p(o["pb_rw"]), _p(o["pb_rb"]),
            _p(o["pb_tw"]), _p(o["pb_tb"]), _p(o["pb_rrw"]), _p(o["pb_rrb"]), _p(o["pb_rtw"]), _p(o["pb_rtb"]), _st())
        _lib.check(rc, "nopesac_refine_vote_backward")
        red = lambda t, like: col_sum(t).view_as(like)              # per-pair partials -> the parameter's gradient (fixed order)
        return (o["g_sf_rot"], o["g_sf_trans"], red(o["pb_rrw"], reg_rot_w), red(o["pb_rrb"], reg_rot_b), red(o["pb_rtw"], reg_trans_w),
                red(o["pb_rtb"], reg_trans_b), o["g_irf"], o["g_itf"], o["g_fr"], o["g_ft"], red(o["pb_rw"], rots_w), red(o["pb_rb"], rots_b),
                red(o["pb_tw"], trans_w), red(o["pb_tb"], trans_b), None, None, None, None, None, None, None)


class _Losses(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred_rot, pred_trans, avg_rot, avg_trans, score_rot, score_trans, l2_dist, rots_all, trans_all, m, gt_pose, weight: float):
        vote = {"pred_rot": pred_rot.contiguous(), "pred_trans": pred_trans.contiguous(), "avg_rot": avg_rot.contiguous(), "avg_trans": avg_trans.contiguous(),
                "score_rot": score_rot.contiguous(), "score_trans": score_trans.contiguous()}
        maps = {"rots_all": rots_all, "trans_all": trans_all, "l2_dist": l2_dist.contiguous()}
        ctx.weight = float(weight)
        ctx.save_for_backward(vote["pred_rot"], vote["pred_trans"], vote["avg_rot"], vote["avg_trans"], vote["score_rot"], vote["score_trans"], rots_all,
                              trans_all, m, gt_pose)
        ctx.l2_shape = tuple(l2_dist.shape)
        return ops.plane_cam_ref_losses(vote, maps, m, gt_pose, weight)

    @staticmethod
    def backward(ctx, g):
        pr, pt, ar, at, sr, st, rots_all, trans_all, m, gt = ctx.saved_tensors
        B, NH = sr.shape
        dev = sr.device
        f32 = dict(device=dev, dtype=torch.float32)
        o = [torch.empty(B, 4, **f32), torch.empty(B, 3, **f32), torch.empty(B, 4, **f32), torch.empty(B, 3, **f32), torch.empty(B, NH, **f32),
             torch.empty(B, NH, **f32), torch.empty(ctx.l2_shape, **f32)]
        rc = _L().nopesac_refine_losses_backward(_p(pr), _p(pt), _p(ar), _p(at), _p(rots_all), _p(trans_all), _p(sr), _p(st), _p(m), _p(gt),
                                                 _p(g.contiguous()), B, NH - 1, ctx.weight, *[_p(t) for t in o], _st())
        _lib.check(rc, "nopesac_refine_losses_backward")
        return o[0], o[1], o[2], o[3], o[4], o[5], o[6], None, None, None, None, None


class _Normalize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, canonical: bool):
        x = x.contiguous()
        ctx.canonical = bool(canonical)
        ctx.save_for_backward(x)
        return ops.normalize_rows(x, canonical_sign=ctx.canonical)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        out = torch.empty_like(x)
        D = x.shape[-1]
        rc = _L().nopesac_normalize_rows_backward(_p(x), _p(g.contiguous()), x.numel() // D, D, int(ctx.canonical), _p(out), _st())
        _lib.check(rc, "nopesac_normalize_rows_backward")
        return out, None


class _PoseLoss(torch.autograd.Function):
    """CameraPoseLoss / the AIM's reconstruction losses (ops.camera_pose_loss) -> f32[2] = (l_x, l_q) * weight."""

    @staticmethod
    def forward(ctx, est_t, est_q, gt_t, gt_q, weight: float, trans_eps: float):
        est_t, est_q, gt_t, gt_q = (t.contiguous() for t in (est_t, est_q, gt_t, gt_q))
        ctx.weight, ctx.eps = float(weight), float(trans_eps)
        ctx.save_for_backward(est_t, est_q, gt_t, gt_q)
        return ops.camera_pose_loss(est_t, est_q, gt_t, gt_q, weight, trans_eps)

    @staticmethod
    def backward(ctx, g):
        est_t, est_q, gt_t, gt_q = ctx.saved_tensors
        B = est_t.shape[0]
        o = [torch.empty_like(est_t), torch.empty_like(est_q), torch.empty_like(gt_t), torch.empty_like(gt_q)]
        rc = _L().nopesac_camera_pose_loss_backward(_p(est_t), _p(est_q), _p(gt_t), 3, _p(gt_q), 4, B, ctx.eps, ctx.weight, _p(g.contiguous()),
                                                    *[_p(t) for t in o], _st())
        _lib.check(rc, "nopesac_camera_pose_loss_backward")
        return o[0], o[1], o[2], o[3], None, None


class RefineTrainer:
    """The refinement head's parameters as f32 leaf tensors + forward / backward / optimiser step.

        tr = RefineTrainer.from_state_dict(sd, nq, device)             # or .from_head(model.camera_head_list[0])
        losses = tr.losses(A, planes1, planes2, n1, n2, init_trans, init_rot, init_trans_feat, init_rot_feat, gt_pose, suffix, weight)
        grads = tr.backward(losses)                                     # {state-dict key: gradient}; tr.input_grads for the feature inputs
        tr.step(lr=1e-4)                                                # AdamW (train_NopeSAC.py:150-153) on the device
        tr.write_back(head)                                             # into the inference model's packed weights
    """

    def __init__(self, params: Dict[str, torch.Tensor], nq: int, warp_in_ref: bool = True):
        self.nq, self.warp_in_ref = int(nq), bool(warp_in_ref)
        self.params = {k: v.detach().clone().float().contiguous().requires_grad_(True) for k, v in params.items()}
        self.state: Dict[str, dict] = {}
        self.steps = 0
        self.input_grads: Dict[str, torch.Tensor] = {}

    @staticmethod
    def parameter_names(sd_keys) -> List[str]:
        out = []
        for k in sd_keys:
            if not k.startswith(PREFIX):
                continue
            r = k[len(PREFIX):]
            if r.split(".")[0] in MLPS + LINEARS:
                out.append(k)
        return sorted(out)

    @classmethod
    def from_state_dict(cls, sd: dict, nq: int, device, warp_in_ref: bool = True) -> "RefineTrainer":
        return cls({k: sd[k].to(device) for k in cls.parameter_names(sd.keys())}, nq, warp_in_ref)

    @classmethod
    def from_head(cls, head) -> "RefineTrainer":
        names = cls.parameter_names(PREFIX + k for k in head._spec_keys)
        return cls({k: head.raw(k[len(PREFIX):]) for k in names}, head.num_queries, head.warp_plane_in_cam_ref_on)

    # ---- forward
    def _mlp(self, x, name: str, final_relu: bool = False):
        i = 0
        while f"{PREFIX}{name}.layers.{i + 1}.weight" in self.params:
            x = _Linear.apply(x, self.params[f"{PREFIX}{name}.layers.{i}.weight"], self.params[f"{PREFIX}{name}.layers.{i}.bias"], True)
            i += 1
        return _Linear.apply(x, self.params[f"{PREFIX}{name}.layers.{i}.weight"], self.params[f"{PREFIX}{name}.layers.{i}.bias"], final_relu)

    def _lin(self, x, name: str):
        return _Linear.apply(x, self.params[f"{PREFIX}{name}.weight"], self.params[f"{PREFIX}{name}.bias"], False)

    def losses(self, A0, planes1, planes2, n1, n2, init_trans, init_rot, init_trans_feat, init_rot_feat, gt_pose, suffix: str = "",
               weight: float = 1.0) -> Dict[str, torch.Tensor]:
        """The seven losses of __forward_PlaneCamRefHead (camera_head.py:883-921) with the autograd tape attached.  The feature / pose inputs
        may require grad (their gradients land in `input_grads` after backward())."""
        P, nq = self.params, self.nq
        B = A0.shape[0]
        with torch.no_grad():
            geo_local, _gg, _sig, geo_enc, m = ops.geo_sequence(A0, planes1, planes2, n1, n2, init_trans.detach(), init_rot.detach(), self.warp_in_ref)
        rows = B * nq
        geo = self._mlp(geo_enc.view(rows, 8), "geo_encoder")
        s1 = self._mlp(geo, "geo_proj_s1")
        f_rot = self._mlp(s1, "decoder_rot")
        s2 = self._mlp(torch.cat([s1, f_rot], dim=1), "geo_proj_s2")
        f_tran = self._mlp(s2, "decoder_tran")
        bc = lambda f: f.unsqueeze(1).expand(B, nq, f.shape[1]).reshape(rows, f.shape[1])
        fused_rot = self._mlp(torch.cat([bc(init_rot_feat), f_rot], dim=1), "decoder_rot2", final_relu=True)       # :980-983 (+ F.relu)
        fused_tran = self._mlp(torch.cat([bc(init_trans_feat), f_tran], dim=1), "decoder_tran2", final_relu=True)
        rot_raw = self._lin(fused_rot, "rots").view(B, nq, 4)
        trans_raw = self._lin(fused_tran, "trans").view(B, nq, 3)
        ns, ps, l2, rots_all, trans_all, dn_sum, dl2_sum = _ScoreMaps.apply(geo_local, rot_raw, trans_raw, init_rot, init_trans, m)
        NH = nq + 1
        sf_rot = self._mlp(ns.view(B * NH, nq), "normal_score_proj").view(B, NH, 64)
        sf_tran = self._mlp(ps.view(B * NH, nq), "param_score_proj").view(B, NH, 64)
        w = lambda n: P[f"{PREFIX}{n}.weight"]
        bb = lambda n: P[f"{PREFIX}{n}.bias"]
        pr, pt, ar, at, sr, st = _Vote.apply(sf_rot, sf_tran, w("rot_score_reg").view(-1), bb("rot_score_reg"), w("trans_score_reg").view(-1),
                                             bb("trans_score_reg"), init_rot_feat, init_trans_feat, fused_rot.view(B, nq, 256),
                                             fused_tran.view(B, nq, 256), w("rots"), bb("rots"), w("trans"), bb("trans"), rots_all, trans_all,
                                             dn_sum, dl2_sum, init_rot.detach(), init_trans.detach(), m)
        lv = _Losses.apply(pr, pt, ar, at, sr, st, l2, rots_all, trans_all, m, gt_pose, float(weight))
        self._inputs = {"init_trans_feat": init_trans_feat, "init_rot_feat": init_rot_feat, "init_trans": init_trans, "init_rot": init_rot}
        for t in self._inputs.values():                    # a caller may chain its own graph in front (CameraHeadTrainer does): non-leaf
            if t.requires_grad and not t.is_leaf:          # inputs keep their gradient only on request, and `input_grads` promises it
                t.retain_grad()
        self.last = {"pred_rot": pr, "pred_trans": pt, "avg_rot": ar, "avg_trans": at, "score_rot": sr, "score_trans": st, "m": m}
        return {"%s_%s" % (nm, suffix): lv[i] for i, nm in enumerate(ops.PLANE_CAM_REF_LOSS_NAMES)}

    # ---- backward
    def backward(self, losses: Dict[str, torch.Tensor], loss_weights: Optional[Dict[str, float]] = None) -> Dict[str, torch.Tensor]:
        """d (sum of the losses, optionally weighted) / d parameters -> {state-dict key: gradient}."""
        for p in self.params.values():
            p.grad = None
        for t in self._inputs.values():
            if t.requires_grad and t.is_leaf:
                t.grad = None
        total = None
        for k, v in losses.items():
            term = v * float(loss_weights.get(k, 1.0)) if loss_weights else v
            total = term if total is None else total + term
        total.backward()
        self.input_grads = {k: t.grad for k, t in self._inputs.items() if t.requires_grad and t.grad is not None}
        return {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in self.params.items()}

    # ---- optimiser (train_NopeSAC.py:88-169: AdamW / SGD over per-parameter groups; norm / embedding groups do not occur in this head)
    def step(self, lr: float = 1e-4, optimizer: str = "ADAMW", weight_decay: float = 0.01, betas=(0.9, 0.999), eps: float = 1e-8,
             momentum: float = 0.9):
        self.steps += 1
        for k, p in self.params.items():
            if p.grad is None:
                continue
            g = p.grad.contiguous()
            st = self.state.setdefault(k, {})
            with torch.no_grad():
                if optimizer.upper() == "ADAMW":
                    if not st:
                        st["m1"], st["m2"] = torch.zeros_like(p), torch.zeros_like(p)
                    rc = _L().nopesac_adamw_step(_p(p), _p(g), _p(st["m1"]), _p(st["m2"]), p.numel(), float(lr), float(betas[0]), float(betas[1]),
                                                 float(eps), float(weight_decay), self.steps, _st())
                    _lib.check(rc, "nopesac_adamw_step")
                elif optimizer.upper() == "SGD":
                    first = "mom" not in st
                    if first:
                        st["mom"] = torch.zeros_like(p)
                    rc = _L().nopesac_sgd_step(_p(p), _p(g), _p(st["mom"]), p.numel(), float(lr), float(momentum), float(weight_decay), int(first), _st())
                    _lib.check(rc, "nopesac_sgd_step")
                else:
                    raise NotImplementedError(f"no optimizer type {optimizer}")           # train_NopeSAC.py:158

    def clip_grad_norm(self, max_norm: float) -> torch.Tensor:
        """torch.nn.utils.clip_grad_norm_ over ALL parameters of this trainer (the reference's FullModelGradientClippingOptimizer,
        train_NopeSAC.py:139-148), on the device: returns the clip coefficient (f32[1]; 1 = not clipped) without a host sync."""
        grads = [p.grad for p in self.params.values() if p.grad is not None]
        acc = torch.zeros(1, device=grads[0].device, dtype=torch.float32)
        coef = torch.empty(1, device=grads[0].device, dtype=torch.float32)
        for g in grads:
            _lib.check(_L().nopesac_sumsq_accumulate_f32(_p(g.contiguous()), g.numel(), _p(acc), _st()), "nopesac_sumsq_accumulate_f32")
        _lib.check(_L().nopesac_clip_coefficient(_p(acc), float(max_norm), _p(coef), _st()), "nopesac_clip_coefficient")
        for p in self.params.values():
            if p.grad is not None:
                if not p.grad.is_contiguous():
                    p.grad = p.grad.contiguous()
                _lib.check(_L().nopesac_scale_by_f32(_p(p.grad), p.grad.numel(), _p(coef), _st()), "nopesac_scale_by_f32")
        return coef

    def step_from_cfg(self, cfg):
        """One optimiser step with the reference's solver settings (Trainer.build_optimizer, train_NopeSAC.py:88-169): SOLVER.OPTIMIZER
        (ADAMW / SGD), BASE_LR, WEIGHT_DECAY, MOMENTUM, CLIP_GRADIENTS (CLIP_TYPE "full_model": global-norm clipping in front of the
        step).  The per-module multipliers (BACKBONE / SEM_SEG_HEAD / PLANE_MATCHER_HEAD) and the norm / embedding weight decays apply
        to modules this trainer does not hold; every camera-head parameter uses the defaults, as in the reference."""
        S = cfg.SOLVER
        cg = S.CLIP_GRADIENTS
        if cg.ENABLED and cg.CLIP_TYPE == "full_model" and cg.CLIP_VALUE > 0.0:
            self.clip_grad_norm(float(cg.CLIP_VALUE))
        elif cg.ENABLED:
            raise NotImplementedError("SOLVER.CLIP_GRADIENTS.CLIP_TYPE %r (the reference's configs use full_model)" % (cg.CLIP_TYPE,))
        self.step(lr=float(S.BASE_LR), optimizer=str(S.OPTIMIZER), weight_decay=float(S.WEIGHT_DECAY), momentum=float(S.MOMENTUM))

    def write_back(self, head):
        """Copy the trained parameters into the inference head (its packed / bf16 copies are rebuilt on the next forward)."""
        with torch.no_grad():
            for k, p in self.params.items():
                head.raw(k[len(PREFIX):]).copy_(p)
        head.invalidate()


class CameraHeadTrainer(RefineTrainer):
    """The camera head in TRAINING mode (reference PlaneCameraHead.forward, camera_head.py:140-344) with every Linear layer trainable:
    the pixel pose net's FC layers + pose regressors (fc_trans / fc_rots / trans / rots), the AIM (rot_emb_proj / trans_emb_proj) and the
    refinement head - 108 tensors.  The pixel pose net's CONVOLUTIONS (pixel decoder, correlation stack) have no backward kernels: their
    output features [B, 768] are computed by the inference kernels and enter as constants, i.e. those layers are frozen.
    Detach points as in the reference: the AIM re-embeds a detached copy of the pixel pose (:694, :723); the geometry sequences are built
    from detached initial poses (:354-365)."""

    EXTRA_MLPS = ("rot_emb_proj", "trans_emb_proj")
    EXTRA_LINEARS = ("fc_trans", "fc_rots")

    @staticmethod
    def parameter_names(sd_keys) -> List[str]:
        out = []
        for k in sd_keys:
            if not k.startswith(PREFIX):
                continue
            if k[len(PREFIX):].split(".")[0] in MLPS + LINEARS + CameraHeadTrainer.EXTRA_MLPS + CameraHeadTrainer.EXTRA_LINEARS:
                out.append(k)
        return sorted(out)

    def pixel_pose(self, yt: torch.Tensor, yr: torch.Tensor):
        """The FC layers + regressors on the conv features yt / yr [B, 768] in the REFERENCE's flatten order (channel-major: c * 6 + hw)."""
        trans_feat = self._lin_relu(yt, "fc_trans")
        rots_feat = self._lin_relu(yr, "fc_rots")
        trans0 = self._lin(trans_feat, "trans")
        rot0 = _Normalize.apply(self._lin(rots_feat, "rots"), False)
        return trans0, rot0, trans_feat, rots_feat

    def _lin_relu(self, x, name: str):
        return _Linear.apply(x, self.params[f"{PREFIX}{name}.weight"], self.params[f"{PREFIX}{name}.bias"], True)

    def aim(self, trans_in: torch.Tensor, rot_in: torch.Tensor):
        """__forward_RotRecHead / __forward_TransRecHead (:685-735) on DETACHED inputs; returns (rec_trans, rec_rot, trans_feat, rot_feat,
        the canonical-sign input rotation, the shifted input translation)."""
        rot_c = ops.normalize_rows(rot_in.detach().contiguous(), canonical_sign=True)
        tr_in = (trans_in.detach() + 1e-10).contiguous()
        rot_feat = self._mlp(rot_c, "rot_emb_proj", final_relu=True)
        rec_rot = _Normalize.apply(self._lin(rot_feat, "rots"), False)
        trans_feat = self._mlp(tr_in, "trans_emb_proj", final_relu=True)
        rec_trans = self._lin(trans_feat, "trans")
        return rec_trans, rec_rot, trans_feat, rot_feat, rot_c, tr_in

    def camera_head_losses(self, head, feats: dict, B: int, gt_planes1, gt_planes2, gt_n1, gt_n2, gt_assignment, gt_pose, planes1=None, planes2=None,
                           n1=None, n2=None, assignment=None, rand_rot=None, rand_trans=None) -> Dict[str, torch.Tensor]:
        """All losses of the training-mode forward (the 34 of PlaneCameraHead.forward_train) with the autograd tape attached."""
        with torch.no_grad():
            yt, yr = head.pixel_pose_net(feats, B, features_only=True)          # [B, hw * 128 + c]: the inference kernels' NHWC order
            to_ref = lambda y: y.view(B, 6, 128).transpose(1, 2).reshape(B, 768).contiguous().float()
            yt, yr = to_ref(yt), to_ref(yr)
        self.conv_feats = (yt, yr)                                # (what the frozen conv stacks handed to the trainable layers)
        losses: Dict[str, torch.Tensor] = {}
        trans0, rot0, tf0, rf0 = self.pixel_pose(yt, yr)
        lp = _PoseLoss.apply(trans0, rot0, gt_pose[:, 0:3], gt_pose[:, 3:7], head.initial_cam_weight, 0.0)
        losses["loss_tran_pixelReg"], losses["loss_rot_pixelReg"] = lp[0], lp[1]

        def rec(trans_in, rot_in, suffix):
            rec_t, rec_r, rec_tf, rec_rf, rot_c, tr_in = self.aim(trans_in, rot_in)
            lr = _PoseLoss.apply(rec_t, rec_r, tr_in, rot_c, 1.0, 0.0)           # (tr_in already carries the + 1e-10)
            losses["loss_rot" + suffix], losses["loss_trans" + suffix] = lr[1], lr[0]
            return rec_t, rec_r, rec_tf, rec_rf

        rec_t, rec_r, rec_tf, rec_rf = rec(trans0, rot0, "_initCamRec")
        passes = [("", gt_planes1, gt_planes2, gt_n1, gt_n2, gt_assignment, head.plane_cam_weight)]
        if assignment is not None:
            passes.append(("_Aux", planes1, planes2, n1, n2, assignment, head.plane_cam_weight_predplane))
        self.recorded = {"trans": [trans0, rec_t], "rot": [rot0, rec_r]}
        inputs = {}
        for sfx, pl1, pl2, c1, c2, A, w in passes:
            for name, it, ir, itf, irf in (("initCamRef", trans0, rot0, tf0, rf0), ("initRecCamRef", rec_t, rec_r, rec_tf, rec_rf)):
                losses.update(self.losses(A, pl1, pl2, c1, c2, it, ir, itf, irf, gt_pose, suffix=name + sfx, weight=w))
                self.recorded["trans"] += [self.last["avg_trans"], self.last["pred_trans"]]
                self.recorded["rot"] += [self.last["avg_rot"], self.last["pred_rot"]]
        if rand_rot is not None:
            rec(rand_trans, rand_rot, "_randCamRecLBS_N1")
        self._inputs = inputs
        return losses
