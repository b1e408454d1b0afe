"""CAMERA_HEAD on HIP kernels (camera_net/camera_head.py:400-640 `inference_Joint`, batched over pairs):

  (i)   pixel pose-regression net: pixel decoder (GN) -> 6 conv/BN/LeakyReLU -> 300x300 correlation softmax
        -> 2 x 6 strided convs -> FC -> initial pose                                (:642-683, camera_modules.py:246-348)
  (ii)  AIM re-embedding MLPs                                                        (:685-735)
  (iii) plane matching (matching_head.MatchingHead) + mutual-NN assignment            (:493-501)
  (iv)  neural one-plane RANSAC: geo encoding -> per-plane pose hypotheses (MFMA MLP stacks) -> (K+1)xK
        scoring maps -> score MLPs -> masked softmax + soft aggregation -> final pose  (:512-583, 925-1115)
  (v)   geometric re-filter of the assignment                                         (:605-629)

The reference asserts batch == 1 pair; here every stage runs on B pairs at once with per-pair plane / match
counts as device int32 vectors, and no stage synchronises with the host.
"""
from __future__ import annotations

import os

import torch

from .. import ops
from ..registry import CAMERA_HEAD_REGISTRY
from ..synth import state_dict_spec
from .params import ConvW, ParamModule, conv_bias, conv_bn, mlp_layers
from .plane_head import run_mlp, run_stacks

CAM_MODES = {"soft": 0, "avg-all": 1, "min-cost": 2, "max-score": 3}


@CAMERA_HEAD_REGISTRY.register()
class PlaneCameraHead(ParamModule):
    CORR_PAD = 304      # 15 x 20 = 300 correlation channels at 480 x 640, padded to a multiple of 8
    # bf16 GEMM mode (round 4): padded to a multiple of 64 instead, which makes the branches' first conv (3x3, 300 -> 128 at 15 x 20)
    # eligible for the LDS-DMA / fragment-streaming MFMA kernels: 41 instead of 75 us per branch at 32 pairs (150 instead of 300
    # workgroups); the 20 extra channels are zeros in both operands
    CORR_PAD_BF16 = 320

    def __init__(self, cfg, input_shape=None):
        C = cfg.MODEL.CAMERA_HEAD
        self.num_queries = cfg.MODEL.SEM_SEG_HEAD.NUM_OBJECT_QUERIES
        self.out_cam_type = C.INFERENCE_OUT_CAM_TYPE
        self.warp_plane_in_cam_ref_on = bool(C.WARP_PLANE_IN_CAM_REF_ON)
        self.matching_score_threshold = float(cfg.TEST.MATCHING_SCORE_THRESHOLD)
        # loss weights of the training-side forward (camera_head.py:46-48); forward_train only
        self.initial_cam_weight = float(getattr(C, "INITIAL_CAM_WEIGHT", 1.0))
        self.plane_cam_weight = float(getattr(C, "PLANE_CAM_WEIGHT", 1.0))
        self.plane_cam_weight_predplane = float(getattr(C, "PLANE_CAM_WEIGHT_PREDPLANE", 0.1))
        assert C.REFINE_ON and C.CAM_REC_ON and not C.INFERENCE_SP_TOPCAM_ON and cfg.MODEL.EMBEDDING_ON and cfg.MODEL.MASK_ON, \
            "implemented: the shipped inference configuration (REFINE_ON, CAM_REC_ON, plane matcher on; inference_mp3d.yaml:17-23)"
        assert not cfg.TEST.POSE_REFINEMENT_WITH_GT_MATCHERS, "GT matchers are an evaluation-only mode"
        assert self.out_cam_type in CAM_MODES
        self.fused_branch_tail = os.environ.get("NOPESAC_BRANCH_TAIL_UNFUSED", "0") != "1"     # bf16 mode: pose-net branch layers 1..5 in one launch
        assert cfg.MODEL.SEM_SEG_HEAD.NORM == "GN" and cfg.MODEL.SEM_SEG_HEAD.CONVS_DIM == 128
        spec = {k[len("camera_head_list.0."):]: v for k, v in state_dict_spec(self.num_queries).items()
                if k.startswith("camera_head_list.0.")}
        super().__init__(spec)
        # MODEL.AMD.POSE_FP32_PARTS: stages of this head that keep f32 operands in bf16 mode (scripts/bf16_attribution.py tables what
        # each one contributes to the bf16 pose error): "decoder" (pixel decoder + the six backbone-side convs), "branches" (the
        # affinity volume and the 2 x 6 strided convs), "fc" (FC + pose regressors), "aim" (re-embedding MLPs), "refine" (RANSAC MLPs)
        from ..config import amd_options
        self.fp32_parts = frozenset(str(amd_options(cfg).POSE_FP32_PARTS).replace(",", " ").split())
        assert self.fp32_parts <= {"decoder", "branches", "fc", "aim", "refine"}, self.fp32_parts

    def _gd(self, part: str):
        return torch.float32 if part in self.fp32_parts else self.gemm_dtype

    # ---------------------------------------------------------------- packing
    def pack(self) -> dict:
        P = {}
        for nm in ("adapter_1", "adapter_2", "layer_1", "layer_2", "layer_3"):
            P[nm] = ConvW(self.raw(f"pixel_decoder.{nm}.weight"))
        P["mask_features"] = conv_bias(self, "pixel_decoder.mask_features")
        for i in (0, 1, 3, 4, 6, 7):
            P[f"cb{i}"] = conv_bn(self, f"convs_backbone.{i}.0.weight", f"convs_backbone.{i}.1", 1e-3)
        for br in ("convs_trans", "convs_rots"):
            for i in range(6):
                # the first conv reads the 300-channel correlation volume: pad Cin to 304 (zeros) so its im2col rows are
                # 8-element aligned and the vector / MFMA-friendly staging path applies
                P[f"{br}.{i}"] = conv_bn(self, f"{br}.{i}.0.weight", f"{br}.{i}.1", 1e-3, cin_pad=self.CORR_PAD if i == 0 else 0)
            P[f"{br}.0.pad64"] = conv_bn(self, f"{br}.0.0.weight", f"{br}.0.1", 1e-3, cin_pad=self.CORR_PAD_BF16)
        for nm in ("fc_trans", "fc_rots"):
            # the reference flattens NCHW (128,2,3) -> index c*6+hw; our activations are NHWC -> hw*128+c
            w = self.raw(nm + ".weight").float().view(256, 128, 6).permute(0, 2, 1).reshape(256, 768)
            P[nm] = ConvW(w.contiguous(), None, self.raw(nm + ".bias"))
        for nm in ("trans", "rots", "rot_score_reg", "trans_score_reg"):
            P[nm] = conv_bias(self, nm)
        for nm in ("rot_emb_proj", "trans_emb_proj", "geo_encoder", "geo_proj_s1", "decoder_rot", "geo_proj_s2", "decoder_tran",
                   "decoder_rot2", "decoder_tran2", "normal_score_proj", "param_score_proj"):
            P[nm] = mlp_layers(self, nm)
        return P

    def _gn(self, x, nm, act):
        return ops.groupnorm(x, self.raw(f"pixel_decoder.{nm}.norm.weight"), self.raw(f"pixel_decoder.{nm}.norm.bias"), 32, 1e-5, act)

    # ---------------------------------------------------------------- (i) pixel pose net
    def pixel_pose_net(self, feats: dict, B: int, canonical_sign: bool = True, features_only: bool = False):
        """feats: NHWC res3..res5 for 2B images (view-1 images first) -> trans0 [B,3], rot0 [B,4] (unit, w>=0),
        trans_feat, rots_feat [B,256].  `canonical_sign=False`: the training forward keeps the regressed sign (:667)."""
        P, gd = self.packed, self._gd("decoder")
        r3, r4, r5 = feats["res3"], feats["res4"], feats["res5"]
        if "decoder" in self.fp32_parts and r5.dtype != torch.float32:
            r3, r4, r5 = r3.float(), r4.float(), r5.float()

        def cv(x, nm, pad=0, stride=1, act=ops.ACT_NONE, out_dtype=None):
            c = P[nm]
            return ops.conv2d(x, c.w(gd if x.dtype == torch.float32 else x.dtype), c.scale, c.bias, stride=stride, pad=pad, act=act,
                              out_dtype=out_dtype)

        y = self._gn(cv(r5, "layer_3", 1), "layer_3", ops.ACT_RELU)
        y = ops.upsample2x_nearest_add(y, self._gn(cv(r4, "adapter_2"), "adapter_2", ops.ACT_NONE))
        y = self._gn(cv(y, "layer_2", 1), "layer_2", ops.ACT_RELU)
        y = ops.upsample2x_nearest_add(y, self._gn(cv(r3, "adapter_1"), "adapter_1", ops.ACT_NONE))
        y = self._gn(cv(y, "layer_1", 1), "layer_1", ops.ACT_RELU)
        x = cv(y, "mask_features", 1)
        x = cv(cv(x, "cb0", 1, act=ops.ACT_LEAKY), "cb1", 1, act=ops.ACT_LEAKY)
        x = ops.maxpool(x, 2, 2, 0)
        x = cv(cv(x, "cb3", 1, act=ops.ACT_LEAKY), "cb4", 1, act=ops.ACT_LEAKY)
        x = ops.maxpool(x, 2, 2, 0)
        x = cv(cv(x, "cb6", 1, act=ops.ACT_LEAKY), "cb7", 1, act=ops.ACT_LEAKY, out_dtype=torch.float32)   # [2B,h,w,256] f32
        _, h, w, _ = x.shape
        assert h * w == 300, \
            f"the correlation stack of the pixel pose net has 300 = 15*20 input channels: inputs must be 480x640 (got a {h}x{w} 1/32 map)"
        x1, x2 = x[:B], x[B:]
        # correlation volume (:1117-1133): channel = view-2 position in (w,h) order, softmax over channels
        x2t = ops.transpose_hw_rows(x2.reshape(B, h * w, 256), h, w)
        corr = ops.conv2d(x1, x2t.view(B, h * w, 1, 1, 256), batched_weights=True)          # [B,h,w,h*w]
        # bf16 GEMM mode: the branch convs round their f32 activations to bf16 while staging them anyway, so the affinity volume and
        # the activations between the branch convs are STORED as bf16 (the same values) - the twelve convs then run on the bf16
        # conv kernels instead of the register-staged mixed-precision one (0.43 -> 0.2 ms per step) and move half the bytes
        gd = self._gd("branches")
        act_dt = torch.bfloat16 if gd == torch.bfloat16 else torch.float32
        # softmax over the channels, written in the conv operand type with zero-padded channels (see pack): 300 -> 304
        wide = act_dt == torch.bfloat16 and (h * w) % 64 != 0
        aff = ops.softmax_rows(corr, out_dtype=act_dt, pad_to=self.CORR_PAD_BF16 if wide else (self.CORR_PAD if (h * w) % 8 else h * w))

        def conv0(name):
            return cv(aff, f"{name}.0" + (".pad64" if wide else ""), 1, 1, ops.ACT_LEAKY)

        def head(t, fc, reg):
            # FC + ReLU, then the regressor on top of it (one launch in bf16 GEMM mode)
            return run_stacks(t.reshape(B, -1), [([P[fc]], ops.ACT_RELU, True), ([P[reg]], ops.ACT_NONE, True)], self._gd("fc"))

        def branch(name):
            t = conv0(name)
            for i in range(1, 6):
                t = cv(t, f"{name}.{i}", 1, 2 if i % 2 == 1 else 1, ops.ACT_LEAKY, out_dtype=torch.float32 if i == 5 else None)
            return t

        if self.fused_branch_tail and act_dt == torch.bfloat16 and (h, w) == (15, 20):
            # layers 1..5 of BOTH branches in one launch, activations resident in LDS (csrc/posenet_branch.hip): 3 launches instead of 12
            if "branch_tail" not in P:
                P["branch_tail"] = ops.PoseBranchTail([P[f"convs_trans.{i}"] for i in range(1, 6)], [P[f"convs_rots.{i}"] for i in range(1, 6)])
            yt, yr = ops.posenet_branch_tail(conv0("convs_trans"), conv0("convs_rots"), P["branch_tail"])
        else:
            yt, yr = branch("convs_trans"), branch("convs_rots")
        if features_only:                                    # training.CameraHeadTrainer: the conv stacks' outputs [B, 2*3*128] (NHWC order)
            return yt.reshape(B, -1), yr.reshape(B, -1)
        (trans_feat, trans0), (rots_feat, rot_raw) = head(yt, "fc_trans", "trans"), head(yr, "fc_rots", "rots")
        rot0 = ops.normalize_rows(rot_raw, canonical_sign=canonical_sign)                         # :667, :436-437
        return trans0, rot0, trans_feat, rots_feat

    # ---------------------------------------------------------------- (ii) AIM
    def aim(self, trans0, rot0):
        P, gd = self.packed, self._gd("aim")
        rot_feat, rot_raw = run_stacks(rot0, [(P["rot_emb_proj"], ops.ACT_RELU, True), ([P["rots"]], ops.ACT_NONE, True)], gd)  # rot0 has w >= 0 (:695-696)
        rec_rot = ops.normalize_rows(rot_raw)
        eps = ops.cached_constant(self.__dict__.setdefault("_eps_row", {}), trans0.device,
                                  lambda: torch.full((1, trans0.shape[-1]), 1e-10, device=trans0.device, dtype=torch.float32))
        trans_feat, rec_trans = run_stacks(ops.add_rows(trans0, eps), [(P["trans_emb_proj"], ops.ACT_RELU, True), ([P["trans"]], ops.ACT_NONE, True)], gd)  # :718
        return rec_trans, rec_rot, trans_feat, rot_feat

    # ---------------------------------------------------------------- (iv) neural one-plane RANSAC
    def refine(self, A0, planes1, planes2, n1, n2, rec_trans, rec_rot, trans_feat, rot_feat, diagnostics=False, train=False):
        """`train=True`: the scoring / aggregation of the training-side twin (mode + 16 of nopesac_ransac_soft_vote)."""
        P, nq, gd = self.packed, self.num_queries, self._gd("refine")
        B = A0.shape[0]
        dev = A0.device
        geo_local, geo_global, sig, geo_enc, m = ops.geo_sequence(A0, planes1, planes2, n1, n2, rec_trans, rec_rot,
                                                                  self.warp_plane_in_cam_ref_on)
        rows = B * nq
        # (:957-990) row-wise MLP stacks; in bf16 GEMM mode each run_stacks call is ONE launch (csrc/mlp_chain.hip): 40 GEMMs -> 5
        cat1280 = torch.empty(rows, 1280, device=dev, dtype=torch.float32)                   # [ geo_proj_s1 out | decoder_rot out ]
        run_stacks(geo_enc.view(rows, 8), [(P["geo_encoder"], ops.ACT_NONE, None), (P["geo_proj_s1"], ops.ACT_NONE, cat1280[:, :1024])], gd)
        f_rot = run_stacks(cat1280[:, :1024], [(P["decoder_rot"], ops.ACT_NONE, cat1280[:, 1024:])], gd)[0]   # geo_fea_rot_all: two consumers
        f_tran = run_stacks(cat1280, [(P["geo_proj_s2"], ops.ACT_NONE, None), (P["decoder_tran"], ops.ACT_NONE, True)], gd)[1]
        # (:980-983) the initial pose features are broadcast over the pair's planes and concatenated in front of the per-plane ones
        fused_rot, rot_raw = run_stacks(f_rot, [(P["decoder_rot2"], ops.ACT_RELU, True), ([P["rots"]], ops.ACT_NONE, True)], gd,
                                        x_bcast=rot_feat, rows_per=nq)
        fused_tran, trans_raw = run_stacks(f_tran, [(P["decoder_tran2"], ops.ACT_RELU, True), ([P["trans"]], ops.ACT_NONE, True)], gd,
                                           x_bcast=trans_feat, rows_per=nq)
        rot_raw, trans_raw = rot_raw.view(B, nq, 4), trans_raw.view(B, nq, 3)
        maps = ops.ransac_score_maps(geo_local, rot_raw, trans_raw, rec_rot, rec_trans, m, diagnostics=diagnostics)
        sf_rot = run_mlp(maps["normal_score"].view(B * (nq + 1), nq), P["normal_score_proj"], gd=gd).view(B, nq + 1, 64)
        sf_tran = run_mlp(maps["param_score"].view(B * (nq + 1), nq), P["param_score_proj"], gd=gd).view(B, nq + 1, 64)
        vote = ops.ransac_soft_vote(sf_rot, sf_tran, P["rot_score_reg"].w2d().view(-1), P["rot_score_reg"].bias,
                                    P["trans_score_reg"].w2d().view(-1), P["trans_score_reg"].bias, rot_feat, trans_feat,
                                    fused_rot.view(B, nq, 256), fused_tran.view(B, nq, 256), P["rots"].w2d(), P["rots"].bias,
                                    P["trans"].w2d(), P["trans"].bias, maps, rec_rot, rec_trans, m,
                                    16 if train else CAM_MODES[self.out_cam_type])
        vote.update(m=m, geo_local=geo_local, geo_global=geo_global, sig=sig, maps=maps)
        return vote

    def forward_plane_cam_ref_head(self, A0, planes1, planes2, n1, n2, initial_trans, initial_rot, initial_trans_feat,
                                   initial_rot_feat, gt_pose, suffix: str = "", weight: float = 1.0):
        """Training-side twin of `refine` = the reference's __forward_PlaneCamRefHead (camera_head.py:737-923), FORWARD ONLY:
        clamp-renormalised hypothesis scores (:814-818, :852-854), average pose from the per-plane features (:858-867), soft pose as the
        prediction, and the seven refinement losses (:883-921) in one launch (nopesac_plane_cam_ref_losses).  The matched-plane
        sequence is given as in `refine` - an assignment matrix over the two plane sets (GT correspondences over GT planes for the
        'initCamRef' / 'initRecCamRef' calls :452-474, the predicted assignment for 'initCamRef_Aux' :480-500); every pair needs
        m >= 1 matches, as in the reference.  This method returns loss VALUES; the differentiable form of the same function - gradients of
        all its parameters from hand-written backward kernels, and an optimiser step - is nopesac_amd.training.RefineTrainer (round 5).
        Returns (losses, pred_cam) with the reference's keys; pred_cam's per-hypothesis entries describe pair 0."""
        out = self.refine(A0, planes1, planes2, n1, n2, initial_trans, initial_rot, initial_trans_feat, initial_rot_feat,
                          diagnostics=True, train=True)
        maps, m = out["maps"], out["m"]
        lv = ops.plane_cam_ref_losses(out, maps, m, gt_pose, weight)
        losses = {"%s_%s" % (nm, suffix): lv[i] for i, nm in enumerate(ops.PLANE_CAM_REF_LOSS_NAMES)}
        m0 = int(m[0])
        pred_cam = {"pred_trans": out["pred_trans"], "pred_rot": out["pred_rot"], "pred_trans_avg": out["avg_trans"],
                    "pred_rot_avg": out["avg_rot"], "all_pred_trans": maps["trans_all"][0:1, :m0 + 1],
                    "all_pred_rots": maps["rots_all"][0:1, :m0 + 1], "score_soft_rot": out["score_rot"][0:1, :m0 + 1, None],
                    "score_soft_offset": out["score_trans"][0:1, :m0 + 1, None], "l2_dist": maps["l2_dist"][0:1, :m0 + 1, :m0],
                    "normal_dist": maps["normal_angle"][0:1, :m0 + 1, :m0], "offset_dist": maps["offset_dist"][0:1, :m0 + 1, :m0]}
        return losses, pred_cam

    def forward_train(self, feats: dict, B: int, gt_planes1, gt_planes2, gt_n1, gt_n2, gt_assignment, gt_pose, planes1=None,
                      planes2=None, n1=None, n2=None, assignment=None, rand_rot=None, rand_trans=None):
        """The reference's PlaneCameraHead.forward in TRAINING mode, FORWARD + LOSSES ONLY (camera_head.py:140-189 ->
        forward_withInitialCam_Joint :191-323, forward_withRandCam_Joint :325-344): pixel pose loss (:672-680), the AIM's
        reconstruction losses on the pixel pose (:700-705, :725-731), the refinement twin from the pixel pose and from its re-embedding over
        the GT planes + GT correspondences ('initCamRef', 'initRecCamRef', weight PLANE_CAM_WEIGHT) and, if predicted planes + an
        assignment are given, over those ('..._Aux', weight PLANE_CAM_WEIGHT_PREDPLANE), and the AIM's losses on caller-provided
        random poses (the reference draws them, :687-690, :718).  Planes are [B,nq,3] zero-padded with counts int32[B]; assignments
        [B,nq,nq]; gt_pose [B,7].  BatchNorm / GroupNorm layers run with their stored statistics: this evaluates the losses of a
        checkpoint (validation curves, loss parity).  The training step over the same losses - backward kernels for every Linear layer
        of the head (pixel-pose FC + regressors, AIM, refinement head; the conv stacks stay frozen) + AdamW / SGD - is
        nopesac_amd.training.CameraHeadTrainer (round 5).  Returns (losses, trans_list, rot_list) as the reference does."""
        losses = {}
        trans0, rot0, tf0, rf0 = self.pixel_pose_net(feats, B, canonical_sign=False)
        lp = ops.camera_pose_loss(trans0, rot0, gt_pose[:, 0:3], gt_pose[:, 3:7], self.initial_cam_weight)
        losses["loss_tran_pixelReg"], losses["loss_rot_pixelReg"] = lp[0], lp[1]

        def rec(trans_in, rot_in, suffix):
            rot_c = ops.normalize_rows(rot_in, canonical_sign=True)          # input_rot * sig (:695-696)
            rec_t, rec_r, rec_tf, rec_rf = self.aim(trans_in, rot_c)
            lr = ops.camera_pose_loss(rec_t, rec_r, trans_in, rot_c, 1.0, trans_eps=1e-10)
            losses["loss_rot" + suffix], losses["loss_trans" + suffix] = lr[1], lr[0]
            return rec_t, rec_r, rec_tf, rec_rf

        rec_t, rec_r, rec_tf, rec_rf = rec(trans0, rot0, "_initCamRec")
        trans_list, rot_list = [trans0, rec_t], [rot0, rec_r]
        passes = [("", gt_planes1, gt_planes2, gt_n1, gt_n2, gt_assignment, self.plane_cam_weight)]
        if assignment is not None:
            passes.append(("_Aux", planes1, planes2, n1, n2, assignment, self.plane_cam_weight_predplane))
        for sfx, pl1, pl2, c1, c2, A, w in passes:
            for name, it, ir, itf, irf in (("initCamRef", trans0, rot0, tf0, rf0), ("initRecCamRef", rec_t, rec_r, rec_tf, rec_rf)):
                ls, pr = self.forward_plane_cam_ref_head(A, pl1, pl2, c1, c2, it, ir, itf, irf, gt_pose, suffix=name + sfx, weight=w)
                losses.update(ls)
                trans_list += [pr["pred_trans_avg"], pr["pred_trans"]]
                rot_list += [pr["pred_rot_avg"], pr["pred_rot"]]
        if rand_rot is not None:
            rec(rand_trans, rand_rot, "_randCamRecLBS_N1")
        return losses, trans_list, rot_list

    # ---------------------------------------------------------------- whole head
    def initial_pose(self, feats: dict, B: int):
        """(i) + (ii): everything that depends only on the backbone maps (can run on a side stream while the
        plane head works on the main stream)."""
        trans0, rot0, tf0, rf0 = self.pixel_pose_net(feats, B)
        return (trans0, rot0) + self.aim(trans0, rot0)

    def forward(self, feats: dict, sel: dict, matching_net, B: int, diagnostics: bool = False,
                forced_assignment: torch.Tensor = None, pose=None, mark=None) -> dict:
        """feats: NHWC backbone maps of the 2B images (view-1 first); sel: output of plane post-selection for
        the 2B images (planes [2B,nq,3], feats [2B,nq,256], n_kept int32[2B]).  Returns device tensors:
        cameras {name: (tran [B,3], rot [B,4])}, assignments [B,nq,nq], log_scores [B,nq+1,nq+1], m [B] ..."""
        mark = mark or (lambda name: None)
        trans0, rot0, rec_t, rec_r, rec_tf, rec_rf = pose if pose is not None else self.initial_pose(feats, B)
        mark("pose_net(main-stream part)")
        n_all = sel["n_kept"]
        n1, n2 = n_all[:B].contiguous(), n_all[B:].contiguous()
        planes1, planes2 = sel["planes"][:B], sel["planes"][B:]
        cam7 = ops.concat_cols(rec_t, rec_r) if rec_t.is_cuda else torch.cat([rec_t, rec_r], dim=-1)   # :455
        log_scores, A0 = matching_net(sel["feats"], n_all, cam7, planes1, planes2, self.matching_score_threshold)
        if forced_assignment is not None:       # benchmark-only K control, see PlaneTR_NopeSAC._force_k
            A0 = forced_assignment
        mark("matcher")
        ref = self.refine(A0, planes1, planes2, n1, n2, rec_t, rec_r, rec_tf, rec_rf, diagnostics)
        A1 = ops.refilter_assignment(A0, planes1, planes2, n1, n2, ref["pred_rot"], ref["pred_trans"])
        def make_zero_pose():
            zr = torch.zeros_like(rot0)
            zr[:, 0] = 1.0
            return torch.zeros_like(trans0), zr

        # the constant "camera_zero" pose (never written to)
        zero_t, zero_r = ops.cached_constant(self.__dict__.setdefault("_zero_cam", {}), (B, trans0.device), make_zero_pose)
        cams = {"camera_zero": (zero_t, zero_r), "camera_init": (trans0, rot0), "camera_initRec": (rec_t, rec_r),
                "camera_avgRef0": (ref["avg_trans"], ref["avg_rot"]), "camera_softRef0": (ref["pred_trans"], ref["pred_rot"]),
                "camera": (ref["pred_trans"], ref["pred_rot"])}                             # sign NOT canonicalised (:596-601)
        return {"cameras": cams, "pred_assignment_beforeRef0": A0, "pred_assignment_afterRef0": A1, "pred_assignment": A1,
                "log_scores_padded": log_scores, "m": ref["m"], "n1": n1, "n2": n2, "refine": ref,
                "init_feats": (rec_tf, rec_rf)}


def build_camera_head(cfg, input_shape=None):
    return CAMERA_HEAD_REGISTRY.get(cfg.MODEL.CAMERA_HEAD.NAME)(cfg, input_shape)
