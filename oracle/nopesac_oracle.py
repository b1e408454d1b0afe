"""CPU oracle: a functional, pure-PyTorch fp32 restatement of NopeSAC's inference hot path.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline`
leg may import this module, and only as the checker / the reported CPU baseline.  The product
(nopesac_amd/) never imports it and has no CPU fallback.

Pinning: the reference ships NO tests or golden vectors -> "parity unpinned
by the reference's own tests".  Instead this restatement is checked against the *reference
itself*, imported and run on CPU in the build container through oracle/ref_shim.py, by
oracle/gen_golden.py (which also writes the committed fixtures under tests/golden/), and
against those fixtures by tests/test_oracle_golden.py everywhere else.  The ResNet-50 lives in
detectron2==0.4, which is absent from /root/reference: the backbone is pinned only up to our
restatement of d2 semantics (oracle/d2_resnet.py).

Everything is a function of a reference-named state dict `sd`.
Citations are path:line under /root/reference/NopeSAC_Net/modeling/.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch
from torch.nn import functional as F

from . import rle_oracle

Tensor = torch.Tensor


@dataclass
class OracleConfig:
    """The config values the hot path reads (config/config.py:5-114 defaults +
    configs/inference_mp3d.yaml)."""
    num_queries: int = 50                 # MODEL.SEM_SEG_HEAD.NUM_OBJECT_QUERIES
    nheads: int = 8
    pixel_mean: tuple = (123.675, 116.280, 103.530)
    pixel_std: tuple = (58.395, 57.120, 57.375)
    overlap_threshold: float = 0.6        # TEST.OVERLAP_THRESHOLD
    plane_score_threshold: float = 0.6    # TEST.PLANE_SCORE_THRESHOLD
    mask_prob_threshold: float = 0.5      # TEST.MASK_PROB_THRESHOLD
    matching_score_threshold: float = 0.2  # TEST.MATCHING_SCORE_THRESHOLD
    offset_multiplier: float = 4.0        # MODEL.MATCHING_HEAD.OFFSET_MULTIPLIER
    normal_multiplier: float = 8.0        # MODEL.MATCHING_HEAD.NORMAL_MULTIPLIER
    sinkhorn_iterations: int = 200        # matching_head.py:38
    out_cam_type: str = "soft"            # MODEL.CAMERA_HEAD.INFERENCE_OUT_CAM_TYPE
    warp_plane_in_cam_ref: bool = True    # MODEL.CAMERA_HEAD.WARP_PLANE_IN_CAM_REF_ON


# ======================================================================================
# small building blocks
# ======================================================================================
def _sub(sd: Dict[str, Tensor], prefix: str) -> Dict[str, Tensor]:
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


def frozen_bn(x: Tensor, sd, p: str, eps: float = 1e-5) -> Tensor:
    """d2 FrozenBatchNorm2d: y = x*w*rsqrt(var+eps) + (b - mean*w*rsqrt(var+eps))."""
    scale = sd[p + ".weight"] * (sd[p + ".running_var"] + eps).rsqrt()
    shift = sd[p + ".bias"] - sd[p + ".running_mean"] * scale
    return x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)


def eval_bn(x: Tensor, sd, p: str, eps: float) -> Tensor:
    """nn.BatchNorm2d in eval mode."""
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"],
                        sd[p + ".bias"], False, 0.0, eps)


def mlp(x: Tensor, sd, p: str) -> Tensor:
    """MLP with ReLU between layers (planeTR_head.py:194-206, camera_modules.py:226-244)."""
    n = 0
    while f"{p}.layers.{n}.weight" in sd:
        n += 1
    for i in range(n):
        x = F.linear(x, sd[f"{p}.layers.{i}.weight"], sd[f"{p}.layers.{i}.bias"])
        if i < n - 1:
            x = F.relu(x)
    return x


def layer_norm(x: Tensor, sd, p: str) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def mha(q_in: Tensor, k_in: Tensor, v_in: Tensor, sd, p: str, nheads: int) -> Tensor:
    """nn.MultiheadAttention forward (eval, no masks) on [L,B,E] tensors."""
    L, B, E = q_in.shape
    S = k_in.shape[0]
    w, b = sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"]
    q = F.linear(q_in, w[:E], b[:E])
    k = F.linear(k_in, w[E:2 * E], b[E:2 * E])
    v = F.linear(v_in, w[2 * E:], b[2 * E:])
    d = E // nheads
    q = q.reshape(L, B * nheads, d).transpose(0, 1) * (d ** -0.5)
    k = k.reshape(S, B * nheads, d).transpose(0, 1)
    v = v.reshape(S, B * nheads, d).transpose(0, 1)
    a = torch.softmax(torch.bmm(q, k.transpose(1, 2)), dim=-1)
    o = torch.bmm(a, v).transpose(0, 1).reshape(L, B, E)
    return F.linear(o, sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])


# ======================================================================================
# a2 preprocess + a3 backbone  (siamese_planeTR.py:85-89,534-542; d2 ResNet, Appendix A)
# ======================================================================================
def preprocess(images: List[Tensor], cfg: OracleConfig) -> Tensor:
    mean = torch.tensor(cfg.pixel_mean).view(-1, 1, 1)
    std = torch.tensor(cfg.pixel_std).view(-1, 1, 1)
    return torch.stack([(x - mean) / std for x in images], 0)


def _conv_bn(x, sd, p, stride=1, pad=0):
    return frozen_bn(F.conv2d(x, sd[p + ".weight"], None, stride, pad), sd, p + ".norm")


def backbone(sd, x: Tensor) -> Dict[str, Tensor]:
    """ResNet-50, stride on the 3x3 (configs/Base.yaml:11), outputs res2..res5."""
    x = F.relu(_conv_bn(x, sd, "backbone.stem.conv1", 2, 3))
    x = F.max_pool2d(x, 3, 2, 1)
    out = {}
    for name, nblk in (("res2", 3), ("res3", 4), ("res4", 6), ("res5", 3)):
        for i in range(nblk):
            p = f"backbone.{name}.{i}"
            stride = 2 if (i == 0 and name != "res2") else 1
            y = F.relu(_conv_bn(x, sd, p + ".conv1"))
            y = F.relu(_conv_bn(y, sd, p + ".conv2", stride, 1))
            y = _conv_bn(y, sd, p + ".conv3")
            s = _conv_bn(x, sd, p + ".shortcut", stride) if (p + ".shortcut.weight") in sd else x
            x = F.relu(y + s)
        out[name] = x
    return out


# ======================================================================================
# a4 PlaneTR head  (planeTR_net/planeTR_head.py:116-192)
# ======================================================================================
def sine_position_embedding(b: int, h: int, w: int, num_pos_feats: int = 128) -> Tensor:
    """transformer/position_encoding.py:29-52 with normalize=True, mask=None."""
    eps, scale, temperature = 1e-6, 2 * math.pi, 10000.0
    y_embed = torch.arange(1, h + 1, dtype=torch.float32).view(1, h, 1).expand(b, h, w)
    x_embed = torch.arange(1, w + 1, dtype=torch.float32).view(1, 1, w).expand(b, h, w)
    y_embed = y_embed / (y_embed[:, -1:, :] + eps) * scale
    x_embed = x_embed / (x_embed[:, :, -1:] + eps) * scale
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / num_pos_feats)
    pos_x = x_embed[:, :, :, None] / dim_t
    pos_y = y_embed[:, :, :, None] / dim_t
    pos_x = torch.stack((pos_x[..., 0::2].sin(), pos_x[..., 1::2].cos()), dim=4).flatten(3)
    pos_y = torch.stack((pos_y[..., 0::2].sin(), pos_y[..., 1::2].cos()), dim=4).flatten(3)
    return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)


def detr_encoder(sd, p: str, src: Tensor, pos: Tensor, nheads: int) -> Tensor:
    """6 post-norm layers + final LN (transformer/transformer.py:86-103,183-199)."""
    i = 0
    while f"{p}.layers.{i}.linear1.weight" in sd:
        lp = f"{p}.layers.{i}"
        q = src + pos
        src = layer_norm(src + mha(q, q, src, sd, lp + ".self_attn", nheads), sd, lp + ".norm1")
        ff = F.linear(F.relu(F.linear(src, sd[lp + ".linear1.weight"], sd[lp + ".linear1.bias"])),
                      sd[lp + ".linear2.weight"], sd[lp + ".linear2.bias"])
        src = layer_norm(src + ff, sd, lp + ".norm2")
        i += 1
    return layer_norm(src, sd, p + ".norm")


def detr_decoder(sd, p: str, tgt: Tensor, memory: Tensor, pos: Tensor, query_pos: Tensor,
                 nheads: int) -> Tensor:
    """6 pre-norm layers; returns norm(output of last layer) = hs[-1]
    (transformer/transformer.py:114-152,293-322)."""
    i = 0
    mem_k = memory + pos
    while f"{p}.layers.{i}.linear1.weight" in sd:
        lp = f"{p}.layers.{i}"
        t2 = layer_norm(tgt, sd, lp + ".norm1")
        q = t2 + query_pos
        tgt = tgt + mha(q, q, t2, sd, lp + ".self_attn", nheads)
        t2 = layer_norm(tgt, sd, lp + ".norm2")
        tgt = tgt + mha(t2 + query_pos, mem_k, memory, sd, lp + ".multihead_attn", nheads)
        t2 = layer_norm(tgt, sd, lp + ".norm3")
        tgt = tgt + F.linear(F.relu(F.linear(t2, sd[lp + ".linear1.weight"], sd[lp + ".linear1.bias"])),
                             sd[lp + ".linear2.weight"], sd[lp + ".linear2.bias"])
        i += 1
    return layer_norm(tgt, sd, p + ".norm")


def _conv_bn_relu_1x1(x, sd, p):
    """planeTR_head.py:209-215: Conv2d(1x1, no bias) + BatchNorm2d(eps 1e-5) + ReLU."""
    return F.relu(eval_bn(F.conv2d(x, sd[p + ".0.weight"]), sd, p + ".1", 1e-5))


def top_down(sd, p: str, feats, memory: Tensor) -> Tensor:
    """planeTR_head.py:241-252."""
    c1, c2, c3, c4 = feats
    up = lambda t: F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=False)
    p4 = _conv_bn_relu_1x1(c4, sd, p + ".c4_conv") + _conv_bn_relu_1x1(memory, sd, p + ".m_conv_dict.m4")
    p3 = _conv_bn_relu_1x1(up(p4), sd, p + ".up_conv3") + _conv_bn_relu_1x1(c3, sd, p + ".c3_conv")
    p2 = _conv_bn_relu_1x1(up(p3), sd, p + ".up_conv2") + _conv_bn_relu_1x1(c2, sd, p + ".c2_conv")
    p1 = _conv_bn_relu_1x1(up(p2), sd, p + ".up_conv1") + _conv_bn_relu_1x1(c1, sd, p + ".c1_conv")
    return p1


def plane_head(sd, feats: Dict[str, Tensor], cfg: OracleConfig):
    """-> (outputs dict, query_feat [B,nq,256]); only the last decoder layer is produced
    (planeTR_head.py:168-192 uses [-1] of everything at inference)."""
    h = "sem_seg_head"
    c1, c2, c3, c4 = feats["res2"], feats["res3"], feats["res4"], feats["res5"]
    b, _, hc, wc = c4.shape
    pos = sine_position_embedding(b, hc, wc).flatten(2).permute(2, 0, 1)
    src = F.conv2d(c4, sd[h + ".input_proj.weight"], sd[h + ".input_proj.bias"]).flatten(2).permute(2, 0, 1)
    memory = detr_encoder(sd, h + ".context_SA", src, pos, cfg.nheads)
    query_pos = sd[h + ".query_embed.weight"].unsqueeze(1).repeat(1, b, 1)
    hs = detr_decoder(sd, h + ".context2plane_decoder", torch.zeros_like(query_pos), memory, pos,
                      query_pos, cfg.nheads).transpose(0, 1)  # b, nq, c
    mem_map = memory.permute(1, 2, 0).reshape(b, -1, hc, wc)
    p_context = top_down(sd, h + ".top_down", (c1, c2, c3, c4), mem_map)
    plane_emb = mlp(hs, sd, h + ".plane_embedding")
    pix_emb = F.conv2d(p_context, sd[h + ".pixel_embedding.weight"], sd[h + ".pixel_embedding.bias"])
    out = {
        "pred_logits": F.linear(hs, sd[h + ".plane_prob.weight"], sd[h + ".plane_prob.bias"]),
        "pred_mask_logits": torch.einsum("bqc,bchw->bqhw", plane_emb, pix_emb),
        "pred_params": mlp(hs, sd, h + ".plane_param"),
        "pred_centers": torch.sigmoid(mlp(hs, sd, h + ".plane_center")),
        "pixel_centers": torch.sigmoid(F.conv2d(p_context, sd[h + ".pixel_plane_center.weight"],
                                                sd[h + ".pixel_plane_center.bias"])),
    }
    return out, hs


# ======================================================================================
# a5 plane post-selection  (meta_arch/siamese_planeTR.py:625-803)
# ======================================================================================
def post_select(logits: Tensor, params: Tensor, mask_logits: Tensor, query_feat: Tensor,
                cfg: OracleConfig, height: int = 480, width: int = 640) -> Dict[str, Tensor]:
    """One image.  Returns kept planes ordered by query index:
    pred_plane [n,3], pred_plane_feats [1,n,256], pred_plane_masks [n,H,W] bool,
    pred_plane_oriIdxs [n], pred_plane_ins_center [n,2], scores [n], areas [n]."""
    nq = logits.shape[0]
    prob_full = F.interpolate(torch.sigmoid(mask_logits)[:, None], size=(height, width),
                              mode="bilinear", align_corners=False)[:, 0]           # :647-648
    cls_prob = F.softmax(logits, dim=-1)
    score, labels = cls_prob.max(dim=-1)
    label_mask = (labels == 0) & (score > cfg.plane_score_threshold)                 # :652-654
    zero_flag = False
    if int(label_mask.sum()) == 0:                                                   # :657-661
        idx = int(cls_prob[:, 0].argmax())
        label_mask[idx] = True
        score = score.clone()
        score[idx] = cls_prob[idx, 0]
        zero_flag = True
    ori_idx = torch.arange(nq)[label_mask]
    v_param, v_score, v_prob = params[label_mask], score[label_mask], prob_full[label_mask]
    v_feat = query_feat[label_mask]
    weighted = v_score.view(-1, 1, 1) * v_prob
    ids = weighted.argmax(0)                                                         # :674
    xs = (torch.arange(width, dtype=torch.float32) / width).view(1, width)
    ys = (torch.arange(height, dtype=torch.float32) / height).view(height, 1)
    keep, masks, centers, areas = [], [], [], []
    max_overlap, max_overlap_id = 0.0, 0
    for pi in range(v_param.shape[0]):                                               # :684-739
        m = (ids == pi) & (weighted[pi] > cfg.mask_prob_threshold)
        area = int(m.sum())
        ori_area = int((v_prob[pi] >= cfg.mask_prob_threshold).sum())
        if not zero_flag:
            if area < 1 or ori_area < 1:
                continue
            overlap = area / ori_area
            if overlap > max_overlap:
                max_overlap, max_overlap_id = overlap, pi
            if overlap < cfg.overlap_threshold:
                continue
        elif area == 0:
            m = m.clone()
            m[0, 0] = True
        mf = m.double()
        n_pix = mf.sum()
        cx = (xs.double() * mf).sum() / (n_pix + 1e-10)
        cy = (ys.double() * mf).sum() / (n_pix + 1e-10)
        keep.append(pi); masks.append(m); areas.append(int(m.sum()))
        centers.append(torch.stack([cx, cy]).float())
    if not keep:                                                                     # :741-788
        pi = max_overlap_id
        m = ids == pi
        mf = m.double()
        n_pix = mf.sum()
        keep.append(pi); masks.append(m); areas.append(int(n_pix))
        centers.append(torch.stack([(xs.double() * mf).sum() / n_pix, (ys.double() * mf).sum() / n_pix]).float())
    k = torch.tensor(keep)
    instances = []                                                                   # :703-720 / :747-766
    for j, m in enumerate(masks):
        rle = rle_oracle.encode(m.numpy())
        instances.append({"category_id": 0, "score": float(v_score[k[j]]),
                          "segmentation": {"size": [height, width], "counts": rle["counts"]},
                          "bbox": rle_oracle.to_bbox(rle).tolist(), "bbox_mode": 1})
    return {
        "instances": instances,
        "pred_plane": v_param[k], "pred_plane_feats": v_feat[k].unsqueeze(0).contiguous(),
        "pred_plane_masks": torch.stack(masks, 0), "pred_plane_oriIdxs": ori_idx[k],
        "pred_plane_ins_center": torch.stack(centers, 0), "scores": v_score[k],
        "areas": torch.tensor(areas),
    }


# ======================================================================================
# geometry helpers (camera_head.py:1135-1177, 1427-1466; matching_head.py:141-224)
# ======================================================================================
_FLIP = torch.tensor([1.0, -1.0, -1.0])


def quat_to_rotmat(q: Tensor) -> Tensor:
    w, x, y, z = q.unbind(-1)
    return torch.stack([
        1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * w * z, 2 * x * z + 2 * w * y,
        2 * x * y + 2 * w * z, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * w * x,
        2 * x * z - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x * x - 2 * y * y], dim=-1).reshape(*q.shape[:-1], 3, 3)


def warp_planes(plane: Tensor, rot_quat: Tensor, tran: Tensor) -> Tensor:
    """plane [...,n,3] (n*d, camera frame), rot_quat [...,4], tran [...,3] -> warped [...,n,3].
    end = R*flip(p) + t; b = end - t; out = ((end.b)/(|b|+1e-5)^2) * b."""
    R = quat_to_rotmat(rot_quat)
    t = tran.unsqueeze(-2)
    end = torch.einsum("...ij,...nj->...ni", R, plane * _FLIP) + t
    b = end - t            # the reference forms b = (R p + t) - t, keep its rounding
    coef = (end * b).sum(-1) / (b.norm(dim=-1) + 1e-5) ** 2
    return coef.unsqueeze(-1) * b


def flip_planes(plane: Tensor) -> Tensor:
    return plane * _FLIP


def _geometric_dists(planes1, planes2, rot, tran, off_min, off_max):
    """normal angle [deg] and offset distance between warped view-1 and flipped view-2 planes
    (matching_head.py:75-96, camera_head.py:605-621).  planes [n,3]; rot [4]; tran [3]."""
    p2 = flip_planes(planes2)
    off2 = p2.norm(dim=-1, keepdim=True)
    n2 = F.normalize(p2, dim=-1)
    p1_r = warp_planes(planes1, rot, torch.zeros_like(tran))
    n1_r = F.normalize(p1_r, dim=-1)
    ang = torch.acos(torch.clamp(n1_r @ n2.T, -1, 1)) / np.pi * 180.0
    p1_rt = warp_planes(planes1, rot, tran)
    off1 = p1_rt.norm(dim=-1, keepdim=True)
    n1_rt = F.normalize(p1_rt, dim=-1)
    ntn = n1_rt @ n2.T
    off = torch.where(ntn < 0, (off1 + off2.T).abs(), (off1 - off2.T).abs())
    return ang, torch.clamp(off, min=off_min, max=off_max)


# ======================================================================================
# a7/a8 pixel pose-regression net + AIM  (camera_head.py:642-735, camera_modules.py:246-348)
# ======================================================================================
def _gn_conv(x, sd, p, pad, relu):
    """d2 Conv2d(bias=False) + GroupNorm(32) [+ ReLU] (camera_modules.py:271-303)."""
    y = F.group_norm(F.conv2d(x, sd[p + ".weight"], None, 1, pad), 32, sd[p + ".norm.weight"],
                     sd[p + ".norm.bias"], 1e-5)
    return F.relu(y) if relu else y


def pixel_decoder(sd, p: str, feats: Dict[str, Tensor]) -> Tensor:
    """BasePixelDecoder.forward_features on res5->res4->res3 (camera_modules.py:335-348)."""
    y = _gn_conv(feats["res5"], sd, p + ".layer_3", 1, True)
    for name, idx in (("res4", 2), ("res3", 1)):
        lat = _gn_conv(feats[name], sd, f"{p}.adapter_{idx}", 0, False)
        y = lat + F.interpolate(y, size=lat.shape[-2:], mode="nearest")
        y = _gn_conv(y, sd, f"{p}.layer_{idx}", 1, True)
    return F.conv2d(y, sd[p + ".mask_features.weight"], sd[p + ".mask_features.bias"], 1, 1)


def _conv_bn_lrelu(x, sd, p, stride=1):
    """camera_modules.py:36-48: conv3x3(no bias) + BatchNorm2d(eps 1e-3) + LeakyReLU(0.01)."""
    return F.leaky_relu(eval_bn(F.conv2d(x, sd[p + ".0.weight"], None, stride, 1), sd, p + ".1", 1e-3), 0.01)


def corr_softmax(f1: Tensor, f2: Tensor) -> Tensor:
    """camera_head.py:1117-1133: view-2 positions enumerated in (w,h) order, softmax over them."""
    b, c, h1, w1 = f1.shape
    f2v = f2.transpose(2, 3).reshape(b, c, -1).transpose(1, 2)   # b, w2*h2, c
    corr = torch.matmul(f2v, f1.reshape(b, c, -1))               # b, w2h2, h1w1
    return F.softmax(corr.view(b, -1, h1, w1), dim=1)


def pixel_pose_net(sd, feats1, feats2, p: str = "camera_head_list.0"):
    def tower(x):
        x = _conv_bn_lrelu(_conv_bn_lrelu(x, sd, p + ".convs_backbone.0"), sd, p + ".convs_backbone.1")
        x = F.max_pool2d(x, 2, 2)
        x = _conv_bn_lrelu(_conv_bn_lrelu(x, sd, p + ".convs_backbone.3"), sd, p + ".convs_backbone.4")
        x = F.max_pool2d(x, 2, 2)
        return _conv_bn_lrelu(_conv_bn_lrelu(x, sd, p + ".convs_backbone.6"), sd, p + ".convs_backbone.7")

    x1 = tower(pixel_decoder(sd, p + ".pixel_decoder", feats1))
    x2 = tower(pixel_decoder(sd, p + ".pixel_decoder", feats2))
    aff = corr_softmax(x1, x2)

    def branch(name, fc):
        y = aff
        for i in range(6):
            y = _conv_bn_lrelu(y, sd, f"{p}.{name}.{i}", 2 if i % 2 == 1 else 1)
        return F.relu(F.linear(y.flatten(1), sd[f"{p}.{fc}.weight"], sd[f"{p}.{fc}.bias"]))

    trans_feat, rots_feat = branch("convs_trans", "fc_trans"), branch("convs_rots", "fc_rots")
    trans = F.linear(trans_feat, sd[p + ".trans.weight"], sd[p + ".trans.bias"])
    rots = F.normalize(F.linear(rots_feat, sd[p + ".rots.weight"], sd[p + ".rots.bias"]), p=2, dim=1)
    return trans, rots, trans_feat, rots_feat, aff


def aim_reembed(sd, trans: Tensor, rot: Tensor, p: str = "camera_head_list.0"):
    """Arbitrary Initialisation Module (camera_head.py:685-735)."""
    sig = ((rot[:, 0:1] >= 0.0).float() - 0.5) * 2.0
    rot_feat = F.relu(mlp(rot * sig, sd, p + ".rot_emb_proj"))
    rec_rot = F.normalize(F.linear(rot_feat, sd[p + ".rots.weight"], sd[p + ".rots.bias"]), p=2, dim=1)
    trans_feat = F.relu(mlp(trans + 1e-10, sd, p + ".trans_emb_proj"))
    rec_trans = F.linear(trans_feat, sd[p + ".trans.weight"], sd[p + ".trans.bias"])
    return rec_trans, rec_rot, trans_feat, rot_feat


# ======================================================================================
# a9/a10/a11 matching head  (matching_net/matching_head.py:43-133,228-306; gnn.py)
# ======================================================================================
def gnn_layer(sd, p: str, x: Tensor, source: Tensor, nheads: int = 8) -> Tensor:
    """LoFTR-style layer (transformer/gnn.py:73-96); x [L,C], source [S,C]."""
    L, C = x.shape
    d = C // nheads
    q = F.linear(x, sd[p + ".q_proj.weight"]).view(L, nheads, d)
    k = F.linear(source, sd[p + ".k_proj.weight"]).view(-1, nheads, d)
    v = F.linear(source, sd[p + ".v_proj.weight"]).view(-1, nheads, d)
    a = torch.softmax(torch.einsum("lhd,shd->lsh", q, k) / d ** 0.5, dim=1)
    msg = torch.einsum("lsh,shd->lhd", a, v).reshape(L, C)
    msg = layer_norm(F.linear(msg, sd[p + ".merge.weight"]), sd, p + ".norm1")
    msg = F.linear(F.relu(F.linear(torch.cat([x, msg], dim=1), sd[p + ".mlp.0.weight"])), sd[p + ".mlp.2.weight"])
    return x + layer_norm(msg, sd, p + ".norm2")


def log_sinkhorn(scores: Tensor, bin_score: Tensor, iters: int) -> Tensor:
    """matching_head.py:259-306 with all rows/cols valid -> [n1+1, n2+1]."""
    n1, n2 = scores.shape
    Z = torch.cat([torch.cat([scores, bin_score.expand(n1, 1)], -1), bin_score.expand(1, n2 + 1)], 0)
    norm = -torch.log(torch.tensor(float(n1 + n2)))
    log_mu = torch.cat([norm.expand(n1), (math.log(n2) + norm).view(1)])
    log_nu = torch.cat([norm.expand(n2), (math.log(n1) + norm).view(1)])
    u, v = torch.zeros_like(log_mu), torch.zeros_like(log_nu)
    for _ in range(iters):
        u = log_mu - torch.logsumexp(Z + v.unsqueeze(0), dim=1)
        v = log_nu - torch.logsumexp(Z + u.unsqueeze(1), dim=0)
    return Z + u.unsqueeze(1) + v.unsqueeze(0) - norm


def matcher_scores(sd, app1: Tensor, app2: Tensor, cam7: Tensor, planes1: Tensor, planes2: Tensor,
                   cfg: OracleConfig, p: str = "matching_head") -> Tensor:
    """Pre-Sinkhorn score matrix [n1,n2] (matching_head.py:75-119).  cam7 = (t[3], q[4])."""
    ang, off = _geometric_dists(planes1, planes2, cam7[3:], cam7[:3], 1e-10, 5.0)
    w, b = sd[p + ".planeApp_proj.weight"][:, :, 0], sd[p + ".planeApp_proj.bias"]
    f0, f1 = F.linear(app1, w, b), F.linear(app2, w, b)
    for i in range(18):                                                      # gnn.py:128-134
        lp = f"{p}.gnn.layers.{i}"
        if i % 2 == 0:
            f0, f1 = gnn_layer(sd, lp, f0, f0), gnn_layer(sd, lp, f1, f1)
        else:
            f0 = gnn_layer(sd, lp, f0, f1)
            f1 = gnn_layer(sd, lp, f1, f0)
    w, b = sd[p + ".planeDesc_proj.weight"][:, :, 0], sd[p + ".planeDesc_proj.bias"]
    d0, d1 = F.linear(f0, w, b), F.linear(f1, w, b)
    return (d0 @ d1.T) / 256 ** 0.5 - off / cfg.offset_multiplier - ang / cfg.normal_multiplier


def matcher(sd, app1, app2, cam7, planes1, planes2, cfg: OracleConfig) -> Tensor:
    s = matcher_scores(sd, app1, app2, cam7, planes1, planes2, cfg)
    return log_sinkhorn(s, sd["matching_head.bin_score"], cfg.sinkhorn_iterations)


def assignment_matrix(log_scores_padded: Tensor, thr: float) -> Tensor:
    """Mutual nearest neighbour + exp(score) > thr (camera_modules.py:15-34) -> binary [n1,n2]."""
    s = log_scores_padded[:-1, :-1]
    v0, i0 = s.max(1)
    i1 = s.max(0).indices
    mutual0 = torch.arange(s.shape[0]) == i1[i0]
    valid0 = mutual0 & (torch.where(mutual0, v0.exp(), v0.new_zeros(())) > thr)
    A = torch.zeros_like(s)
    rows = torch.arange(s.shape[0])[valid0]
    A[rows, i0[valid0]] = 1.0
    return A


# ======================================================================================
# a12/a13/a14 neural one-plane RANSAC  (camera_head.py:512-640, 925-1115, 1352-1425)
# ======================================================================================
def geo_sequence(planes1, planes2, A, nq, rot=None, tran=None):
    """Matched plane pairs in row-major nonzero order, zero-padded to nq rows -> ([nq,6], m)."""
    idx = torch.nonzero(A)
    m = idx.shape[0]
    p1, p2 = planes1[idx[:, 0]], planes2[idx[:, 1]]
    if rot is not None:
        p1, p2 = warp_planes(p1, rot, tran), flip_planes(p2)
    seq = torch.zeros(nq, 6)
    seq[:m] = torch.cat([p1, p2], -1)
    return seq, m


def ransac_refine(sd, init_trans_feat, init_rot_feat, geo_global, geo_local, sig_seq, m: int,
                  init_trans, init_rot, cfg: OracleConfig, p: str = "camera_head_list.0"):
    """__inference_PlaneCamRefHead for one pair (camera_head.py:925-1115).
    feats [256]; geo_* [nq,6]; sig_seq [nq,1]; init_trans [3]; init_rot [4]."""
    nq = geo_global.shape[0]
    src = geo_global if cfg.warp_plane_in_cam_ref else geo_local
    g0, g1 = src[:, :3], src[:, 3:]
    o0, o1 = g0.norm(dim=-1, keepdim=True), g1.norm(dim=-1, keepdim=True)
    n0, n1 = g0 / (o0 + 1e-10), g1 / (o1 + 1e-10)
    if cfg.warp_plane_in_cam_ref:
        o0, n0 = o0 * sig_seq, n0 * sig_seq
    geo = mlp(torch.cat((n0, o0, n1, o1), -1), sd, p + ".geo_encoder")
    s1 = mlp(geo, sd, p + ".geo_proj_s1")
    f_rot = mlp(s1, sd, p + ".decoder_rot")
    s2 = mlp(torch.cat([s1, f_rot], -1), sd, p + ".geo_proj_s2")
    f_tran = mlp(s2, sd, p + ".decoder_tran")
    if m == 0:                                                              # :964-969
        return {"pred_trans": init_trans, "pred_rot": init_rot, "pred_trans_avg": init_trans,
                "pred_rot_avg": init_rot}
    lin = lambda x, n: F.linear(x, sd[f"{p}.{n}.weight"], sd[f"{p}.{n}.bias"])
    mask = torch.zeros(nq + 1, nq)
    mask[:m + 1, :m] = 1.0
    fused_rot = F.relu(mlp(torch.cat((init_rot_feat.expand(nq, -1), f_rot), -1), sd, p + ".decoder_rot2"))
    fused_tran = F.relu(mlp(torch.cat((init_trans_feat.expand(nq, -1), f_tran), -1), sd, p + ".decoder_tran2"))
    rots_all = torch.cat([init_rot.view(1, 4), F.normalize(lin(fused_rot, "rots"), dim=-1)], 0)   # nq+1,4
    trans_all = torch.cat([init_trans.view(1, 3), lin(fused_tran, "trans")], 0)                     # nq+1,3
    pl1 = flip_planes(geo_local[:, 3:]).unsqueeze(0).expand(nq + 1, -1, -1)
    pl0_r = warp_planes(geo_local[:, :3].unsqueeze(0).expand(nq + 1, -1, -1), rots_all, torch.zeros(nq + 1, 3))
    nrm0, nrm1 = F.normalize(pl0_r, dim=-1), F.normalize(pl1, dim=-1)
    ang = torch.acos(torch.clamp((nrm0 * nrm1).sum(-1), -1.0, 1.0)) / np.pi * 180.0
    d_normal = (nrm0 - nrm1).norm(dim=-1) * mask
    d_normal_sum = d_normal.sum(-1)
    sc = lin(mlp(torch.exp(-d_normal) * mask, sd, p + ".normal_score_proj"), "rot_score_reg")      # nq+1,1
    score_rot = torch.zeros_like(sc)
    score_rot[:m + 1] = sc[:m + 1].softmax(0)
    pl0_rt = warp_planes(geo_local[:, :3].unsqueeze(0).expand(nq + 1, -1, -1), rots_all, trans_all)
    off0, off1 = pl0_rt.norm(dim=-1), pl1.norm(dim=-1)
    ntn = (F.normalize(pl0_rt, dim=-1) * nrm1).sum(-1)
    d_off = torch.where(ntn < 0, (off0 + off1).abs(), (off0 - off1).abs())
    d_l2 = (pl0_rt - pl1).norm(dim=-1)
    d_l2_sum = (d_l2 * mask).sum(-1)
    st = lin(mlp(torch.exp(-(d_l2 * mask)) * mask, sd, p + ".param_score_proj"), "trans_score_reg")
    score_tran = torch.zeros_like(st)
    score_tran[:m + 1] = st[:m + 1].softmax(0)
    avg = mask[:, 0:1] / (mask[:, 0:1].sum() + 1e-10)
    feats_t = torch.cat((init_trans_feat.view(1, -1), fused_tran), 0)
    feats_r = torch.cat((init_rot_feat.view(1, -1), fused_rot), 0)
    if m > 1:                                                               # :1052-1063
        ft_avg, fr_avg = (feats_t * avg).sum(0), (feats_r * avg).sum(0)
    else:
        ft_avg = (fused_tran * avg[1:] / avg[1:].sum()).sum(0)
        fr_avg = (fused_rot * avg[1:] / avg[1:].sum()).sum(0)
    rot_avg = F.normalize(lin(fr_avg.view(1, -1), "rots"), dim=-1)[0]
    tran_avg = lin(ft_avg.view(1, -1), "trans")[0]
    if m <= 1:                                                              # :1068-1075
        return {"pred_trans": tran_avg, "pred_rot": rot_avg, "pred_trans_avg": tran_avg, "pred_rot_avg": rot_avg}
    if cfg.out_cam_type == "avg-all":
        tran_f, rot_f = tran_avg, rot_avg
    elif cfg.out_cam_type == "soft":                                        # :1082-1087
        rot_f = F.normalize(lin((feats_r * score_rot).sum(0).view(1, -1), "rots"), dim=-1)[0]
        tran_f = lin((feats_t * score_tran).sum(0).view(1, -1), "trans")[0]
    elif cfg.out_cam_type == "min-cost":
        rot_f = rots_all[int(d_normal_sum[:m + 1].argmin())]
        tran_f = trans_all[int(d_l2_sum[:m + 1].argmin())]
    elif cfg.out_cam_type == "max-score":
        rot_f = rots_all[int(score_rot[:m + 1, 0].argmax())]
        tran_f = trans_all[int(score_tran[:m + 1, 0].argmax())]
    else:
        raise ValueError(cfg.out_cam_type)
    return {"pred_trans": tran_f, "pred_rot": rot_f, "pred_trans_avg": tran_avg, "pred_rot_avg": rot_avg,
            "all_pred_trans": trans_all[:m + 1], "all_pred_rots": rots_all[:m + 1],
            "score_soft_rot": score_rot[:m + 1], "score_soft_offset": score_tran[:m + 1],
            "l2_dist": d_l2[:m + 1, :m], "normal_dist": ang[:m + 1, :m], "offset_dist": d_off[:m + 1, :m]}


def camera_pose_loss(est_pose: Tensor, gt_pose: Tensor):
    """CameraPoseLoss.forward, reduce=True, no mask (camera_modules.py:355-365): poses are [B,7] = trans | quaternion."""
    l_x = (gt_pose[:, 0:3] - est_pose[:, 0:3]).norm(dim=1).mean()
    l_q = (F.normalize(gt_pose[:, 3:], dim=1) - F.normalize(est_pose[:, 3:], dim=1)).norm(dim=1).mean()
    return l_x, l_q


def ransac_refine_train(sd, init_trans_feat, init_rot_feat, geo_global, geo_local, sig_seq, ms, init_trans, init_rot, gt_pose,
                        cfg: OracleConfig, suffix: str = "", weight: float = 1.0, p: str = "camera_head_list.0"):
    """__forward_PlaneCamRefHead, the TRAINING-side twin of ransac_refine, for a batch (camera_head.py:737-923): scores are
    clamped to [0.01, 0.9] and renormalised (:816-818, :852-854), the average pose uses the per-plane features only (:862-865),
    the soft pose is always the prediction, and the seven refinement losses are returned (:883-921).
    feats [B,256]; geo_* [B,nq,6]; sig_seq [B,nq,1]; ms: list of B ints (each >= 1); init_trans [B,3]; init_rot [B,4];
    gt_pose [B,7] (trans | quaternion).  Returns (losses, pred_cam) with the reference's keys."""
    B, nq, _ = geo_global.shape
    src = geo_global if cfg.warp_plane_in_cam_ref else geo_local
    g0, g1 = src[..., :3], src[..., 3:]
    o0, o1 = g0.norm(dim=-1, keepdim=True), g1.norm(dim=-1, keepdim=True)
    n0, n1 = g0 / (o0 + 1e-10), g1 / (o1 + 1e-10)
    if cfg.warp_plane_in_cam_ref:
        o0, n0 = o0 * sig_seq, n0 * sig_seq
    geo = mlp(torch.cat((n0, o0, n1, o1), -1), sd, p + ".geo_encoder")
    s1 = mlp(geo, sd, p + ".geo_proj_s1")
    f_rot = mlp(s1, sd, p + ".decoder_rot")
    s2 = mlp(torch.cat([s1, f_rot], -1), sd, p + ".geo_proj_s2")
    f_tran = mlp(s2, sd, p + ".decoder_tran")
    lin = lambda x, n: F.linear(x, sd[f"{p}.{n}.weight"], sd[f"{p}.{n}.bias"])
    mask = torch.zeros(B, nq + 1, nq)
    for b, m in enumerate(ms):
        mask[b, :m + 1, :m] = 1.0
    live = mask[:, :, 0:1]                                                   # [B,nq+1,1]: hypothesis h <= m (and m >= 1)
    fused_rot = F.relu(mlp(torch.cat((init_rot_feat.unsqueeze(1).expand(-1, nq, -1), f_rot), -1), sd, p + ".decoder_rot2"))
    fused_tran = F.relu(mlp(torch.cat((init_trans_feat.unsqueeze(1).expand(-1, nq, -1), f_tran), -1), sd, p + ".decoder_tran2"))
    rots_all = torch.cat([init_rot.unsqueeze(1), F.normalize(lin(fused_rot, "rots"), dim=-1)], 1)      # B,nq+1,4
    trans_all = torch.cat([init_trans.unsqueeze(1), lin(fused_tran, "trans")], 1)                        # B,nq+1,3
    pl0 = geo_local[:, :, :3].unsqueeze(1).expand(-1, nq + 1, -1, -1)                                    # B,nq+1,nq,3
    pl1 = flip_planes(geo_local[:, :, 3:]).unsqueeze(1).expand(-1, nq + 1, -1, -1)
    warp = lambda t: torch.stack([warp_planes(pl0[b], rots_all[b], t[b]) for b in range(B)])
    pl0_r = warp(torch.zeros(B, nq + 1, 3))
    nrm0, nrm1 = F.normalize(pl0_r, dim=-1), F.normalize(pl1, dim=-1)
    ang = torch.acos(torch.clamp((nrm0 * nrm1).sum(-1), -1.0, 1.0)) / np.pi * 180.0
    d_normal = (nrm0 - nrm1).norm(dim=-1) * mask
    sc = lin(mlp(torch.exp(-d_normal) * mask, sd, p + ".normal_score_proj"), "rot_score_reg")           # B,nq+1,1

    def clamp_renorm(raw):                                                   # :811-818
        s = torch.zeros_like(raw)
        for b, m in enumerate(ms):
            s[b, :m + 1] = raw[b, :m + 1].softmax(0)
        s = torch.clamp(s, max=0.9, min=0.01) * live
        return s / (s.sum(dim=1, keepdim=True) + 1e-10)

    score_rot = clamp_renorm(sc)
    pl0_rt = warp(trans_all)
    off0, off1 = pl0_rt.norm(dim=-1), pl1.norm(dim=-1)
    ntn = (F.normalize(pl0_rt, dim=-1) * nrm1).sum(-1)
    d_off = torch.where(ntn < 0, (off0 + off1).abs(), (off0 - off1).abs())
    d_l2 = (pl0_rt - pl1).norm(dim=-1)                                       # B,nq+1,nq (unmasked: the loss reads its diagonal)
    st = lin(mlp(torch.exp(-(d_l2 * mask)) * mask, sd, p + ".param_score_proj"), "trans_score_reg")
    score_tran = clamp_renorm(st)
    avg = live / (live.sum(dim=1, keepdim=True) + 1e-10)                     # :858-861
    w_avg = avg[:, 1:] / avg[:, 1:].sum(dim=1, keepdim=True)
    ft_avg, fr_avg = (fused_tran * w_avg).sum(1), (fused_rot * w_avg).sum(1)
    rot_avg, tran_avg = F.normalize(lin(fr_avg, "rots"), dim=-1), lin(ft_avg, "trans")
    feats_t = torch.cat((init_trans_feat.unsqueeze(1), fused_tran), 1)
    feats_r = torch.cat((init_rot_feat.unsqueeze(1), fused_rot), 1)
    rot_soft = F.normalize(lin((feats_r * score_rot).sum(1), "rots"), dim=-1)
    tran_soft = lin((feats_t * score_tran).sum(1), "trans")
    m0 = ms[0]
    pred_cam = {"pred_trans": tran_soft, "pred_rot": rot_soft, "pred_trans_avg": tran_avg, "pred_rot_avg": rot_avg,
                "all_pred_trans": trans_all[0:1, :m0 + 1], "all_pred_rots": rots_all[0:1, :m0 + 1],
                "score_soft_rot": score_rot[0:1, :m0 + 1], "score_soft_offset": score_tran[0:1, :m0 + 1],
                "l2_dist": d_l2[0:1, :m0 + 1, :m0], "normal_dist": ang[0:1, :m0 + 1, :m0], "offset_dist": d_off[0:1, :m0 + 1, :m0]}
    # ---- losses (:883-921)
    l_t_avg, l_r_avg = camera_pose_loss(torch.cat((tran_avg, rot_avg), -1), gt_pose)
    l_t_soft, l_r_soft = camera_pose_loss(torch.cat((tran_soft, rot_soft), -1), gt_pose)
    bi = torch.arange(B)
    rot_err = (F.normalize(gt_pose[:, 3:].unsqueeze(1), dim=-1) - F.normalize(rots_all, dim=-1)).norm(dim=-1)
    rot_err = rot_err.masked_fill(live[:, :, 0] < 0.5, 1e10)
    l_rot_idx = (1.0 - score_rot[:, :, 0][bi, rot_err.argmin(dim=-1)]).abs().mean()
    tr_err = (gt_pose[:, :3].unsqueeze(1) - trans_all).norm(dim=-1).masked_fill(live[:, :, 0] < 0.5, 1e10)
    l_tr_idx = (1.0 - score_tran[:, :, 0][bi, tr_err.argmin(dim=-1)]).abs().mean()
    l_param = sum(torch.diag(d_l2[b, 1:]).sum() / ms[b] for b in range(B)) / B
    losses = {f"loss_tran_planeAvgReg_{suffix}": l_t_avg * weight, f"loss_rot_planeAvgReg_{suffix}": l_r_avg * weight,
              f"loss_tran_planeSoftReg_{suffix}": l_t_soft * weight, f"loss_rot_planeSoftReg_{suffix}": l_r_soft * weight,
              f"loss_rotIdx_{suffix}": l_rot_idx * 0.01 * weight, f"loss_transIdx_{suffix}": l_tr_idx * 0.02 * weight,
              f"loss_paramL2_dist_{suffix}": l_param * 0.1 * weight}
    return losses, pred_cam


def _refine_train_from_assignment(sd, planes1, planes2, A, init_trans, init_rot, trans_feat, rot_feat, gt_pose, cfg, suffix, weight, p):
    """forawrd_refineLoop (camera_head.py:346-398) for a batch given as padded planes [B,nq,3] + assignment [B,nq,nq]."""
    B, nq = A.shape[0], cfg.num_queries
    gl, gg, sg, ms = [], [], [], []
    for b in range(B):
        l, m = geo_sequence(planes1[b], planes2[b], A[b], nq)
        g, _ = geo_sequence(planes1[b], planes2[b], A[b], nq, init_rot[b], init_trans[b])
        a, _ = geo_sequence(planes1[b], planes2[b], A[b], nq, init_rot[b], torch.zeros(3))
        gl.append(l); gg.append(g); ms.append(m)
        sg.append((((g[:, 0:1] * a[:, 0:1]) >= 0).float() - 0.5) * 2.0)
    return ransac_refine_train(sd, trans_feat, rot_feat, torch.stack(gg), torch.stack(gl), torch.stack(sg), ms, init_trans, init_rot,
                               gt_pose, cfg, suffix=suffix, weight=weight, p=p)


def camera_head_train(sd, feats1, feats2, gt_planes1, gt_planes2, gt_A, gt_pose, planes1, planes2, A, cfg: OracleConfig,
                      initial_cam_weight: float = 1.0, plane_cam_weight: float = 1.0, plane_cam_weight_predplane: float = 0.1,
                      rand_rot=None, rand_trans=None, p: str = "camera_head_list.0"):
    """PlaneCameraHead.forward in TRAINING mode, forward + losses only (camera_head.py:140-189 -> forward_withInitialCam_Joint
    :191-323, forward_withRandCam_Joint :325-344 with the random poses given), CAM_REC_ON and REFINE_ON set.  BatchNorm layers use
    their running statistics (the fixtures are taken with the sub-modules in eval mode: a loss evaluation, not an optimiser step).
    feats: dicts res2..res5 [B,C,H,W]; gt_planes* / planes* [B,nq,3] zero-padded; gt_A / A [B,nq,nq] (ground-truth correspondences over
    the GT planes / over the predicted planes); gt_pose [B,7].  Returns (losses, trans_list, rot_list)."""
    losses = {}
    trans0, rot0, tf0, rf0, _ = pixel_pose_net(sd, feats1, feats2, p)       # :642-683 (no sign flip in training)
    l_t, l_r = camera_pose_loss(torch.cat((trans0, rot0), -1), gt_pose)
    losses["loss_tran_pixelReg"], losses["loss_rot_pixelReg"] = l_t * initial_cam_weight, l_r * initial_cam_weight

    def rec_losses(trans_in, rot_in, suffix):                               # :685-735
        sig = ((rot_in[:, 0:1] >= 0.0).float() - 0.5) * 2.0
        rec_t, rec_r, rec_tf, rec_rf = aim_reembed(sd, trans_in, rot_in, p)
        out = {}
        if rot_in is not None:
            out["loss_rot" + suffix] = (F.normalize(rot_in * sig, dim=1) - rec_r).norm(dim=1).mean()
            out["loss_trans" + suffix] = ((trans_in + 1e-10) - rec_t).norm(dim=1).mean()
        return out, rec_t, rec_r, rec_tf, rec_rf

    l, rec_t, rec_r, rec_tf, rec_rf = rec_losses(trans0, rot0, "_initCamRec")
    losses.update(l)
    trans_list, rot_list = [trans0, rec_t], [rot0, rec_r]
    for sfx, pl1, pl2, AA, w in (("", gt_planes1, gt_planes2, gt_A, plane_cam_weight), ("_Aux", planes1, planes2, A, plane_cam_weight_predplane)):
        for name, it, ir, itf, irf in (("initCamRef", trans0, rot0, tf0, rf0), ("initRecCamRef", rec_t, rec_r, rec_tf, rec_rf)):
            ls, pr = _refine_train_from_assignment(sd, pl1, pl2, AA, it, ir, itf, irf, gt_pose, cfg, name + sfx, w, p)
            losses.update(ls)
            trans_list += [pr["pred_trans_avg"], pr["pred_trans"]]
            rot_list += [pr["pred_rot_avg"], pr["pred_rot"]]
    if rand_rot is not None:                                                # :325-344: AIM on random poses (given, not drawn)
        l, _, _, _, _ = rec_losses(rand_trans, rand_rot, "_randCamRecLBS_N1")
        losses.update(l)
    return losses, trans_list, rot_list


def camera_head(sd, feats1, feats2, planes1, planes2, app1, app2, cfg: OracleConfig, forced_assignment=None):
    """PlaneCameraHead.inference_Joint for ONE pair (camera_head.py:400-640).
    feats: dict res2..res5 [1,C,H,W]; planes [n,3]; app [n,256].  Returns (cameras, assignments, aux).
    `forced_assignment` ([n1,n2] 0/1, benchmark K control only): the matcher still runs, its
    assignment is replaced."""
    p = "camera_head_list.0"
    trans0, rot0, tf0, rf0, _ = pixel_pose_net(sd, feats1, feats2, p)
    if rot0[0, 0] < 0:                                                      # :436-437
        rot0 = -rot0
    cams = {"camera_zero": (torch.zeros(3), torch.tensor([1.0, 0, 0, 0])), "camera_init": (trans0[0], rot0[0])}
    rec_t, rec_r, rec_tf, rec_rf = aim_reembed(sd, trans0, rot0, p)         # :451-465
    cams["camera_initRec"] = (rec_t[0], rec_r[0])
    cam7 = torch.cat([rec_t[0], rec_r[0]])
    log_scores = matcher(sd, app1, app2, cam7, planes1, planes2, cfg)       # :493-497
    A0 = assignment_matrix(log_scores, cfg.matching_score_threshold)         # :501
    if forced_assignment is not None:
        A0 = forced_assignment.to(A0.dtype)
    nq = cfg.num_queries
    geo_local, m = geo_sequence(planes1, planes2, A0, nq)
    geo_global, _ = geo_sequence(planes1, planes2, A0, nq, rec_r[0], rec_t[0])
    geo_aux, _ = geo_sequence(planes1, planes2, A0, nq, rec_r[0], torch.zeros(3))
    sig = (((geo_global[:, 0:1] * geo_aux[:, 0:1]) >= 0).float() - 0.5) * 2.0           # :568-569
    ref = ransac_refine(sd, rec_tf[0], rec_rf[0], geo_global, geo_local, sig, m, rec_t[0], rec_r[0], cfg, p)
    cams["camera_avgRef0"] = (ref["pred_trans_avg"], ref["pred_rot_avg"])
    cams["camera_softRef0"] = (ref["pred_trans"], ref["pred_rot"])
    cams["camera"] = (ref["pred_trans"], ref["pred_rot"])                   # sign NOT canonicalised (:596-601)
    r_soft = -ref["pred_rot"] if ref["pred_rot"][0] < 0 else ref["pred_rot"]
    ang, off = _geometric_dists(planes1, planes2, r_soft, ref["pred_trans"], 1e-4, 10.0)  # :605-621
    A1 = A0 * ((ang < 45.0) & (off < 1.0)).float()
    if "all_pred_trans" in ref:
        cams["camera_onePP"] = (ref["all_pred_trans"], ref["all_pred_rots"])
    assign = {"pred_assignment_beforeRef0": A0, "pred_assignment_afterRef0": A1, "pred_assignment": A1}
    aux = {"log_scores_padded": log_scores, "matched_num": m, "refine": ref, "geo_local": geo_local,
           "geo_global": geo_global, "sig_seq": sig}
    return cams, assign, aux


# ======================================================================================
# a1 end-to-end  (meta_arch/siamese_planeTR.py:338-473)
# ======================================================================================
def inference_single(sd, image: Tensor, cfg: OracleConfig):
    x = preprocess([image], cfg)
    feats = backbone(sd, x)
    out, qf = plane_head(sd, feats, cfg)
    sel = post_select(out["pred_logits"][0], out["pred_params"][0], out["pred_mask_logits"][0], qf[0], cfg,
                      image.shape[-2], image.shape[-1])
    out = dict(out, _query_feat=qf)
    return feats, out, sel


def force_k(out1: dict, qf1: Tensor, forced: dict, b: int):
    """Benchmark K control of SURVEY.md §8d for pair `b` (the CPU twin of PlaneTR_NopeSAC._force_k in the build): the K
    highest-scoring queries of view 1, view-2 appearance = permuted view-1 appearance + noise, planes / matches from `forced`
    ({"K", "planes" [2B,nq,3], "assignment" [B,nq,nq], "perm" [B,K], "noise" [B,K,256]})."""
    K = forced["K"]
    B = forced["perm"].shape[0]
    score = out1["pred_logits"][0, :, 0] - out1["pred_logits"][0, :, 1]
    idx = torch.topk(score, K).indices.sort().values
    f1 = qf1[0][idx]
    f2 = f1[forced["perm"][b]] + forced["noise"][b]
    return (forced["planes"][b, :K], forced["planes"][B + b, :K], f1, f2, forced["assignment"][b, :K, :K])


def inference(sd, batched_inputs: List[dict], cfg: Optional[OracleConfig] = None, forced: Optional[dict] = None) -> List[dict]:
    """Per pair (the reference asserts batch 1; we simply loop).  Output dict keys as SURVEY §8 a1."""
    cfg = cfg or OracleConfig()
    results = []
    with torch.no_grad():
        for b, item in enumerate(batched_inputs):
            f1, o1, s1 = inference_single(sd, item["0"]["image"], cfg)
            f2, _, s2 = inference_single(sd, item["1"]["image"], cfg)
            if forced is not None:
                p1, p2, a1, a2, A = force_k(o1, o1["_query_feat"], forced, b)
                cams, assign, aux = camera_head(sd, f1, f2, p1, p2, a1, a2, cfg, forced_assignment=A)
            else:
                cams, assign, aux = camera_head(sd, f1, f2, s1["pred_plane"], s2["pred_plane"],
                                                s1["pred_plane_feats"][0], s2["pred_plane_feats"][0], cfg)
            res = {"0": s1, "1": s2, "pred_aff": None, "depth": {"0": None, "1": None}}
            for k, (t, r) in cams.items():
                res[k] = {"tran": t.numpy(), "rot": r.numpy()}
            res.update({k: v.unsqueeze(0) if False else v for k, v in assign.items()})
            res["_aux"] = aux
            results.append(res)
    return results
