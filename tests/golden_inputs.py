"""Seeded generators of *designed* stage inputs, shared by oracle/gen_golden.py (which feeds them
to the imported reference and stores only the OUTPUTS under tests/golden/) and by the tests
(which regenerate the same inputs from the seed).  Pure data — no reference code involved.
"""
from __future__ import annotations

import math

import torch
from torch.nn import functional as F


from nopesac_amd.synth import _g, consistent_planes, quat_to_rotmat, rand_planes, rand_unit_quat  # noqa: E402,F401  (moved: bench.py uses them too)


def matcher_case(n1: int, n2: int, seed: int):
    """(app1 [n1,256], app2 [n2,256], cam7 [7]=(t,q), planes1, planes2): matched planes share
    appearance (+1% noise) and geometry; the rest are distractors."""
    g = _g(seed)
    nc = max(min(n1, n2) - (min(n1, n2) // 4), 1) if min(n1, n2) > 1 else 1
    planes1, planes2, perm, (t, q) = consistent_planes(n1, n2, nc, g)
    app1 = torch.randn(n1, 256, generator=g)
    app2 = torch.randn(n2, 256, generator=g)
    for i1 in range(n1):
        if perm[i1] >= 0:
            app2[perm[i1]] = app1[i1] + 0.01 * torch.randn(256, generator=g)
    # the matcher gets a perturbed pose, as it would from the pixel pose net
    qn = F.normalize(q + 0.03 * torch.randn(4, generator=g), dim=0)
    cam7 = torch.cat([t + 0.05 * torch.randn(3, generator=g), qn])
    return app1, app2, cam7, planes1, planes2


def refine_case(nq: int, m: int, seed: int, n1: int | None = None, n2: int | None = None):
    """Inputs of the one-plane RANSAC stage: planes, a binary assignment with exactly m ones
    (at most one per row/col), the re-embedded initial pose and its two 256-d features."""
    g = _g(seed)
    n1 = n1 if n1 is not None else max(m, 1) + (3 if m + 3 <= nq else 0)
    n2 = n2 if n2 is not None else max(m, 1) + (1 if m + 1 <= nq else 0)
    planes1, planes2, perm, (t, q) = consistent_planes(n1, n2, m, g, noise=0.05) if m > 0 else (
        rand_planes(n1, g), rand_planes(n2, g), torch.full((n1,), -1, dtype=torch.long),
        (0.4 * torch.randn(3, generator=g), rand_unit_quat(g)))
    A = torch.zeros(n1, n2)
    for i1 in range(n1):
        if perm[i1] >= 0:
            A[i1, perm[i1]] = 1.0
    init_rot = F.normalize(q + 0.05 * torch.randn(4, generator=g), dim=0)
    if init_rot[0] < 0:
        init_rot = -init_rot
    init_trans = t + 0.1 * torch.randn(3, generator=g)
    trans_feat = F.relu(torch.randn(256, generator=g))
    rot_feat = F.relu(torch.randn(256, generator=g))
    return {"planes1": planes1, "planes2": planes2, "A": A, "init_trans": init_trans, "init_rot": init_rot,
            "trans_feat": trans_feat, "rot_feat": rot_feat}


def gt_pose_case(B: int, seed: int):
    """Ground-truth relative poses [B,7] = trans | quaternion for the training-side refine twin (the quaternions are NOT unit
    length: the losses normalise them, camera_modules.py:363)."""
    g = _g(1000 + seed)
    q = torch.stack([rand_unit_quat(g) for _ in range(B)]) * (0.7 + 0.6 * torch.rand(B, 1, generator=g))
    return torch.cat([0.6 * torch.randn(B, 3, generator=g), q], dim=1)


def camera_train_case(nq: int, ms, seed: int):
    """Inputs of the camera head's training forward (camera_head.py:140-323): backbone maps of a batch, GT planes + GT
    correspondences per pair, noisy 'predicted' planes with the same correspondences, GT poses, and the random poses of the AIM's
    LBS losses."""
    cases = [refine_case(nq, m, seed + 10 * b) for b, m in enumerate(ms)]
    B = len(cases)
    g = _g(2000 + seed)
    pad = lambda t: torch.cat([t, torch.zeros(nq - t.shape[0], 3)], 0)
    A = torch.zeros(B, nq, nq)
    for b, c in enumerate(cases):
        A[b, : c["A"].shape[0], : c["A"].shape[1]] = c["A"]
    gp1, gp2 = torch.stack([pad(c["planes1"]) for c in cases]), torch.stack([pad(c["planes2"]) for c in cases])
    noisy = lambda P: P + 0.03 * torch.randn(P.shape, generator=g) * (P.abs().sum(-1, keepdim=True) > 0)
    rr = F.normalize(torch.randn(2 * B, 4, generator=g), dim=1)
    return {"feats1": feature_maps(seed + 100, batch=B), "feats2": feature_maps(seed + 200, batch=B), "gt_planes1": gp1, "gt_planes2": gp2,
            "gt_A": A, "planes1": noisy(gp1), "planes2": noisy(gp2), "A": A.clone(), "gt_pose": gt_pose_case(B, seed),
            "n1": torch.tensor([c["planes1"].shape[0] for c in cases], dtype=torch.int32),
            "n2": torch.tensor([c["planes2"].shape[0] for c in cases], dtype=torch.int32),
            "rand_rot": rr, "rand_trans": (torch.rand(2 * B, 3, generator=g) - 0.5) * 5.0}


def _blob_field(h, w, g, cells=6):
    """Smooth random field in [-1,1] (bilinear up-sampling of a coarse random grid)."""
    coarse = torch.rand(1, 1, cells, cells + 2, generator=g) * 2 - 1
    return F.interpolate(coarse, size=(h, w), mode="bilinear", align_corners=True)[0, 0]


def postselect_case(kind: str, seed: int, nq: int = 50, h: int = 120, w: int = 160):
    """(pred_logits [nq,2], pred_params [nq,3], mask_logits [nq,h,w], query_feat [nq,256]).
    kinds: 'multi' (several disjoint planes), 'none_pass' (no query beats the score threshold ->
    arg-max fallback), 'all_overlap_rejected' (every candidate loses its area -> max-overlap
    fallback), 'full' (every query is a plane)."""
    g = _g(seed)
    logits = torch.zeros(nq, 2)
    logits[:, 1] = 2.0 + torch.rand(nq, generator=g)           # non-plane by default
    params = rand_planes(nq, g)
    feat = torch.randn(nq, 256, generator=g)
    mask = -6.0 + 0.5 * torch.randn(nq, h, w, generator=g)
    n_on = {"multi": min(9, nq), "none_pass": 4, "all_overlap_rejected": 5, "full": nq}[kind]
    on = torch.randperm(nq, generator=g)[:n_on].sort().values
    # label map: vertical/horizontal stripes of unequal width with wavy borders
    lab = torch.zeros(h, w, dtype=torch.long)
    cols = max(int(math.ceil(math.sqrt(n_on * w / h))), 1)
    rows = int(math.ceil(n_on / cols))
    yy, xx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    wob = (_blob_field(h, w, g) * 4).round().long()
    lab = (((yy + wob).clamp(0, h - 1) * rows) // h) * cols + ((xx + wob).clamp(0, w - 1) * cols) // w
    lab = lab.clamp(max=n_on - 1)
    for k, q in enumerate(on.tolist()):
        inside = (lab == k).float()
        mask[q] = (12.0 * inside - 6.0) + 0.8 * _blob_field(h, w, g) + 0.2 * torch.randn(h, w, generator=g)
        if kind != "none_pass":
            logits[q, 0] = 3.0 + 2.0 * torch.rand(1, generator=g).item()
            logits[q, 1] = 0.0
    if kind == "none_pass":
        logits[:, 0] = 1.0 + 0.5 * torch.rand(nq, generator=g)  # p0 < 0.5 everywhere
    if kind == "all_overlap_rejected":
        # every "on" query is positive over the whole image -> original area = H*W while its
        # owned area is ~1/n_on of that -> overlap << 0.6 for all
        for k, q in enumerate(on.tolist()):
            mask[q] = 3.0 + 2.0 * (lab == k).float() + 0.3 * _blob_field(h, w, g)
    return logits, params, mask, feat


def feature_maps(seed: int, h5: int = 15, w5: int = 20, batch: int = 1):
    """Random non-negative res2..res5 maps (post-ReLU statistics) at strides 4..32."""
    g = _g(seed)
    out = {}
    for name, c, s in (("res2", 256, 8), ("res3", 512, 4), ("res4", 1024, 2), ("res5", 2048, 1)):
        x = torch.randn(batch, c, h5 * s, w5 * s, generator=g)
        lowf = F.interpolate(torch.randn(batch, c, 4, 5, generator=g), size=(h5 * s, w5 * s), mode="bilinear",
                             align_corners=False)
        out[name] = F.relu(0.6 * x + 1.2 * lowf + 0.3)
    return out


def matching_eval_case(seed: int, h: int = 48, w: int = 64, n_pairs: int = 3):
    """Inputs of the plane-matching evaluator (mp3d_evaluation.py:746-849) as plain arrays: per pair, per view GT masks (a
    partition of the image into blobs), predicted masks (eroded / shifted copies of some GT blobs + a spurious one), the GT
    correspondences, and three predicted assignment matrices (perfect on the detected planes / one swapped match / empty)."""
    import numpy as np
    g = _g(seed)
    rng = np.random.default_rng(seed)
    pairs = []
    for pi in range(n_pairs):
        views = []
        for v in range(2):
            n_gt = 4 + int(torch.randint(0, 3, (1,), generator=g))
            yy, xx = np.mgrid[0:h, 0:w]
            cx, cy = rng.uniform(0, w, n_gt), rng.uniform(0, h, n_gt)
            lab = np.argmin((xx[None] - cx[:, None, None]) ** 2 + (yy[None] - cy[:, None, None]) ** 2, 0)
            gt = np.stack([lab == k for k in range(n_gt)])
            preds, src = [], []
            for k in range(n_gt):
                if rng.uniform() < 0.8:
                    m = np.roll(gt[k], int(rng.integers(-1, 2)), axis=int(rng.integers(0, 2)))
                    if rng.uniform() < 0.3:                      # a poor detection: IoU with its GT blob drops below 0.5
                        m = m & (xx % 3 == 0)
                    preds.append(m); src.append(k)
            preds.append((xx + yy) % 7 == 0); src.append(-1)     # spurious plane
            views.append({"gt": gt, "pred": np.stack(preds), "src": np.asarray(src)})
        n0, n1 = len(views[0]["src"]), len(views[1]["src"])
        ng = min(views[0]["gt"].shape[0], views[1]["gt"].shape[0])
        gt_corrs = [[k, (k * 2 + pi) % views[1]["gt"].shape[0]] for k in range(ng) if k % 3 != 2]
        corr = dict(map(tuple, gt_corrs))
        A_good = np.zeros((n0, n1), np.float32)
        for i, k in enumerate(views[0]["src"]):
            if k in corr:
                js = np.flatnonzero(views[1]["src"] == corr[k])
                if len(js):
                    A_good[i, js[0]] = 1.0
        A_swap = A_good.copy()
        rows = np.flatnonzero(A_good.sum(1))
        if len(rows) >= 2:
            A_swap[[rows[0], rows[1]]] = A_swap[[rows[1], rows[0]]]
        A_swap[n0 - 1, n1 - 1] = 1.0                             # the spurious planes matched to each other
        pairs.append({"views": views, "gt_corrs": gt_corrs, "pred_assignment": A_good, "pred_assignment_afterRef0": A_swap,
                      "pred_assignment_beforeRef0": np.zeros((n0, n1), np.float32)})
    return pairs
