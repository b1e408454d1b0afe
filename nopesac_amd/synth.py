"""Name-seeded synthetic checkpoint + synthetic image pairs.

No pretrained weights exist offline (configs/inference_mp3d.yaml:24 points at a Google-Drive
file), so parity and benchmarks run on synthetic weights that every side (the reference import,
the CPU oracle, the HIP product) can materialise from the *key list alone*:

    g = torch.Generator().manual_seed(crc32(key));  tensor = rule(key, shape, g)

The key list is the reference's state-dict contract (942 tensors without
`criterion.empty_weight`), enumerated here by construction from the config
(`MODEL.SEM_SEG_HEAD.NUM_OBJECT_QUERIES` is the only size that varies) and checked against the
imported reference's `state_dict()` by oracle/gen_golden.py.
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict

import torch

RES_STAGES = (("res2", 3, 64, 256), ("res3", 4, 128, 512), ("res4", 6, 256, 1024), ("res5", 3, 512, 2048))


def _bn(spec, prefix, c, tracked=False):
    for n in ("weight", "bias", "running_mean", "running_var"):
        spec[f"{prefix}.{n}"] = (c,)
    if tracked:
        spec[f"{prefix}.num_batches_tracked"] = ()


def _mlp(spec, prefix, dims):
    for i, (a, b) in enumerate(zip(dims[:-1], dims[1:])):
        spec[f"{prefix}.layers.{i}.weight"] = (b, a)
        spec[f"{prefix}.layers.{i}.bias"] = (b,)


def _mha(spec, prefix, d=256):
    spec[f"{prefix}.in_proj_weight"] = (3 * d, d)
    spec[f"{prefix}.in_proj_bias"] = (3 * d,)
    spec[f"{prefix}.out_proj.weight"] = (d, d)
    spec[f"{prefix}.out_proj.bias"] = (d,)


def _ln(spec, prefix, d=256):
    spec[f"{prefix}.weight"] = (d,)
    spec[f"{prefix}.bias"] = (d,)


def state_dict_spec(num_queries: int = 50) -> "OrderedDict[str, tuple]":
    """Ordered {key: shape} for the whole model (reference naming, Appendix B)."""
    s: "OrderedDict[str, tuple]" = OrderedDict()
    # ---- backbone (detectron2 ResNet-50 naming)
    s["backbone.stem.conv1.weight"] = (64, 3, 7, 7)
    _bn(s, "backbone.stem.conv1.norm", 64)
    cin = 64
    for name, n, cmid, cout in RES_STAGES:
        for i in range(n):
            p = f"backbone.{name}.{i}"
            if cin != cout:
                s[f"{p}.shortcut.weight"] = (cout, cin, 1, 1)
                _bn(s, f"{p}.shortcut.norm", cout)
            s[f"{p}.conv1.weight"] = (cmid, cin, 1, 1)
            _bn(s, f"{p}.conv1.norm", cmid)
            s[f"{p}.conv2.weight"] = (cmid, cmid, 3, 3)
            _bn(s, f"{p}.conv2.norm", cmid)
            s[f"{p}.conv3.weight"] = (cout, cmid, 1, 1)
            _bn(s, f"{p}.conv3.norm", cout)
            cin = cout
    # ---- PlaneTR head (planeTR_head.py:74-109)
    h = "sem_seg_head"
    s[f"{h}.input_proj.weight"] = (256, 2048, 1, 1)
    s[f"{h}.input_proj.bias"] = (256,)
    for i in range(6):
        p = f"{h}.context_SA.layers.{i}"
        _mha(s, f"{p}.self_attn")
        s[f"{p}.linear1.weight"] = (1024, 256); s[f"{p}.linear1.bias"] = (1024,)
        s[f"{p}.linear2.weight"] = (256, 1024); s[f"{p}.linear2.bias"] = (256,)
        _ln(s, f"{p}.norm1"); _ln(s, f"{p}.norm2")
    _ln(s, f"{h}.context_SA.norm")
    s[f"{h}.query_embed.weight"] = (num_queries, 256)
    for i in range(6):
        p = f"{h}.context2plane_decoder.layers.{i}"
        _mha(s, f"{p}.self_attn"); _mha(s, f"{p}.multihead_attn")
        s[f"{p}.linear1.weight"] = (1024, 256); s[f"{p}.linear1.bias"] = (1024,)
        s[f"{p}.linear2.weight"] = (256, 1024); s[f"{p}.linear2.bias"] = (256,)
        _ln(s, f"{p}.norm1"); _ln(s, f"{p}.norm2"); _ln(s, f"{p}.norm3")
    _ln(s, f"{h}.context2plane_decoder.norm")
    for nm, c in (("up_conv3", 256), ("up_conv2", 256), ("up_conv1", 256), ("c4_conv", 2048),
                  ("c3_conv", 1024), ("c2_conv", 512), ("c1_conv", 256), ("m_conv_dict.m4", 256)):
        s[f"{h}.top_down.{nm}.0.weight"] = (256, c, 1, 1)
        _bn(s, f"{h}.top_down.{nm}.1", 256, tracked=True)
    _mlp(s, f"{h}.plane_embedding", (256, 256, 256, 256))
    s[f"{h}.pixel_embedding.weight"] = (256, 256, 1, 1); s[f"{h}.pixel_embedding.bias"] = (256,)
    s[f"{h}.plane_prob.weight"] = (2, 256); s[f"{h}.plane_prob.bias"] = (2,)
    _mlp(s, f"{h}.plane_param", (256, 256, 256, 3))
    _mlp(s, f"{h}.plane_center", (256, 256, 256, 2))
    s[f"{h}.pixel_plane_center.weight"] = (2, 256, 1, 1); s[f"{h}.pixel_plane_center.bias"] = (2,)
    # ---- matching head (matching_head.py:30-41, gnn.py:46-71)
    m = "matching_head"
    s[f"{m}.bin_score"] = ()
    for i in range(18):
        p = f"{m}.gnn.layers.{i}"
        for nm in ("q_proj", "k_proj", "v_proj", "merge"):
            s[f"{p}.{nm}.weight"] = (256, 256)
        s[f"{p}.mlp.0.weight"] = (512, 512)
        s[f"{p}.mlp.2.weight"] = (256, 512)
        _ln(s, f"{p}.norm1"); _ln(s, f"{p}.norm2")
    s[f"{m}.planeDesc_proj.weight"] = (256, 256, 1); s[f"{m}.planeDesc_proj.bias"] = (256,)
    s[f"{m}.planeApp_proj.weight"] = (256, 256, 1); s[f"{m}.planeApp_proj.bias"] = (256,)
    # ---- camera head (camera_head.py:60-138, camera_modules.py:246-322)
    c = "camera_head_list.0"
    for nm, shp in (("adapter_1", (128, 512, 1, 1)), ("layer_1", (128, 128, 3, 3)),
                    ("adapter_2", (128, 1024, 1, 1)), ("layer_2", (128, 128, 3, 3)),
                    ("layer_3", (128, 2048, 3, 3))):
        s[f"{c}.pixel_decoder.{nm}.weight"] = shp
        s[f"{c}.pixel_decoder.{nm}.norm.weight"] = (128,)
        s[f"{c}.pixel_decoder.{nm}.norm.bias"] = (128,)
    s[f"{c}.pixel_decoder.mask_features.weight"] = (256, 128, 3, 3)
    s[f"{c}.pixel_decoder.mask_features.bias"] = (256,)
    for i in (0, 1, 3, 4, 6, 7):
        s[f"{c}.convs_backbone.{i}.0.weight"] = (256, 256, 3, 3)
        _bn(s, f"{c}.convs_backbone.{i}.1", 256, tracked=True)
    for br in ("convs_trans", "convs_rots"):
        for i in range(6):
            s[f"{c}.{br}.{i}.0.weight"] = (128, 300 if i == 0 else 128, 3, 3)
            _bn(s, f"{c}.{br}.{i}.1", 128, tracked=True)
    for nm in ("fc_trans", "fc_rots"):
        s[f"{c}.{nm}.weight"] = (256, 768); s[f"{c}.{nm}.bias"] = (256,)
    s[f"{c}.trans.weight"] = (3, 256); s[f"{c}.trans.bias"] = (3,)
    s[f"{c}.rots.weight"] = (4, 256); s[f"{c}.rots.bias"] = (4,)
    _mlp(s, f"{c}.rot_emb_proj", (4, 256, 256, 256, 256, 256, 256))
    _mlp(s, f"{c}.trans_emb_proj", (3, 256, 256, 256, 256, 256, 256))
    _mlp(s, f"{c}.geo_encoder", (8, 1024, 1024, 1024, 1024, 1024, 1024))
    _mlp(s, f"{c}.geo_proj_s1", (1024, 1024, 1024, 1024))
    _mlp(s, f"{c}.decoder_rot", (1024, 512, 512, 512, 512, 512, 256))
    _mlp(s, f"{c}.geo_proj_s2", (1280, 1024, 1024, 1024))
    _mlp(s, f"{c}.decoder_tran", (1024, 512, 512, 512, 512, 512, 256))
    _mlp(s, f"{c}.decoder_rot2", (512, 512, 512, 256))
    _mlp(s, f"{c}.decoder_tran2", (512, 512, 512, 256))
    _mlp(s, f"{c}.normal_score_proj", (num_queries, 128, 128, 64))
    s[f"{c}.rot_score_reg.weight"] = (1, 64); s[f"{c}.rot_score_reg.bias"] = (1,)
    _mlp(s, f"{c}.param_score_proj", (num_queries, 128, 128, 64))
    s[f"{c}.trans_score_reg.weight"] = (1, 64); s[f"{c}.trans_score_reg.bias"] = (1,)
    return s


def _is_norm_affine(key: str) -> bool:
    parts = key.split(".")
    owner = parts[-2]
    return owner.startswith("norm") or owner == "1" and "top_down" in key or (
        owner == "1" and ("convs_" in key))


def synth_tensor(key: str, shape: tuple) -> torch.Tensor:
    """Deterministic fp32 tensor for `key` (rule documented in the module docstring)."""
    g = torch.Generator().manual_seed(zlib.crc32(key.encode()))
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros((), dtype=torch.int64)
    if key.endswith("bin_score"):
        return torch.tensor(1.0)
    if leaf == "running_var":
        return 0.5 + torch.rand(shape, generator=g)
    if leaf == "running_mean":
        return 0.1 * torch.randn(shape, generator=g)
    if _is_norm_affine(key):
        if leaf == "weight":
            w = 1.0 + 0.1 * torch.randn(shape, generator=g)
            if key.startswith("backbone.") and ".conv3.norm." in key:
                w = 0.3 * w  # damp residual-branch growth through 16 bottlenecks
            if ".convs_backbone.7.1." in key:
                w = 0.3 * w  # keep the 300x300 correlation softmax of the pose net smooth (not arg-max-like)
            return w
        b = 0.05 * torch.randn(shape, generator=g)
        return 0.3 * b if ".convs_backbone.7.1." in key else b
    if key.endswith("query_embed.weight"):
        return 3.0 * torch.randn(shape, generator=g)
    if leaf in ("bias", "in_proj_bias"):
        return 0.02 * torch.randn(shape, generator=g)
    fan_in = 1
    for d in shape[1:]:
        fan_in *= d
    w = torch.randn(shape, generator=g) * (1.4 / math.sqrt(max(fan_in, 1)))
    # Gains that keep a random-weight DETR from collapsing every token/query onto one vector
    # (measured: without them the 50 plane queries differ by ~3e-4 and every discrete decision
    # downstream is a coin flip): sharper q/k, weaker residual branches, unsaturated mask logits.
    if leaf == "in_proj_weight":
        e = shape[0] // 3
        w[: 2 * e] *= 3.0
    elif key.endswith("out_proj.weight"):
        w *= 0.1
    elif key.endswith("linear2.weight"):
        w *= 0.2
    elif key.endswith("pixel_embedding.weight"):
        w *= 0.012
    elif key.endswith("planeDesc_proj.weight"):
        w *= 0.15
    return w


def synth_state_dict(num_queries: int = 50) -> "OrderedDict[str, torch.Tensor]":
    return OrderedDict((k, synth_tensor(k, shp)) for k, shp in state_dict_spec(num_queries).items())


def synth_image(seed: int, h: int = 480, w: int = 640) -> torch.Tensor:
    """uint8-valued fp32 RGB [3,h,w], the mapper contract (planercnn_transforms.py:225-227)."""
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 256, (3, h, w), generator=g).float()


def structured_image(seed: int, h: int = 480, w: int = 640, nrect: int = 24) -> torch.Tensor:
    """uint8-valued fp32 RGB [3,h,w] made of random flat rectangles + mild noise: unlike iid noise
    it gives spatially diverse features, so parity tests see more than one plane per view."""
    g = torch.Generator().manual_seed(seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi, (1,), generator=g))
    img = torch.randint(0, 256, (3, 1, 1), generator=g).float().expand(3, h, w).clone()
    for _ in range(nrect):
        y0, x0 = ri(0, max(h - h // 12, 1)), ri(0, max(w - w // 16, 1))
        hh, ww = ri(h // 12, h // 2), ri(w // 16, w // 2)
        img[:, y0:y0 + hh, x0:x0 + ww] = torch.randint(0, 256, (3, 1, 1), generator=g).float()
    img = img + torch.randint(-12, 13, (3, h, w), generator=g).float()
    return img.clamp(0, 255).round()


def synth_pair(pair_idx: int, h: int = 480, w: int = 640, structured: bool = False) -> dict:
    """One element of `batched_inputs`: seeds 1000+idx (view 0), 500000+idx."""
    out = {}
    gen = structured_image if structured else synth_image
    for v, base in (("0", 1000), ("1", 500000)):
        out[v] = {"image": gen(base + pair_idx, h, w), "image_id": f"p{pair_idx}_v{v}",
                  "file_name": f"synthetic/p{pair_idx}_v{v}.png", "height": h, "width": w}
    return out


# ---- seeded plane-pair generators (shared by bench.py's K control and the designed stage inputs of tests/golden_inputs.py)
from torch.nn import functional as F  # noqa: E402


def _g(seed: int) -> torch.Generator:
    return torch.Generator().manual_seed(seed)


def rand_unit_quat(g, w_nonneg=True) -> torch.Tensor:
    q = F.normalize(torch.randn(4, generator=g), dim=0)
    return -q if (w_nonneg and q[0] < 0) else q


def rand_planes(n: int, g) -> torch.Tensor:
    """n*d plane vectors [n,3], offsets in [0.5, 4]."""
    nrm = F.normalize(torch.randn(n, 3, generator=g), dim=-1)
    d = 0.5 + 3.5 * torch.rand(n, 1, generator=g)
    return nrm * d


def quat_to_rotmat(q):
    w, x, y, z = q.tolist()
    return torch.tensor([[1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * w * z, 2 * x * z + 2 * w * y],
                         [2 * x * y + 2 * w * z, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * w * x],
                         [2 * x * z - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x * x - 2 * y * y]])


def consistent_planes(n1: int, n2: int, n_common: int, g, noise=0.02):
    """Two plane sets related by a ground-truth pose on the first `n_common` (permuted) planes:
    warp(view-1 plane) ~= flip(view-2 plane), so geometric matching terms are informative.
    Returns planes1 [n1,3], planes2 [n2,3], perm (view-2 index of view-1 plane i, -1 if none), (t,q)."""
    q = rand_unit_quat(g)
    t = 0.4 * torch.randn(3, generator=g)
    R = quat_to_rotmat(q)
    flip = torch.tensor([1.0, -1.0, -1.0])
    planes1 = rand_planes(n1, g)
    planes2 = rand_planes(n2, g)
    perm2 = torch.randperm(n2, generator=g)[:n_common]
    idx1 = torch.randperm(n1, generator=g)[:n_common]
    perm = torch.full((n1,), -1, dtype=torch.long)
    for i1, i2 in zip(idx1.tolist(), perm2.tolist()):
        p = planes1[i1] * flip
        nrm = R @ (p / p.norm())
        d = p.norm() + (t * nrm).sum()
        glob = nrm * d + noise * torch.randn(3, generator=g)
        planes2[i2] = glob * flip
        perm[i1] = i2
    return planes1, planes2, perm, (t, q)


def make_forced(B: int, K: int, nq: int, device, seed: int) -> dict:
    """Device-resident K control tensors of the benchmark workload (K planes per view, K matches): consistent
    plane pairs under a random pose and a K-permutation, consumed by PlaneTR_NopeSAC._force_k and the oracle's `forced=`."""
    g = torch.Generator().manual_seed(seed)
    planes = torch.zeros(2 * B, nq, 3)
    A = torch.zeros(B, nq, nq)
    perm = torch.zeros(B, K, dtype=torch.long)
    for b in range(B):
        p1, p2, pm, _ = consistent_planes(K, K, K, g, noise=0.02)
        planes[b, :K], planes[B + b, :K] = p1, p2
        inv = torch.empty(K, dtype=torch.long)
        inv[pm] = torch.arange(K)            # view-2 row j shows view-1 plane inv[j]
        perm[b] = inv
        A[b, torch.arange(K), pm] = 1.0
    noise = 0.01 * torch.randn(B, K, 256, generator=g)
    return {"K": K, "planes": planes.to(device), "assignment": A.to(device), "perm": perm.to(device), "noise": noise.to(device)}
