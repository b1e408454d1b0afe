"""Shared helpers for the parity tests."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
LOOSE = ["TEST.OVERLAP_THRESHOLD", 0.0, "TEST.PLANE_SCORE_THRESHOLD", 0.5, "TEST.MATCHING_SCORE_THRESHOLD", 0.0,
         "TEST.MASK_PROB_THRESHOLD", 0.3]


def gold(name):
    return {k: torch.from_numpy(v) if v.dtype.kind in "fiub" else v for k, v in np.load(os.path.join(GOLD, name + ".npz")).items()}


def rel_err(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def abs_err(a, b):
    """max |a - b| - the ABSOLUTE gate SURVEY.md Appendix D / north_star ask for on camera.tran, camera.rot and pred_plane."""
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max()) if a.numel() else 0.0


def quat_abs_err(a, b):
    """abs_err up to the quaternion sign (q and -q are the same rotation; BASELINE.md: 'camera.rot up to sign')."""
    return min(abs_err(a, b), abs_err(a, -torch.as_tensor(b)))


def oracle_f64(sd, inputs, cfg, **kw):
    """The oracle evaluated in float64: the canonical value of the reference ALGORITHM.  The reference's own fp32 CPU result is
    not unique (oneDNN blocking, thread count): on these inputs it sits 2e-5 .. 8e-5 (absolute) away from the float64 evaluation
    in pred_plane - i.e. the 1e-4 absolute gate is at the reference's own rounding noise for outputs of magnitude ~2.  The e2e
    tests therefore gate pred_plane on |hip - f64| < 1e-4 AND |hip - cpu32| < 1e-4 + |cpu32 - f64| (triangle bound)."""
    from oracle import nopesac_oracle as O
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    inp64 = [{v: ({**p[v], "image": p[v]["image"].double()} if v in ("0", "1") else p[v]) for v in p} for p in inputs]
    if kw.get("forced") is not None:
        kw = dict(kw)
        kw["forced"] = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in kw["forced"].items()}
    old = torch.get_default_dtype()
    try:
        torch.set_default_dtype(torch.float64)
        return O.inference(sd64, inp64, cfg, **kw)
    finally:
        torch.set_default_dtype(old)


def loose_oracle_cfg(nq=50):
    from oracle.nopesac_oracle import OracleConfig
    return OracleConfig(num_queries=nq, overlap_threshold=0.0, plane_score_threshold=0.5, matching_score_threshold=0.0,
                        mask_prob_threshold=0.3)


_MODELS = {}


def make_model(device, overrides=(), nq=50, dtype="float32", config="inference_mp3d.yaml"):
    """PlaneTR_NopeSAC (HIP) with the name-seeded synthetic checkpoint, cached per configuration."""
    key = (str(device), tuple(overrides), nq, dtype, config)
    if key not in _MODELS:
        from nopesac_amd.config import get_cfg
        from nopesac_amd.registry import build_model
        from nopesac_amd.synth import synth_state_dict
        cfg = get_cfg()
        cfg.merge_from_file(os.path.join(ROOT, "configs", config))
        cfg.merge_from_list(["MODEL.DEVICE", str(device), "MODEL.SEM_SEG_HEAD.NUM_OBJECT_QUERIES", nq,
                             "MODEL.AMD.COMPUTE_DTYPE", dtype] + list(overrides))
        cfg.freeze()
        model = build_model(cfg).eval()
        model.load_state_dict(synth_state_dict(nq))
        _MODELS[key] = model
    return _MODELS[key]


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()
