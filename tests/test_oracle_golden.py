"""CPU: the oracle restatement vs the golden vectors captured from the imported reference
(oracle/gen_golden.py).  Runs anywhere (no GPU, no /root/reference)."""
import pytest
import torch

from oracle import nopesac_oracle as O
from tests import golden_inputs as GI
from tests.util import gold, loose_oracle_cfg, rel_err

CFG = O.OracleConfig()


def test_state_dict_contract(sd50):
    g = gold("state_dict_keys")
    assert list(g["keys"]) == list(sd50)
    sums = torch.tensor([float(v.double().sum()) for v in sd50.values()], dtype=torch.float64)
    assert rel_err(sums, g["checksum"]) < 1e-9


def test_backbone(sd50):
    from nopesac_amd.synth import synth_pair
    img = synth_pair(11, 64, 96)["0"]["image"]
    with torch.no_grad():
        f = O.backbone(sd50, O.preprocess([img], CFG))
    g = gold("A_backbone_64x96")
    for k, v in f.items():
        assert rel_err(v.flatten()[:: max(v.numel() // 64, 1)][:64], g[k + "_probe"]) < 1e-5
        assert abs(float(v.double().sum()) - float(g[k + "_sum"])) < 1e-4 * abs(float(g[k + "_sum"]))


def test_backbone_full_resolution(sd50):
    """The oracle's backbone at the real input size (480x640) against the imported reference's (fixture A_backbone_480x640)."""
    from nopesac_amd.synth import synth_pair
    img = synth_pair(21, structured=True)["0"]["image"]
    with torch.no_grad():
        f = O.backbone(sd50, O.preprocess([img], CFG))
    g = gold("A_backbone_480x640")
    for k, v in f.items():
        assert rel_err(v.flatten()[:: max(v.numel() // 64, 1)][:64], g[k + "_probe"]) < 1e-5
        assert abs(float(v.double().sum()) - float(g[k + "_sum"])) < 1e-4 * abs(float(g[k + "_sum"]))


def test_plane_head(sd50):
    with torch.no_grad():
        out, q = O.plane_head(sd50, GI.feature_maps(21, 6, 8), CFG)
    g = gold("B_plane_head_6x8")
    assert rel_err(q, g["query_feat"]) < 3e-4
    assert rel_err(out["pred_logits"], g["pred_logits"]) < 3e-4
    assert rel_err(out["pred_params"], g["pred_params"]) < 3e-4
    assert rel_err(out["pred_mask_logits"][0, :, ::4, ::4], g["mask_logits_sub"]) < 3e-4


@pytest.mark.parametrize("kind,seed", [("multi", 31), ("none_pass", 32), ("all_overlap_rejected", 33), ("full", 34)])
def test_postselect(kind, seed):
    s = O.post_select(*GI.postselect_case(kind, seed), CFG)
    g = gold(f"C_postselect_{kind}")
    assert s["pred_plane_oriIdxs"].tolist() == g["idx"].tolist()
    assert torch.equal(s["areas"], g["areas"])
    assert torch.equal(s["pred_plane_masks"].sum(2).to(torch.int32), g["mask_rowsum"])
    assert rel_err(s["pred_plane_ins_center"], g["centers"]) < 1e-5
    assert rel_err(s["pred_plane"], g["planes"]) == 0


def test_posenet(sd50):
    with torch.no_grad():
        t, r, tf, rf, _ = O.pixel_pose_net(sd50, GI.feature_maps(41), GI.feature_maps(42))
    g = gold("D_posenet")
    assert rel_err(t, g["trans"]) < 2e-5 and rel_err(r, g["rot"]) < 2e-5 and rel_err(rf, g["rots_feat"]) < 2e-5


@pytest.mark.parametrize("n1,n2,seed", [(1, 1, 50), (5, 3, 51), (17, 40, 52), (32, 32, 53), (50, 50, 54)])
def test_matcher(sd50, n1, n2, seed):
    with torch.no_grad():
        ls = O.matcher(sd50, *GI.matcher_case(n1, n2, seed), CFG)
    g = gold(f"E_matcher_{n1}x{n2}")
    assert rel_err(ls, g["log_scores"]) < 2e-5
    assert torch.equal(O.assignment_matrix(ls, 0.2), g["assignment"])


@pytest.mark.parametrize("m", [0, 1, 2, 7, 32, 50])
def test_refine(sd50, m):
    c = GI.refine_case(50, m, 60 + m)
    gl, mm = O.geo_sequence(c["planes1"], c["planes2"], c["A"], 50)
    gg, _ = O.geo_sequence(c["planes1"], c["planes2"], c["A"], 50, c["init_rot"], c["init_trans"])
    ga, _ = O.geo_sequence(c["planes1"], c["planes2"], c["A"], 50, c["init_rot"], torch.zeros(3))
    sig = (((gg[:, 0:1] * ga[:, 0:1]) >= 0).float() - 0.5) * 2.0
    with torch.no_grad():
        r = O.ransac_refine(sd50, c["trans_feat"], c["rot_feat"], gg, gl, sig, mm, c["init_trans"], c["init_rot"], CFG)
    g = gold(f"F_refine_nq50_m{m}_soft")
    assert mm == m and rel_err(gg, g["geo_global"]) < 1e-5
    for k, v in r.items():
        assert rel_err(v, g[k]) < 5e-5, k


def _refine_train_inputs(nq, ms, seeds):
    cases = [GI.refine_case(nq, m, s) for m, s in zip(ms, seeds)]
    geo = []
    for c in cases:
        gl, mm = O.geo_sequence(c["planes1"], c["planes2"], c["A"], nq)
        gg, _ = O.geo_sequence(c["planes1"], c["planes2"], c["A"], nq, c["init_rot"], c["init_trans"])
        ga, _ = O.geo_sequence(c["planes1"], c["planes2"], c["A"], nq, c["init_rot"], torch.zeros(3))
        geo.append((gl, gg, (((gg[:, 0:1] * ga[:, 0:1]) >= 0).float() - 0.5) * 2.0, mm))
    st = lambda k: torch.stack([c[k] for c in cases])
    return cases, st, torch.stack([g[0] for g in geo]), torch.stack([g[1] for g in geo]), torch.stack([g[2] for g in geo]), [g[3] for g in geo]


REFINE_TRAIN_CASES = [(50, (7, 2, 32, 50, 1), (67, 62, 92, 110, 61), "initCamRef", 1.0), (50, (32,), (92,), "initRecCamRef", 0.5),
                      (64, (33, 64), (133, 164), "initCamRef_Aux", 2.0)]


@pytest.mark.parametrize("nq,ms,seeds,tag,weight", REFINE_TRAIN_CASES)
def test_refine_training_twin(sd50, nq, ms, seeds, tag, weight):
    """Oracle restatement of __forward_PlaneCamRefHead (camera_head.py:737-923) against the reference's own outputs and losses."""
    from nopesac_amd.synth import synth_state_dict
    sd = sd50 if nq == 50 else synth_state_dict(nq)
    cases, st, gl, gg, sig, mm = _refine_train_inputs(nq, ms, seeds)
    g = gold(f"H_refine_train_nq{nq}_{tag}")
    assert mm == list(ms) and rel_err(gg, g["geo_global"]) < 1e-5 and rel_err(gl, g["geo_local"]) < 1e-6 and torch.equal(sig, g["sig"])
    gt = GI.gt_pose_case(len(ms), seeds[0])
    assert torch.equal(gt, g["gt_pose"])
    with torch.no_grad():
        losses, pr = O.ransac_refine_train(sd, st("trans_feat"), st("rot_feat"), gg, gl, sig, mm, st("init_trans"), st("init_rot"), gt,
                                           O.OracleConfig(num_queries=nq, out_cam_type="soft"), suffix=tag, weight=weight)
    assert len(losses) == 7 and all(k.endswith("_" + tag) for k in losses)
    for k, v in {**losses, **pr}.items():
        assert rel_err(v, g[k]) < 5e-5, k


def test_camera_head_training_forward(sd50):
    """Oracle restatement of PlaneCameraHead.forward in training mode (camera_head.py:140-344: 34 losses + the pose lists) against
    the imported reference's outputs."""
    c = GI.camera_train_case(50, (7, 2, 19), 80)
    torch.set_num_threads(8)
    with torch.no_grad():
        losses, tl, rl = O.camera_head_train(sd50, c["feats1"], c["feats2"], c["gt_planes1"], c["gt_planes2"], c["gt_A"], c["gt_pose"],
                                             c["planes1"], c["planes2"], c["A"], CFG, 1.0, 1.0, 0.1, c["rand_rot"], c["rand_trans"])
    g = gold("I_camhead_train_seed80")
    assert len(losses) == 34 and len(tl) == len(rl) == 10
    for k, v in losses.items():
        assert rel_err(v, g[k]) < 5e-5, k
    for i, (t, r) in enumerate(zip(tl, rl)):
        assert rel_err(t, g[f"trans_list_{i}"]) < 5e-5 and rel_err(r, g[f"rot_list_{i}"]) < 5e-5, i


@pytest.mark.parametrize("tag,structured,idx", [("default_noise", False, 0), ("loose_structured", True, 2)])
def test_e2e(sd50, tag, structured, idx):
    from nopesac_amd.synth import synth_pair
    cfg = loose_oracle_cfg() if structured else CFG
    torch.set_num_threads(8)
    r = O.inference(sd50, [synth_pair(idx, structured=structured)], cfg)[0]
    g = gold(f"e2e_{tag}_{idx}")
    for v in "01":
        assert r[v]["pred_plane_oriIdxs"].tolist() == g[f"v{v}_idx"].tolist()
        assert rel_err(r[v]["pred_plane"], g[f"v{v}_planes"]) < 1e-4
    for k in ("camera", "camera_init", "camera_initRec", "camera_avgRef0"):
        assert rel_err(r[k]["tran"], g[k + "_tran"]) < 1e-4 and rel_err(r[k]["rot"], g[k + "_rot"]) < 1e-4
    assert torch.equal(r["pred_assignment"], g["pred_assignment"])
