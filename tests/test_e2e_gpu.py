"""End-to-end parity on the GPU: PlaneTR_NopeSAC (HIP, fp32 path) vs the CPU oracle and the golden
vectors of the imported reference, on the same synthetic pairs; batching invariance; bf16 sanity."""
import numpy as np
import pytest
import torch

from oracle import rle_oracle as R
from tests.util import LOOSE, abs_err, gold, loose_oracle_cfg, make_model, oracle_f64, quat_abs_err, rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-4   # north_star: outputs within 1e-4 of the reference fp32 path
# camera.tran, camera.rot (up to the quaternion sign) and pred_plane are gated on the ABSOLUTE difference;
# everything else (feature vectors, centres, scores) relative to the output's scale.


def _check_pair(res, ref, g=None, soft_masks=False, ref64=None):
    """soft_masks: under the relaxed thresholds the synthetic checkpoint's masks are decided by ~1e-6
    differences between queries, so mask-derived quantities (pixels, areas, centroids) get a looser,
    explicit tolerance there; ids / plane parameters / cameras / assignments keep the strict gate."""
    c_tol, px_tol = (5e-3, 3000) if soft_masks else (1e-3, 200)
    for v in "01":
        assert res[v]["pred_plane_oriIdxs"] == ref[v]["pred_plane_oriIdxs"].tolist()
        e32 = abs_err(res[v]["pred_plane"], ref[v]["pred_plane"])
        if ref64 is not None and ref64[v]["pred_plane_oriIdxs"].tolist() == ref[v]["pred_plane_oriIdxs"].tolist():
            # canonical value = the float64 evaluation of the reference algorithm (tests/util.py::oracle_f64): our own error
            # must be inside the absolute 1e-4 gate; against the fp32 CPU result the reference's own rounding noise is allowed
            noise = abs_err(ref[v]["pred_plane"].double(), ref64[v]["pred_plane"])
            assert abs_err(res[v]["pred_plane"].double().cpu(), ref64[v]["pred_plane"]) < TOL, (v, e32, noise)
            assert e32 < TOL + noise, (v, e32, noise)
        else:
            assert e32 < TOL, (v, e32)
        assert rel_err(res[v]["pred_plane_feats"], ref[v]["pred_plane_feats"]) < 5e-4
        assert rel_err(res[v]["pred_plane_ins_center"], ref[v]["pred_plane_ins_center"]) < c_tol
        mism = int((res[v]["pred_plane_masks"].cpu() != ref[v]["pred_plane_masks"]).sum())
        assert mism <= px_tol, mism       # 480x640xn booleans; float ties of the bilinear up-sampling only
        # `instances` (siamese_planeTR.py:703-720): COCO RLE + bbox describe exactly the masks returned, scores match
        assert len(res[v]["instances"]) == len(ref[v]["instances"])
        for k, (ins, rins) in enumerate(zip(res[v]["instances"], ref[v]["instances"])):
            own = R.encode(res[v]["pred_plane_masks"][k].cpu().numpy())
            assert ins["segmentation"] == {"size": list(res[v]["pred_plane_masks"].shape[1:]), "counts": own["counts"]}
            assert ins["bbox"] == R.to_bbox(own).tolist() and ins["bbox_mode"] == 1 and ins["category_id"] == 0
            assert abs(ins["score"] - rins["score"]) < TOL
            if mism == 0:
                assert ins["segmentation"] == rins["segmentation"] and ins["bbox"] == rins["bbox"]
        if g is not None:
            assert res[v]["pred_plane_oriIdxs"] == g[f"v{v}_idx"].tolist()
            assert abs_err(res[v]["pred_plane"], g[f"v{v}_planes"]) < TOL      # fixture = the imported reference's fp32 output: north_star's 1e-4
            assert (res[v]["pred_plane_areas"].long() - g[f"v{v}_areas"].long()).abs().max() <= px_tol
    for k in ref:
        if "camera" in k:
            assert k in res, k
            if ref64 is not None and k in ref64:
                # absolute 1e-4 against the float64 evaluation of the reference algorithm; against the fp32 CPU result the
                # reference's own rounding noise is allowed on top (refined poses reach |t| ~ 12 with the synthetic weights:
                # 1e-4 there is 8e-6 relative, ~100 ulp)
                nt, nr = abs_err(ref[k]["tran"].astype("float64"), ref64[k]["tran"]), quat_abs_err(ref[k]["rot"].astype("float64"), ref64[k]["rot"])
                assert abs_err(res[k]["tran"].astype("float64"), ref64[k]["tran"]) < TOL, (k, nt)
                assert quat_abs_err(res[k]["rot"].astype("float64"), ref64[k]["rot"]) < TOL, (k, nr)
                assert abs_err(res[k]["tran"], ref[k]["tran"]) < TOL + nt, (k, nt)
                assert quat_abs_err(res[k]["rot"], ref[k]["rot"]) < TOL + nr, (k, nr)
            else:
                assert abs_err(res[k]["tran"], ref[k]["tran"]) < TOL, k
                assert quat_abs_err(res[k]["rot"], ref[k]["rot"]) < TOL, k
            if g is not None:
                # fixture = the imported reference's fp32 result: north_star's absolute 1e-4, no allowance
                assert abs_err(res[k]["tran"], g[k + "_tran"]) < TOL and quat_abs_err(res[k]["rot"], g[k + "_rot"]) < TOL, k
        if "assignment" in k:
            assert torch.equal(res[k], ref[k]), k
    assert set(k for k in res if "camera" in k) == set(k for k in ref if "camera" in k)


def test_e2e_default_config(device, sd50):
    from nopesac_amd.synth import synth_pair
    from oracle import nopesac_oracle as O
    model = make_model(device)
    inp = [synth_pair(0), synth_pair(3)]
    res = model(inp)
    ref = O.inference(sd50, inp, O.OracleConfig())
    ref64 = oracle_f64(sd50, inp, O.OracleConfig())
    _check_pair(res[0], ref[0], gold("e2e_default_noise_0"), ref64=ref64[0])
    _check_pair(res[1], ref[1], ref64=ref64[1])
    for r in res:
        assert r["pred_aff"] is None and r["depth"] == {"0": None, "1": None}
        assert isinstance(r["camera"]["tran"], np.ndarray) and r["camera"]["rot"].shape == (4,)


def test_e2e_loose_structured_batch(device, sd50):
    """Three structured pairs in ONE batch under relaxed TEST.* thresholds (several planes per view)."""
    from nopesac_amd.synth import synth_pair
    from oracle import nopesac_oracle as O
    model = make_model(device, LOOSE)
    inp = [synth_pair(i, structured=True) for i in (0, 2, 1)]
    res = model(inp)
    ref = O.inference(sd50, inp, loose_oracle_cfg())
    ref64 = oracle_f64(sd50, inp, loose_oracle_cfg())
    _check_pair(res[0], ref[0], gold("e2e_loose_structured_0"), soft_masks=True, ref64=ref64[0])
    _check_pair(res[1], ref[1], gold("e2e_loose_structured_2"), soft_masks=True, ref64=ref64[1])
    _check_pair(res[2], ref[2], soft_masks=True, ref64=ref64[2])
    assert max(len(r["0"]["pred_plane_oriIdxs"]) for r in res) >= 3
    # batching invariance: the same pair alone gives the same answer
    solo = model([inp[1]])[0]
    assert solo["0"]["pred_plane_oriIdxs"] == res[1]["0"]["pred_plane_oriIdxs"]
    assert rel_err(solo["camera"]["tran"], res[1]["camera"]["tran"]) < 1e-5
    assert torch.equal(solo["pred_assignment"], res[1]["pred_assignment"])


@pytest.mark.parametrize("K,nq", [(32, 50), (64, 64), (128, 128)])
def test_bench_workload_forced_k_matches_oracle(device, K, nq):
    """The bench workload (SURVEY.md §8d K control: K planes per view, K matches) on the fp32 HIP path against the oracle
    driven through the same K control - the configuration bench.py times is itself parity-checked, K = 32 (config 2)
    K = 64 / nq = 64 (config 3) and K = 128 / nq = 128 (the K of config 5)."""
    import bench
    from nopesac_amd.synth import synth_pair, synth_state_dict
    from oracle import nopesac_oracle as O
    B = 2
    model = make_model(device, nq=nq)
    inp = [synth_pair(20 + i) for i in range(B)]
    forced = bench.make_forced(B, K, nq, device, 5)
    with torch.no_grad():
        d = model.forward_tensors(model.preprocess_image(inp), B, 480, 640, forced=forced)
    cam = d["cam"]
    cpu_forced = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in forced.items()}
    ref = O.inference(synth_state_dict(nq), inp, O.OracleConfig(num_queries=nq), forced=cpu_forced)
    assert cam["m"].tolist() == [K] * B and [r["_aux"]["matched_num"] for r in ref] == [K] * B
    for key in ("camera_init", "camera_initRec", "camera_avgRef0", "camera"):
        t, r = cam["cameras"][key]
        for b in range(B):
            assert abs_err(t[b].cpu().numpy(), ref[b][key]["tran"]) < TOL, (key, b)          # absolute 1e-4 (|tran| is ~10 here)
            assert quat_abs_err(r[b].cpu().numpy(), ref[b][key]["rot"]) < TOL, (key, b)


def test_bench_configuration_bf16_forced_k32_b32(device):
    """EXACTLY what the driver times (bench.py defaults: bf16, 32 pairs per step, K = 32 forced, raw-image fused stem) against the
    fp32 HIP path (the 1e-4-parity path, test above) under the SAME forced control: every pair keeps m = 32 hypotheses, the
    quaternions are unit, everything is finite, and the pose error stays inside the bounds bench.py prints as
    `pose_err_vs_fp32_path.bench_workload` (measured on MI355X: camera T 0.04 of |t| = 11, R 0.25 deg mean / 0.45 max;
    camera_init T 0.003 / R 0.8 deg mean, 2.9 max)."""
    import bench
    B, K, nq = 32, 32, 50
    m16, m32 = bench.build_model(device, nq, "bfloat16"), bench.build_model(device, nq, "float32")
    err = bench.bench_workload_pose_error(m16, m32, device, B, K, nq)
    assert err["m_bf16"] == [K] * B and err["m_fp32"] == [K] * B
    assert err["finite"] and err["max_quat_norm_dev"] < 1e-3
    cam, ini, rec = err["camera"], err["camera_init"], err["camera_initRec"]
    # FIXED bounds (round 5; rounds 2-4 used 1.5 x whatever had been measured): refined camera R <= 1 deg, pixel pose R <= 2.5 deg,
    # its re-embedding R <= 4.5 deg.  Measured on MI355X with the AIM on f32 operands (MODEL.AMD.POSE_FP32_PARTS default "aim"):
    # camera 0.43 / camera_init 2.03 / camera_initRec 3.61 deg max (profiles/r5_c_aim_fp32_ab.txt).  The re-embedding cannot be held to
    # 3 deg by its own precision: with f32 operands it still maps the pixel pose's 0.80 deg mean / 2.03 max error to 1.41 / 3.61 - the
    # AIM of the synthetic checkpoint amplifies an input perturbation 1.8x whatever arithmetic evaluates it.
    assert cam["R_err_deg_max"] < 1.0 and cam["T_err_max"] < 0.0117 * cam["mean_abs_t"], cam
    assert ini["R_err_deg_max"] < 2.5 and ini["T_err_max"] < 0.0105, ini
    assert rec["R_err_deg_max"] < 4.5, rec


def test_e2e_scannet_config_nq64(device):
    """BASELINE configs[2]: configs/inference_scannet.yaml through the model with NUM_OBJECT_QUERIES = 64 (K = 64 needs nq = 64,
    SURVEY.md fact 4): end-to-end parity with the oracle, and the K = 64 forced workload of `bench.py --config scannet --k 64`."""
    import bench
    from nopesac_amd.synth import synth_pair, synth_state_dict
    from oracle import nopesac_oracle as O
    nq, K, B = 64, 64, 2
    model = make_model(device, nq=nq, config="inference_scannet.yaml")
    assert model.cfg.DATASETS.TEST == ("scannet_test",) and model.num_queries == nq
    inp = [synth_pair(11), synth_pair(12)]
    res = model(inp)
    ref = O.inference(synth_state_dict(nq), inp, O.OracleConfig(num_queries=nq))
    ref64 = oracle_f64(synth_state_dict(nq), inp, O.OracleConfig(num_queries=nq))
    for a, b, c in zip(res, ref, ref64):
        _check_pair(a, b, ref64=c)
    forced = bench.make_forced(B, K, nq, device, 3)
    with torch.no_grad():
        cam = model.forward_tensors(model.preprocess_image(inp), B, 480, 640, forced=forced)["cam"]
    cpu_forced = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in forced.items()}
    ref = O.inference(synth_state_dict(nq), inp, O.OracleConfig(num_queries=nq), forced=cpu_forced)
    assert cam["m"].tolist() == [K] * B
    for key in ("camera_init", "camera"):
        t, r = cam["cameras"][key]
        for b in range(B):
            assert abs_err(t[b].cpu().numpy(), ref[b][key]["tran"]) < TOL and quat_abs_err(r[b].cpu().numpy(), ref[b][key]["rot"]) < TOL, (key, b)


def test_bf16_backbone_pose_error(device):
    """bf16 dense convs: judged by pose error against the fp32 HIP path (discrete decisions may flip,
    SURVEY.md §7 'hard parts'), not by the 1e-4 gate."""
    from nopesac_amd.synth import synth_pair
    m32, m16 = make_model(device), make_model(device, dtype="bfloat16")
    inp = [synth_pair(i) for i in range(2)]
    a, b = m32(inp), m16(inp)
    for x, y in zip(a, b):
        t_err = float(np.linalg.norm(x["camera_init"]["tran"] - y["camera_init"]["tran"]))
        q = abs(float(np.dot(x["camera_init"]["rot"], y["camera_init"]["rot"])))
        r_err = 2 * np.degrees(np.arccos(min(q, 1.0)))
        assert t_err < 0.05 * (1 + np.linalg.norm(x["camera_init"]["tran"])) and r_err < 10.0, (t_err, r_err)
        assert np.isfinite(y["camera"]["tran"]).all() and np.isfinite(y["camera"]["rot"]).all()


def test_bf16_refined_pose_on_natural_matches(device):
    """The REFINED camera in bf16 on NON-forced inputs with >= 2 matches per pair (round-5 verdict, parity hole (a); round 6).
    Twin pairs - both views show the same structured image, relaxed thresholds - are the synthetic checkpoint's only source of natural
    matches (a random-weight matcher finds none between unrelated images).  A pair takes part when both precisions keep the SAME
    planes and find the SAME >= 2 matches: then the two paths differ by arithmetic only.  Fixed bounds: refined rotation <= 2.5 deg,
    refined translation <= 5 % of |t| (measured on MI355X: 0.79 / 1.61 deg mean / max, 2.3 / 3.2 %, `scripts/twin_pairs_gate.py`,
    profiles/r6_d_bf16_refined_pose_non_forced.txt; the same pairs move by 0.22 / 0.50 deg and 1.0 / 2.3 % in PURE f32 under +-0.5 grey
    levels of input noise).  What the round-5 record's `loose_structured.camera` (T 0.74, R 4.4 deg) showed is something else: pairs
    with m <= 1 and pairs whose kept-plane sets differ between the precisions, where the random-weight refinement head moves by 5 % / up
    to 8 deg under that f32 input noise alone (same file) - conditioning of the synthetic checkpoint, not a bf16 defect."""
    from nopesac_amd import runner
    from nopesac_amd.synth import synth_pair
    m32, m16 = make_model(device, LOOSE), make_model(device, LOOSE, dtype="bfloat16")
    cand = [2, 8, 9, 11, 21, 32, 33, 42, 43, 50, 52, 55, 59, 63, 64, 65, 70, 72, 77, 80, 81, 84, 86, 89, 90, 95]       # fp32: m >= 2 as twins
    took = []
    for lo in range(0, len(cand), 13):
        inp = []
        for i in cand[lo:lo + 13]:
            d = synth_pair(i, structured=True)
            d["1"] = dict(d["0"], image=d["0"]["image"].clone(), image_id=d["1"]["image_id"], file_name=d["1"]["file_name"])
            inp.append(d)
        for i, x, y in zip(cand[lo:lo + 13], m32(inp), m16(inp)):
            if int(x["matched_num"]) < 2 or int(x["matched_num"]) != int(y["matched_num"]):
                continue
            if any(x[v]["pred_plane_oriIdxs"] != y[v]["pred_plane_oriIdxs"] for v in "01"):
                continue
            if not np.array_equal(np.asarray(x["pred_assignment_beforeRef0"]) > 0, np.asarray(y["pred_assignment_beforeRef0"]) > 0):
                continue
            t = float(runner.translation_error(y["camera"]["tran"][None], x["camera"]["tran"][None])[0]) / float(np.linalg.norm(x["camera"]["tran"]))
            r = float(runner.rotation_error_deg(y["camera"]["rot"][None], x["camera"]["rot"][None])[0])
            took.append((i, int(x["matched_num"]), round(t, 4), round(r, 3)))
    assert len(took) >= 3, took                      # (MI355X: six pairs qualify; a flipped plane decision on another box may cost one or two)
    assert max(t for _, _, t, _ in took) < 0.05 and max(r for _, _, _, r in took) < 2.5, took


@pytest.mark.parametrize("replay", ["launches", "graph"])
def test_hip_graph_mode_reproduces_the_eager_results(device, replay):
    """MODEL.AMD.USE_HIP_GRAPH: the forward of a batch as one hipGraph replay (warm-up call eager, capture on a slot's second
    call, replays afterwards; two slots round-robin).  Every call must return exactly what the eager bf16 model returns for the
    same inputs (same kernels, same order), for changing inputs, and results handed out earlier must not change when the
    slot's static buffers are overwritten by later replays."""
    import copy
    from nopesac_amd.synth import synth_pair
    eager = make_model(device, dtype="bfloat16")
    graph = make_model(device, ["MODEL.AMD.USE_HIP_GRAPH", True, "MODEL.AMD.GRAPH_REPLAY", replay], dtype="bfloat16")
    assert graph.use_hip_graph and not eager.use_hip_graph and graph.graph_replay == replay
    kept = []
    for it in range(7):                                   # slots 1, 0, 1, 0, ...: eager, eager, capture, capture, replay x 3
        inp = [synth_pair(3 * it + i) for i in range(2)]
        a, b = eager(inp), graph(inp)
        for x, y in zip(a, b):
            for k in ("camera", "camera_init", "camera_initRec", "camera_softRef0"):
                assert np.array_equal(x[k]["tran"], y[k]["tran"]) and np.array_equal(x[k]["rot"], y[k]["rot"]), (it, k)
            assert x["matched_num"] == y["matched_num"] and torch.equal(x["pred_assignment"], y["pred_assignment"])
            for v in "01":
                assert torch.equal(x[v]["pred_plane"], y[v]["pred_plane"]) and x[v]["pred_plane_oriIdxs"] == y[v]["pred_plane_oriIdxs"]
                assert torch.equal(x[v]["pred_plane_feats"], y[v]["pred_plane_feats"]) and torch.equal(x[v]["winner_map"], y[v]["winner_map"])
                assert [i["segmentation"] for i in x[v]["instances"]] == [i["segmentation"] for i in y[v]["instances"]]
        kept.append((b, [{v: (r[v]["pred_plane_feats"].clone(), r[v]["winner_map"].clone(), r[v]["pred_plane"].clone(),
                              r["pred_assignment"].clone()) for v in "01"} for r in b]))
    assert len(graph._graphs) == 2 and all(st["graph"] is not None for st in graph._graphs.values())
    if replay == "launches":
        # the launch tape must be what replayed (csrc/tape.hip): every slot has one, it holds the forward's ~270 kernel launches
        assert getattr(graph, "tape_error", None) is None, graph.tape_error
        assert all(st["tape"] is not None for st in graph._graphs.values())
        assert graph.tape_counts["kernels"] > 150 and graph.tape_counts["streams"] >= 2, graph.tape_counts   # (the pose net's side stream is kept)
    else:
        assert all(st["tape"] is None for st in graph._graphs.values())
    for b, snap in kept:                                  # earlier results are untouched by the later replays
        for r, s0 in zip(b, snap):
            for v in "01":
                assert torch.equal(r[v]["pred_plane_feats"], s0[v][0]) and torch.equal(r[v]["winner_map"], s0[v][1])
                # (host views of the fetch buffer: the fetch is part of the graph and writes the same pinned buffer on every replay)
                assert torch.equal(r[v]["pred_plane"], s0[v][2]) and torch.equal(r["pred_assignment"], s0[v][3])
    # a checkpoint load drops the captured graphs (they hold the addresses of the old packed weights): the new weights take effect
    sd = {k: v.clone() for k, v in eager.state_dict().items()}
    sd["camera_head_list.0.trans.weight"] = sd["camera_head_list.0.trans.weight"] * 1.5
    old = {k: v.clone() for k, v in eager.state_dict().items()}
    try:
        eager.load_state_dict(sd)
        graph.load_state_dict(sd)
        assert graph._graphs == {}
        inp = [synth_pair(90 + i) for i in range(2)]
        for it in range(3):
            a, b = eager(inp), graph(inp)
            for x, y in zip(a, b):
                assert np.array_equal(x["camera_init"]["tran"], y["camera_init"]["tran"]) and np.array_equal(x["camera"]["tran"], y["camera"]["tran"])
        assert not np.array_equal(kept[-1][0][0]["camera_init"]["tran"], b[0]["camera_init"]["tran"])
    finally:                                              # the models are shared with other tests (tests/util.make_model cache)
        eager.load_state_dict(old)
        graph.load_state_dict(old)


def test_uint8_images_give_the_float32_results(device):
    """uint8 CHW image tensors at the boundary (data.PairMapper(uint8=True)) are widened on the device: the results are identical
    to those for the reference mapper's float32 tensors, in the fp32 path (preprocess kernel) and the bf16 path (raw-input stem)."""
    from nopesac_amd.synth import synth_pair
    for dtype in ("float32", "bfloat16"):
        model = make_model(device, dtype=dtype)
        inp = [synth_pair(40 + i) for i in range(2)]
        inp8 = [{v: dict(p[v], image=p[v]["image"].round().clamp(0, 255).to(torch.uint8)) if v in "01" else p[v] for v in p} for p in inp]
        inpf = [{v: dict(p[v], image=p[v]["image"].float()) if v in "01" else p[v] for v in p} for p in inp8]
        a, b = model(inpf), model(inp8)
        for x, y in zip(a, b):
            for k in ("camera", "camera_init"):
                assert np.array_equal(x[k]["tran"], y[k]["tran"]) and np.array_equal(x[k]["rot"], y[k]["rot"]), (dtype, k)
            for v in "01":
                assert torch.equal(x[v]["pred_plane"], y[v]["pred_plane"]) and torch.equal(x[v]["winner_map"], y[v]["winner_map"])


def test_cli_runner_end_to_end(device, tmp_path):
    """`python -m nopesac_amd.run` flow (cfg -> build_model -> checkpoint -> batch loop -> evaluator) on the GPU."""
    import json
    from nopesac_amd import run
    from tests.util import ROOT
    import os
    out = tmp_path / "res.json"
    res = run.main(["--config-file", os.path.join(ROOT, "configs", "inference_mp3d.yaml"), "--eval-only", "--synthetic-weights",
                    "--synthetic-pairs", "3", "--pairs-per-batch", "2", "--output", str(out), "MODEL.DEVICE", str(device)])
    assert res["pairs"]["count"] == 3 and res["timing(rank0)"]["pairs"] == 3
    assert json.load(open(out))["pairs"]["count"] == 3


def test_cli_runner_rows_match_the_oracle(device, tmp_path, sd50):
    """The reference's harness flow (test_NopeSAC.py:157-179: inference_on_dataset -> evaluator.process -> evaluate; rows and error
    formulas mp3d_evaluation.py:184-257, 389-465) through `python -m nopesac_amd.run`: 3 synthetic pairs with a ground-truth
    rel_pose, 2 pairs per batch (ragged last batch), 2 batches in flight, fp32.  What the runner hands to its evaluator and writes
    to NopeSAC_instances_predictions.pth must be the oracle's per-pair results within the absolute 1e-4 gate (pair 0 also against
    the imported reference's fixture), and the pose-error table must be the one the oracle's poses give."""
    import json
    import os
    from nopesac_amd import run, runner
    from nopesac_amd.evaluation import camera_metrics
    from nopesac_amd.synth import synth_pair
    from oracle import nopesac_oracle as O
    from tests.util import ROOT
    rng = np.random.default_rng(5)
    pairs = []
    for i in (0, 3, 5):
        p = synth_pair(i)
        q = rng.normal(size=4)
        q = (q / np.linalg.norm(q) * (1 if q[0] >= 0 else -1)).astype(np.float32)
        p["rel_pose"] = {"position": rng.normal(size=3).astype(np.float32).tolist(), "rotation": q.tolist()}
        pairs.append(p)
    torch.save(pairs, tmp_path / "pairs.pt")
    res = run.main(["--config-file", os.path.join(ROOT, "configs", "inference_mp3d.yaml"), "--eval-only", "--synthetic-weights",
                    "--pairs-file", str(tmp_path / "pairs.pt"), "--pairs-per-batch", "2", "--inflight", "2", "--dump-dir", str(tmp_path / "dump"),
                    "--output", str(tmp_path / "res.json"), "MODEL.DEVICE", str(device)])
    assert res["pairs"]["count"] == 3 and res["timing(rank0)"]["batches_in_flight"] == 2
    ref = O.inference(sd50, pairs, O.OracleConfig())
    preds = torch.load(tmp_path / "dump" / "NopeSAC_instances_predictions.pth", weights_only=False)
    assert [p["0"]["image_id"] for p in preds] == [p["0"]["image_id"] for p in pairs]          # dataset order kept across the pipelined batches
    g = gold("e2e_default_noise_0")
    for i, (pr, rf, inp) in enumerate(zip(preds, ref, pairs)):
        for v in "01":
            assert abs_err(pr[v]["pred_plane"], rf[v]["pred_plane"]) < TOL, (i, v)
            assert len(pr[v]["instances"]) == len(rf[v]["instances"])
        for k in ("camera_init", "camera_initRec", "camera"):
            assert abs_err(pr[k]["pred"]["tran"], rf[k]["tran"]) < TOL and quat_abs_err(pr[k]["pred"]["rot"], rf[k]["rot"]) < TOL, (i, k)
            assert pr[k]["gts"]["tran"] == inp["rel_pose"]["position"] and pr[k]["gts"]["rot"] == inp["rel_pose"]["rotation"]
        assert torch.equal(pr["pred_assignment"], rf["pred_assignment"])
    for v in "01":
        assert abs_err(preds[0][v]["pred_plane"], g[f"v{v}_planes"]) < TOL
    for k in ("camera_init", "camera"):
        assert abs_err(preds[0][k]["pred"]["tran"], g[k + "_tran"]) < TOL and quat_abs_err(preds[0][k]["pred"]["rot"], g[k + "_rot"]) < TOL, k
    # the error table: identical formulas on the oracle's poses
    gt_t = np.array([p["rel_pose"]["position"] for p in pairs], np.float32)
    gt_q = np.array([p["rel_pose"]["rotation"] for p in pairs], np.float32)
    for k in ("camera_init", "camera"):
        want = camera_metrics(np.stack([r[k]["tran"] for r in ref]).astype(np.float32), np.stack([r[k]["rot"] for r in ref]).astype(np.float32), gt_t, gt_q)
        got = json.load(open(tmp_path / "res.json"))[k]
        assert set(got) == set(want)
        for name in want:
            assert abs(got[name] - want[name]) <= 2e-3 * (1 + abs(want[name])), (k, name, got[name], want[name])     # degrees / metres of the error itself
    te = runner.translation_error(np.stack([p["camera"]["pred"]["tran"] for p in preds]), gt_t)
    assert np.isfinite(te).all()


def test_cli_runner_on_a_dataset_split_from_disk(device, tmp_path, monkeypatch):
    """The runner on a (tiny) Matterport3D-style split on disk: json -> LazyPairs (decoder threads running ahead) -> uint8 images ->
    batches in flight -> evaluator; 5 pairs, 2 per batch (ragged last batch), pose rows for every pair with a rel_pose.  The PNG frames
    go through the library's batch decoder into one pinned buffer per batch that crosses PCIe in one transfer (round 5); the same run with
    NOPESAC_PNG_NATIVE=0 (PIL per image, one tensor and one transfer per image - the reference's decode) must give the same rows."""
    import json
    import os
    from PIL import Image
    from nopesac_amd import run
    from tests.util import ROOT
    rng = np.random.default_rng(11)
    root = tmp_path / "datasets" / "mp3d_dataset"
    (root / "mp3d_planercnn_json").mkdir(parents=True)
    entries = []
    for k in range(5):
        pair = {"rel_pose": {"position": [0.1 * k, 0.2, 0.3], "rotation": [1.0, 0.0, 0.0, 0.0]}}
        for v in "01":
            f = root / f"img_{k}_{v}.png"
            Image.fromarray(rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)).save(f)
            pair[v] = {"file_name": str(f), "image_id": f"house_{k}_{v}", "height": 480, "width": 640}
        entries.append(pair)
    json.dump({"categories": [], "data": entries}, open(root / "mp3d_planercnn_json" / "cached_set_test.json", "w"))
    argv = ["--config-file", os.path.join(ROOT, "configs", "inference_mp3d.yaml"), "--eval-only", "--synthetic-weights",
            "--dataset", "mp3d_test", "--datasets-dir", str(tmp_path / "datasets"), "--pairs-per-batch", "2", "--uint8-images", "--inflight", "2"]
    opts = ["MODEL.DEVICE", str(device), "MODEL.AMD.AUTOTUNE", False]
    res = run.main(argv + ["--output", str(tmp_path / "native.json")] + opts)
    assert res["pairs"]["count"] == 5 and res["timing(rank0)"]["pairs"] == 5 and res["timing(rank0)"]["batches_in_flight"] == 2
    monkeypatch.setenv("NOPESAC_PNG_NATIVE", "0")
    ref = run.main(argv + ["--output", str(tmp_path / "pil.json")] + opts)
    assert ref["pairs"]["count"] == 5
    ja, jb = json.load(open(tmp_path / "native.json")), json.load(open(tmp_path / "pil.json"))
    seen = []

    def same(x, y, path):
        if isinstance(x, dict):
            assert sorted(x) == sorted(y), path
            for k in x:
                same(x[k], y[k], path + "/" + k)
        elif isinstance(x, list):
            assert len(x) == len(y), path
            for i, (p, q) in enumerate(zip(x, y)):
                same(p, q, path + "/%d" % i)
        elif isinstance(x, float):
            assert x == y, (path, x, y)
            seen.append(path)
    same(ja["pairs"], jb["pairs"], "pairs")
    assert len(seen) >= 2, seen


def test_cli_runner_on_a_scannet_style_jpeg_split(device, tmp_path, monkeypatch):
    """The runner on a ScanNet-style split on disk: 968 x 1296 JPEG frames -> reader threads (file + marker walk) -> GPU JPEG decode of
    whole batches ahead of the consumer -> GPU resize to 480 x 640 -> batches in flight -> evaluator.  The decode is bit-exact with the
    PIL decode the reference performs, so the run with NOPESAC_GPU_JPEG=0 (host decode) must give the same pose rows."""
    import json
    import os
    from PIL import Image
    from nopesac_amd import run
    from tests.util import ROOT
    rng = np.random.default_rng(12)
    root = tmp_path / "datasets" / "scannet_dataset"
    (root / "scannet_json").mkdir(parents=True)
    yy, xx = np.mgrid[0:968, 0:1296].astype(np.float32)
    entries = []
    for k in range(3):
        pair = {"rel_pose": {"position": [0.1 * k, 0.2, 0.3], "rotation": [1.0, 0.0, 0.0, 0.0]}}
        for v in "01":
            f = root / f"frame_{k}_{v}.jpg"
            a = np.stack([128 + 90 * np.sin(xx / (30 + 7 * k) + yy / 80), 128 + 70 * np.cos(yy / (25 + 3 * int(v))) * np.sin(xx / 120),
                          120 + 100 * ((xx // (100 + 20 * k) + yy // 90) % 2)], -1)
            Image.fromarray(np.clip(a + rng.normal(0, 4, a.shape), 0, 255).astype(np.uint8)).save(f, format="JPEG", quality=88, subsampling=2)
            pair[v] = {"file_name": str(f), "image_id": f"scene_{k}_{v}", "height": 480, "width": 640}
        entries.append(pair)
    json.dump({"categories": [], "data": entries}, open(root / "scannet_json" / "cached_set_testV2.json", "w"))
    argv = ["--config-file", os.path.join(ROOT, "configs", "inference_scannet.yaml"), "--eval-only", "--synthetic-weights",
            "--dataset", "scannet_test", "--datasets-dir", str(tmp_path / "datasets"), "--pairs-per-batch", "2", "--uint8-images",
            "--inflight", "2"]
    opts = ["MODEL.DEVICE", str(device), "MODEL.AMD.AUTOTUNE", False]
    a = run.main(argv + ["--output", str(tmp_path / "gpu.json")] + opts)
    monkeypatch.setenv("NOPESAC_GPU_JPEG", "0")
    b = run.main(argv + ["--output", str(tmp_path / "host.json")] + opts)
    assert a["pairs"]["count"] == b["pairs"]["count"] == 3
    ja, jb = json.load(open(tmp_path / "gpu.json")), json.load(open(tmp_path / "host.json"))
    seen = []

    def same(x, y, path):
        if isinstance(x, dict):
            assert sorted(x) == sorted(y), path
            for k in x:
                same(x[k], y[k], path + "/" + k)
        elif isinstance(x, list):
            assert len(x) == len(y), path
            for i, (p, q) in enumerate(zip(x, y)):
                same(p, q, path + "/%d" % i)
        elif isinstance(x, float):
            assert abs(x - y) <= 1e-6 * (1 + abs(x)), (path, x, y)
            seen.append(path)
    same(ja["pairs"], jb["pairs"], "pairs")
    assert len(seen) >= 2, seen


def test_cli_runner_autotunes_and_keeps_a_routing_file(device, tmp_path):
    """bfloat16 mode of the runner: MODEL.AMD.AUTOTUNE times the kernel candidates of every conv / GEMM shape of a pairs-per-batch
    forward before the first batch, MODEL.AMD.ROUTING_FILE persists the decisions; a second run loads them and measures nothing;
    the poses of the two runs agree (the routed kernels compute the same convolutions)."""
    import json
    import os
    from nopesac_amd import ops, run
    from tests.util import ROOT
    routing = tmp_path / "routing.json"
    saved = (ops.TUNER.best, ops.TUNER.loaded, ops.TUNER.log)
    ops.TUNER.best, ops.TUNER.loaded, ops.TUNER.log = {}, {}, []
    try:
        head = ["--config-file", os.path.join(ROOT, "configs", "inference_mp3d.yaml"), "--eval-only", "--synthetic-weights", "--synthetic-pairs", "4",
                "--pairs-per-batch", "2"]
        opts = ["MODEL.DEVICE", str(device), "MODEL.AMD.COMPUTE_DTYPE", "bfloat16", "MODEL.AMD.ROUTING_FILE", str(routing)]
        a = run.main(head + ["--output", str(tmp_path / "a.json")] + opts)
        doc = json.load(open(routing))
        assert doc["format"] == "nopesac_amd.ConvTuner/1" and len(doc["routing"]) > 30     # (40 distinct shapes since the pose-net branches are one launch)
        n_measured = len(ops.TUNER.log)
        assert n_measured > 30
        ops.TUNER.best, ops.TUNER.log = {}, []                     # a fresh process would start like this
        b = run.main(head + ["--output", str(tmp_path / "b.json")] + opts)
        assert len(ops.TUNER.log) == 0 and len(ops.TUNER.loaded) == len(doc["routing"])
        assert a["pairs"]["count"] == b["pairs"]["count"] == 4
        ja, jb = json.load(open(tmp_path / "a.json")), json.load(open(tmp_path / "b.json"))
        for k in ja["pairs"]:
            if isinstance(ja["pairs"][k], float):
                assert abs(ja["pairs"][k] - jb["pairs"][k]) <= 1e-6 * (1 + abs(ja["pairs"][k])), k
    finally:
        ops.TUNER.best, ops.TUNER.loaded, ops.TUNER.log = saved


@pytest.mark.parametrize("nq", [64, 128])
def test_e2e_more_queries(device, nq):
    """BASELINE configs 3/5 need NUM_OBJECT_QUERIES = 64 / 128: end-to-end parity at those sizes."""
    from nopesac_amd.synth import synth_pair, synth_state_dict
    from oracle import nopesac_oracle as O
    model = make_model(device, nq=nq)
    inp = [synth_pair(7)]
    res = model(inp)
    ref = O.inference(synth_state_dict(nq), inp, O.OracleConfig(num_queries=nq))
    _check_pair(res[0], ref[0], ref64=oracle_f64(synth_state_dict(nq), inp, O.OracleConfig(num_queries=nq))[0])


def test_graft_entry_build_then_smoke_in_one_process():
    """build() loads the C-ABI library before anything touched torch.cuda; smoke() must still find the device (the library
    has to share PyTorch's bundled HIP runtime, see nopesac_amd/_lib.py::load)."""
    import subprocess
    import sys
    from tests.util import ROOT
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build(); g.smoke(); print('SMOKE-OK')"], cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "SMOKE-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_config5_fp8_backbone_k128(device):
    """BASELINE config 5 on one GPU: fp8 3x3 backbone convs (MODEL.AMD.BACKBONE_FP8) + K = 128 forced hypotheses, nq = 128, on the
    bench workload (8 pairs) against the fp32 HIP path under the same K control (scripts/fp8_error.py, same seeds).  Measured in
    round 3 (fp8 / plain bf16): camera_init R max 4.32 / 1.37 deg, T max 0.0246 / 0.0067 (|t| = 0.36); camera_initRec R max 12.9 / 3.2;
    refined camera R max 0.34 / 0.34 deg, T max 0.21 / 0.06 (|t| = 11.2).  Gates = 1.5x the measured fp8 maxima (round 4; were 2x)."""
    import bench
    from nopesac_amd import ops
    B, K, nq = 8, 128, 128
    m8 = bench.build_model(device, nq, "bfloat16", ["MODEL.AMD.BACKBONE_FP8", True])
    m32 = bench.build_model(device, nq, "float32")
    g = torch.Generator().manual_seed(1000)
    raw = torch.randint(0, 256, (2 * B, 3, 480, 640), generator=g).float().to(device)
    forced = bench.make_forced(B, K, nq, device, 7)
    with torch.no_grad():
        m8.backbone.calibrate_fp8(ops.preprocess(raw[:4], m8.pixel_mean, m8.pixel_std, m8.backbone.STEM_CIN_PAD, m8.compute_dtype))
    e = bench.bench_workload_pose_error(m8, m32, device, B, K, nq, raw=raw, forced=forced)
    assert e["m_bf16"] == [K] * B == e["m_fp32"] and e["finite"] and e["max_quat_norm_dev"] < 1e-3
    assert e["camera_init"]["R_err_deg_max"] < 6.5 and e["camera_init"]["T_err_max"] < 0.037, e["camera_init"]
    assert e["camera_initRec"]["R_err_deg_max"] < 19.4, e["camera_initRec"]
    assert e["camera"]["R_err_deg_max"] < 0.51 and e["camera"]["T_err_max"] < 0.32, e["camera"]


def test_other_resolutions_are_rejected_like_the_reference(device):
    """The pixel pose net's correlation stack has 15*20 = 300 input channels (camera_modules.py: convs_trans.0 / convs_rots.0), so the
    reference architecture only runs on 480x640 inputs; any other size must fail with a clear message, not a kernel error."""
    from nopesac_amd.synth import synth_pair
    model = make_model(device)
    with pytest.raises(AssertionError, match="480x640"):
        model([synth_pair(60, 256, 384)])


def test_nonfinite_outputs_raise(device):
    """the reference traps NaN poses with pdb.set_trace(); here Inf / NaN in the returned poses / plane parameters are
    counted on the device (nopesac_count_nonfinite) and `model(...)` raises FloatingPointError when the results are fetched."""
    from nopesac_amd import ops
    from nopesac_amd.synth import synth_pair
    t = torch.tensor([1.0, float("nan"), float("inf"), -float("inf"), 0.0, -3.0], device=device)
    assert int(ops.count_nonfinite([t, t[:1]])[0]) == 3
    model = make_model(device)
    inp = [synth_pair(5)]
    model(inp)                                                     # finite: no error
    head = model.camera_head_list[0]
    w = head.raw("trans.weight")                                  # the last linear of the pixel pose net: no ReLU behind it swallows the NaN
    saved = w.detach().clone()
    try:
        with torch.no_grad():
            w.fill_(float("nan"))
        head.invalidate()
        with pytest.raises(FloatingPointError, match="non-finite"):
            model(inp)
    finally:
        with torch.no_grad():
            w.copy_(saved)
        head.invalidate()
    model(inp)


@pytest.mark.parametrize("launcher", ["torchrun", "self"])
def test_bench_two_gpus_rccl(launcher):
    """bench.py --gpus 2 over RCCL, under torchrun (the driver's scaling run is the 8-GPU version of exactly this command) and with
    bench.py starting its own two ranks (round 4): skipped unless the box has two GPUs.  Rank 0's line must report n_gpus = 2,
    two RCCL ranks and a global batch of 2 x pairs."""
    import json
    import os
    import subprocess
    import sys
    from tests.util import ROOT
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    bench_cmd = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--pairs", "8", "--no-cpu-baseline", "--no-accuracy",
                 "--no-fp32-path"]
    cmd = [sys.executable] + (["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
                               "29533"] if launcher == "torchrun" else []) + bench_cmd
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[-1])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 16 and out["value"] > 0 and out["config"]["nonfinite_outputs"] == 0
    assert out["config"]["rccl_ranks"] == 2 and abs(out["config"]["pairs_per_s_per_gpu"] * 2 - out["value"]) < 1e-2 * out["value"]


def test_steps_in_flight_are_bit_identical(device):
    """Round-3 regression (profiles/r3_packed_fp32_hazard.txt): with four batches in flight on their own HIP streams (+ the pose net's
    side streams) every step of the benchmark loop must return the SAME rows when every slot holds the same images.  Before the library
    was built without packed-f32 VALU instructions, 40-50 % of such steps deviated (1e-4 .. 3e-2 on the refined pose of a few pairs):
    ransac_score_maps_kernel's v_pk_*_f32 results were corrupted in lanes 48-63 whenever another batch's MFMA kernels shared its SIMD."""
    import bench
    from nopesac_amd import runner
    B, K, nq, slots, steps = 16, 32, 50, 4, 40
    model = bench.build_model(device, nq, "bfloat16")
    raw = torch.randint(0, 256, (2 * B, 3, 480, 640), generator=torch.Generator().manual_seed(1000)).float().to(device)
    raws = [raw] + [raw.clone() for _ in range(slots - 1)]
    forced = bench.make_forced(B, K, nq, device, 7)
    loop = runner.InflightLoop(slots, B, device, 1)

    def device_step(slot):
        cam = model.forward_tensors(None, B, 480, 640, forced=forced, raw_images=raws[slot])["cam"]
        rows = runner.metric_rows(cam["cameras"]["camera"][0], cam["cameras"]["camera"][1], cam["n1"], cam["n2"], cam["m"], 0)
        rows[:, 10:13] = cam["cameras"]["camera_init"][0]
        rows[:, 13:16] = cam["refine"]["maps"]["normal_score"].sum((1, 2)).view(-1, 1)
        return None, rows

    blocks, hist = [], []
    with torch.no_grad():
        for i in range(steps):
            slot = i % slots
            if loop.done[slot] is not None and len(hist) >= slots:
                loop.done[slot].synchronize()
                blocks.append(loop.host_bufs[slot].clone())
            loop.step(i, device_step)
            hist.append(slot)
        loop.barrier()
    blocks += [loop.host_bufs[s].clone() for s in hist[-slots:]]
    assert len(blocks) == steps
    off = [i for i, b in enumerate(blocks) if not torch.equal(b, blocks[0])]
    assert not off, "steps whose rows differ from step 0: %s" % off


def test_bench_line_carries_the_launch_tape_leg(device):
    """`python bench.py` (the driver's command, shortened): the JSON line must carry the contract's keys, `roofline`, and the `launch_tape`
    leg - whose capture check compares a replay of every in-flight slot with the slot's eager rows.  With the grouped gather (--gather-every
    > 1, the default) that check once read buffers the loop never writes and the leg was silently skipped: both gather modes are run."""
    import json
    import os
    import subprocess
    import sys
    from tests.util import ROOT
    for gather in ("8", "1"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--pairs", "8", "--gather-every", gather,
                            "--no-cpu-baseline", "--no-boundary", "--no-fp32-path", "--no-accuracy", "--no-other-configs", "--no-autotune"],
                           capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        assert "capture failed" not in r.stderr, r.stderr[-2000:]
        line = json.loads(r.stdout.strip().splitlines()[-1])
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
            assert k in line, k
        assert line["steps"] == 6 and line["n_gpus"] == 1 and line["value"] > 0 and line["config"]["steps_per_all_gather"] == int(gather)
        tape = line.get("launch_tape")
        assert tape and tape["value"] > 0 and tape["replay"] == "launch tape", tape
