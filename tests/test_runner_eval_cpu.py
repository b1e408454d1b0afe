"""CPU: the evaluator / CLI host logic (no GPU): metric formulas vs hand values, process/evaluate bookkeeping,
config -> CLI wiring, ragged multi-rank gather (gloo, world size 2)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from tests.util import ROOT


def _out(t, q, n1=2, n2=3, m=1):
    cam = {"tran": np.asarray(t, np.float32), "rot": np.asarray(q, np.float32)}
    return {"0": {"pred_plane": torch.zeros(n1, 3)}, "1": {"pred_plane": torch.zeros(n2, 3)}, "matched_num": m,
            "camera": cam, "camera_init": cam}


def test_camera_metrics_table():
    from nopesac_amd.evaluation import camera_metrics, create_small_table
    gt_t = np.zeros((4, 3), np.float32)
    gt_q = np.tile(np.array([1, 0, 0, 0], np.float32), (4, 1))
    pt = np.array([[0.1, 0, 0], [0.4, 0, 0], [0.9, 0, 0], [3, 0, 0]], np.float32)
    ang = np.radians([4.0, 12.0, 28.0, 90.0]) / 2
    pq = np.stack([np.cos(ang), np.sin(ang), 0 * ang, 0 * ang], 1).astype(np.float32)
    pq[1] *= -1                                                     # q and -q are the same rotation
    m = camera_metrics(pt, pq, gt_t, gt_q)
    assert m["T err < 0.2"] == 25.0 and m["T err < 0.5"] == 50.0 and m["T err < 1.0"] == 75.0
    assert m["R err < 10"] == 25.0 and m["R err < 15"] == 50.0 and m["R err < 30"] == 75.0
    assert abs(m["T median err"] - 0.65) < 1e-6 and abs(m["R median err"] - 20.0) < 1e-3
    assert create_small_table(m).count("\n") == 2


def test_pose_evaluator_single_process():
    from nopesac_amd.evaluation import PoseEvaluator
    ev = PoseEvaluator(camera_keys=("camera", "camera_init"))
    inputs = [{"rel_pose": {"position": [0.0, 0, 0], "rotation": [1.0, 0, 0, 0]}}, {}]
    outputs = [_out([0.3, 0.4, 0.0], [1, 0, 0, 0]), _out([9, 9, 9], [0, 1, 0, 0])]
    ev.process(inputs, outputs)
    res = ev.evaluate()
    assert res["pairs"]["count"] == 2 and res["pairs"]["mean planes/view"] == 2.5
    assert abs(res["camera"]["T mean err"] - 0.5) < 1e-6 and res["camera"]["R mean err"] < 1e-3   # only the pair with GT counts


def test_cli_setup_and_synthetic_dataset():
    from nopesac_amd import run
    args = run.default_argument_parser().parse_args(
        ["--config-file", os.path.join(ROOT, "configs", "inference_scannet.yaml"), "--eval-only", "--synthetic-pairs", "3",
         "MODEL.DEVICE", "cpu", "TEST.PLANE_SCORE_THRESHOLD", "0.7"])
    cfg = run.setup(args)
    assert cfg.MODEL.DEVICE == "cpu" and cfg.TEST.PLANE_SCORE_THRESHOLD == 0.7 and cfg.DATASETS.TEST == ("scannet_test",)
    pairs = run.load_pairs(args)
    assert len(pairs) == 3 and pairs[0]["0"]["image"].shape == (3, 480, 640)
    with pytest.raises(FileNotFoundError):
        run.load_checkpoint(torch.nn.Linear(1, 1), cfg, synthetic=False)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from nopesac_amd import runner
    from nopesac_amd.evaluation import PoseEvaluator
    runner.init_distributed("gloo")
    lo, hi = runner.shard_range(5, rank, world)                       # ragged: 3 + 2 pairs
    ev = PoseEvaluator(camera_keys=("camera",), device=torch.device("cpu"))
    for i in range(lo, hi):
        ev.process([{"rel_pose": {"position": [0.0, 0, 0], "rotation": [1.0, 0, 0, 0]}}], [_out([float(i), 0, 0], [1, 0, 0, 0])])
    res = ev.evaluate()
    torch.distributed.barrier()
    q.put((rank, res))
    torch.distributed.destroy_process_group()


def test_ragged_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, res in got:
        assert res["pairs"]["count"] == 5
        assert abs(res["camera"]["T mean err"] - 2.0) < 1e-6 and abs(res["camera"]["T median err"] - 2.0) < 1e-6


def test_prediction_dumps(tmp_path):
    """`eval_full_scene` artefacts: NopeSAC_instances_predictions.pth (prediction schema of mp3d_evaluation.py:193-257)
    and continuous.pkl (get_optimized_dict :259-313)."""
    import pickle
    from nopesac_amd.evaluation import PoseEvaluator, dump_predictions
    ev = PoseEvaluator(camera_keys=("camera",), keep_predictions=True)
    inp = {"0": {"image_id": "a_0", "file_name": "a0.png"}, "1": {"image_id": "a_1", "file_name": "a1.png"},
           "rel_pose": {"position": [0.0, 0, 1], "rotation": [1.0, 0, 0, 0], "tran_cls": 3, "rot_cls": 4}}
    out = _out([0.3, 0.4, 0.0], [1, 0, 0, 0])
    seg = {"size": [2, 2], "counts": b"04"}
    out["0"]["instances"] = [{"image_id": "a_0", "file_name": "a0.png", "category_id": 0, "score": 0.9, "segmentation": seg,
                              "bbox": [0.0, 0.0, 2.0, 2.0], "bbox_mode": 1}] * 2
    out["1"]["instances"] = []
    out["pred_assignment"] = torch.tensor([[0., 1, 0], [0, 0, 0]])
    out["camera_onePP"] = {"tran": np.zeros((2, 3), np.float32), "rot": np.zeros((2, 4), np.float32)}
    ev.process([inp], [out])
    files = dump_predictions(ev._predictions, str(tmp_path))
    preds = torch.load(files["instances_predictions"], weights_only=False)
    assert len(preds) == 1 and preds[0]["0"]["instances"][0]["segmentation"] == seg
    assert preds[0]["camera"]["gts"]["tran_cls"] == 3 and "camera_onePP" in preds[0] and preds[0]["corrs"] == {"0": {}, "1": {}}
    cont = pickle.load(open(files["continuous"], "rb"))
    assert set(cont[0]) == {"n_corr", "cost", "best_camera", "gt_camera", "best_assignment", "plane_param_override", "image_ids"}
    assert cont[0]["n_corr"] == 1 and cont[0]["image_ids"] == {"0": "a_0", "1": "a_1"}
    assert cont[0]["plane_param_override"]["1"].shape == (3, 3) and cont[0]["gt_camera"]["position"] == [0.0, 0, 1]


@pytest.mark.parametrize("seed", [3, 4])
def test_evaluate_for_matchings_matches_the_reference(seed):
    """Plane-matching P / R / F (mp3d_evaluation.py:746-849) on the seeded case of tests/golden_inputs.py against the numbers the
    imported reference function produced (tests/golden/G_matching_eval_*.npz, oracle/gen_golden.py stage G).  The product decodes
    compressed COCO RLE strings and takes the IoU on dense masks; the fixture side used run-merging on uncompressed RLEs."""
    from nopesac_amd import evaluation as E
    from nopesac_amd import rle
    from oracle import rle_oracle as R
    from tests import golden_inputs as GI
    from tests.util import gold
    case = GI.matching_eval_case(seed)
    g = gold(f"G_matching_eval_{seed}")
    keys = ("pred_assignment", "pred_assignment_afterRef0", "pred_assignment_beforeRef0")
    preds, dataset = [], {}
    for pi, pr in enumerate(case):
        ids = (f"a{pi}", f"b{pi}")
        pred, entry = {}, {"gt_corrs": pr["gt_corrs"]}
        for v, vid in zip("01", ids):
            view = pr["views"][int(v)]
            segs = [R.encode(m) for m in view["pred"]]
            assert all(np.array_equal(rle.decode(s), m) for s, m in zip(segs, view["pred"]))       # product decoder round trip
            pred[v] = {"image_id": vid, "instances": [{"segmentation": s} for s in segs]}
            entry[v] = {"annotations": [{"segmentation": {"size": list(m.shape), "counts": R.run_lengths(m)}} for m in view["gt"]]}   # uncompressed
        for k in keys:
            pred[k] = torch.from_numpy(pr[k])
        dataset[ids[0] + "__" + ids[1]] = entry
        preds.append(pred)
    res = E.evaluate_for_matchings(preds, dataset)
    assert set(res) == set(keys)
    for k in keys:
        want = g[k].numpy()
        got = np.array([res[k][n] for n in ("precision", "recall", "F-score", "TP", "Pred. Num.", "GT Num.")], np.float64)
        if np.isnan(want[0]):                 # the reference divides by zero when nothing was matched; the drop-in reports zeros
            assert got[4] == 0 and got[0] == 0.0 and got[2] == 0.0
        else:
            np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)
    assert 0 < res["pred_assignment_afterRef0"]["precision"] < res["pred_assignment"]["precision"] <= 1.0
    with pytest.raises(TypeError, match="RLE dict"):
        bad = {k: ({**v, "0": {"annotations": [{"segmentation": [[0, 0, 1, 1, 2, 2]]}]}} if isinstance(v, dict) else v) for k, v in dataset.items()}
        E.evaluate_for_matchings(preds, bad)


def test_rle_iou_small_cases():
    from nopesac_amd import rle
    from oracle import rle_oracle as R
    a = np.zeros((4, 5), bool); a[1:3, 1:4] = True
    b = np.zeros((4, 5), bool); b[2:4, 2:5] = True
    m = rle.iou([R.encode(a), R.encode(b)], [R.encode(a), R.encode(b), R.encode(np.zeros((4, 5), bool))])
    assert m.shape == (2, 3) and m[0, 0] == 1.0 and m[1, 1] == 1.0 and abs(m[0, 1] - 2 / 10) < 1e-12 and m[0, 2] == 0.0
    assert rle.iou([], [R.encode(a)]).shape == (0, 1)
    assert abs(rle.iou([R.encode(b)], [R.encode(a)], [1])[0, 0] - 2 / 6) < 1e-12          # crowd: intersection / area(dt)
