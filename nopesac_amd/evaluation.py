"""Camera-pose evaluator counterpart of `MP3DEvaluator` (evaluation/mp3d_evaluation.py), restricted to the part
that consumes the hot path's outputs: per-pair pose errors and their summary table.

  * process(inputs, outputs)  ~ mp3d_evaluation.py:184-257: collects every output key containing "camera"
    together with the ground truth `input["rel_pose"]{"position","rotation"}`;
  * evaluate()                ~ :315-366 + `_eval_camera_reg` :382-425: gathers across ranks (ONE all_gather of
    fixed-width fp32 rows over RCCL instead of pickled predictions over Gloo), then
    T err = ||t - t_gt||2, R err = 2 acos|<q, q_gt>| 180/pi, medians / means / accuracy at 1.0|0.5|0.2 m, 30|15|10 deg.
  * dump(output_dir)          ~ :330-341 (`eval_full_scene`): `NopeSAC_instances_predictions.pth` (torch.save of the
    per-pair prediction dicts, schema of :193-257) and `continuous.pkl` (`get_optimized_dict` :259-313 + `save_dict`
    :852-860) - the two files the reference's offline eval.py / vis tools read.
  * evaluate_for_matchings()  ~ :746-849: plane-matching precision / recall / F-score from the predictions' RLE instances, the
    GT annotations' RLE masks and `gt_corrs` (mask IoU from nopesac_amd/rle.py instead of pycocotools.mask.iou).
Plane AP (COCO tooling, pycocotools.cocoeval) stays out of scope.
"""
from __future__ import annotations

import os
import pickle
from typing import Dict, List, Optional

import numpy as np
import torch

from . import runner


def camera_metrics(pred_tran: np.ndarray, pred_rot: np.ndarray, gt_tran: np.ndarray, gt_rot: np.ndarray) -> Dict[str, float]:
    """The reference's camera table (mp3d_evaluation.py:382-418), same keys."""
    t_err = runner.translation_error(pred_tran, gt_tran)
    r_err = runner.rotation_error_deg(pred_rot, gt_rot)
    n = max(len(t_err), 1)
    return {
        "T median err": float(np.median(t_err)), "T mean err": float(np.mean(t_err)),
        "T err < 1.0": 100.0 * float((t_err < 1.0).sum()) / n, "T err < 0.5": 100.0 * float((t_err < 0.5).sum()) / n,
        "T err < 0.2": 100.0 * float((t_err < 0.2).sum()) / n,
        "R median err": float(np.median(r_err)), "R mean err": float(np.mean(r_err)),
        "R err < 30": 100.0 * float((r_err < 30).sum()) / n, "R err < 15": 100.0 * float((r_err < 15).sum()) / n,
        "R err < 10": 100.0 * float((r_err < 10).sum()) / n,
    }


class PoseEvaluator:
    """DatasetEvaluator-style: reset() / process(inputs, outputs) / evaluate()."""

    def __init__(self, camera_keys=("camera", "camera_init", "camera_initRec", "camera_avgRef0", "camera_softRef0"),
                 device: Optional[torch.device] = None, keep_predictions: bool = False):
        self.camera_keys = tuple(camera_keys)
        self.device = device
        self.keep_predictions = keep_predictions
        self.reset()

    def reset(self):
        self._rows: List[np.ndarray] = []
        self._predictions: List[dict] = []

    @staticmethod
    def prediction_record(inp: dict, out: dict) -> dict:
        """One entry of the reference's `self._predictions` (mp3d_evaluation.py:193-257), CPU-only objects."""
        gt = inp.get("rel_pose") or {}
        gt_cam = {"tran": gt.get("position"), "rot": gt.get("rotation"), "tran_cls": gt.get("tran_cls"), "rot_cls": gt.get("rot_cls")}
        pred = {"0": {}, "1": {}}
        for v in "01":
            pred[v]["image_id"] = inp[v].get("image_id")
            pred[v]["file_name"] = inp[v].get("file_name")
            if out[v] is not None and "instances" in out[v]:
                pred[v]["instances"] = out[v]["instances"]
            pred[v]["pred_plane"] = out[v]["pred_plane"].detach().cpu().clone()      # package() hands out views of one pinned buffer per batch
        for k, val in out.items():
            if "camera" in k and "cls" not in k:
                # own copies: package() hands out numpy views of the batch's pinned host buffer, which would stay alive with the record
                pred[k] = {"pred": {kk: (np.array(vv) if isinstance(vv, np.ndarray) else vv) for kk, vv in val.items()} if isinstance(val, dict) else val,
                           "gts": gt_cam}
            elif "assignment" in k:
                pred[k] = val.detach().cpu().clone()
        pred["corrs"] = {"0": {}, "1": {}}
        return pred

    def process(self, inputs: List[dict], outputs: List[dict]):
        for inp, out in zip(inputs, outputs):
            gt = inp.get("rel_pose")
            gt_t = np.asarray(gt["position"], dtype=np.float32) if gt else np.zeros(3, np.float32)
            gt_q = np.asarray(gt["rotation"], dtype=np.float32) if gt else np.array([1, 0, 0, 0], np.float32)
            row = [gt_t, gt_q, np.array([1.0 if gt else 0.0, len(out["0"]["pred_plane"]), len(out["1"]["pred_plane"]),
                                          float(out.get("matched_num", 0))], np.float32)]
            for k in self.camera_keys:
                cam = out.get(k)
                row.append(np.asarray(cam["tran"], np.float32).reshape(-1)[:3] if cam else np.zeros(3, np.float32))
                row.append(np.asarray(cam["rot"], np.float32).reshape(-1)[:4] if cam else np.array([1, 0, 0, 0], np.float32))
            self._rows.append(np.concatenate(row))
            if self.keep_predictions:
                self._predictions.append(self.prediction_record(inp, out))

    @property
    def row_width(self) -> int:
        return 3 + 4 + 4 + 7 * len(self.camera_keys)

    def evaluate(self) -> Dict[str, dict]:
        local = np.stack(self._rows) if self._rows else np.zeros((0, self.row_width), np.float32)
        rows = torch.from_numpy(local)
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            dev = self.device or (torch.device("cuda", torch.cuda.current_device()) if torch.distributed.get_backend() == "nccl"
                                  else torch.device("cpu"))
            # ranks may hold different pair counts (ceil sharding): pad to the max, gather, drop the padding
            n = torch.tensor([rows.shape[0]], device=dev)
            counts = [torch.zeros_like(n) for _ in range(torch.distributed.get_world_size())]
            torch.distributed.all_gather(counts, n)
            nmax = int(max(c.item() for c in counts))
            padded = torch.zeros(nmax, self.row_width, device=dev)
            padded[: rows.shape[0]] = rows.to(dev)
            allrows = runner.gather_metrics(padded).cpu()
            rows = torch.cat([allrows[r * nmax: r * nmax + int(c.item())] for r, c in enumerate(counts)], 0)
        r = rows.numpy()
        res: Dict[str, dict] = {"pairs": {"count": int(r.shape[0]), "mean planes/view": float(r[:, 8:10].mean()) if len(r) else 0.0,
                                          "mean matches": float(r[:, 10].mean()) if len(r) else 0.0}}
        has_gt = r[:, 7] > 0 if len(r) else np.zeros(0, bool)
        for i, k in enumerate(self.camera_keys):
            o = 11 + 7 * i
            if has_gt.any():
                res[k] = camera_metrics(r[has_gt, o:o + 3], r[has_gt, o + 3:o + 7], r[has_gt, 0:3], r[has_gt, 3:7])
        return res


def evaluate_for_matchings(predictions: List[dict], dataset_dict: Dict[str, dict], iou_thresh: float = 0.5) -> Dict[str, dict]:
    """Plane-matching precision / recall / F-score (mp3d_evaluation.py:746-849).  For every pair: each predicted plane of a view
    is assigned the GT plane with the highest mask IoU (instances[k]["segmentation"] vs the GT annotations' RLE masks); a
    predicted correspondence (i, j) counts as correct when both IoUs reach `iou_thresh` and [gt_i, gt_j] is one of the pair's
    `gt_corrs`.  precision = TP / #predicted, recall = TP / #GT, over all pairs, separately for every "*assignment*" key of the
    predictions.  Returns {assignment key: {"precision", "recall", "F-score", "TP", "Pred. Num.", "GT Num."}} - the reference logs
    one table per key and returns only the last one; it also ignores its iou_thresh argument (0.5 is hard-coded, :830) and divides by
    zero when nothing was matched (here: 0.0).  GT masks must be RLE dicts (compressed or not); polygon annotations need
    cocoapi's rasteriser (frPyObjects), which is not part of this package."""
    from . import rle
    keys = [k for k in predictions[0] if "assignment" in k] if predictions else []
    stats = {k: {"tp": 0, "pred": 0} for k in keys}
    gt_total = 0
    for pred in predictions:
        pair = dataset_dict[pred["0"]["image_id"] + "__" + pred["1"]["image_id"]]
        gt_corr = {(int(a), int(b)) for a, b in pair["gt_corrs"]}
        gt_total += len(pair["gt_corrs"])
        best_iou, best_gt = [], []
        for v in ("0", "1"):
            gt_rles = []
            for ann in pair[v]["annotations"]:
                seg = ann["segmentation"]
                if not isinstance(seg, dict):
                    raise TypeError("evaluate_for_matchings: GT segmentation must be an RLE dict (polygons need cocoapi frPyObjects)")
                gt_rles.append(seg)
            m = rle.iou([ins["segmentation"] for ins in pred[v]["instances"]], gt_rles, [0] * len(gt_rles))
            if m.shape[1] == 0:
                best_iou.append(np.zeros(m.shape[0])); best_gt.append(np.full(m.shape[0], -1))
            else:
                best_iou.append(m.max(-1)); best_gt.append(m.argmax(-1))       # first maximum, like torch.max
        for k in keys:
            A = pred[k]
            A = A.detach().cpu().numpy() if torch.is_tensor(A) else np.asarray(A)
            idx = np.argwhere(A != 0)
            stats[k]["pred"] += int(idx.shape[0])
            for i, j in idx:
                if best_iou[0][i] >= iou_thresh and best_iou[1][j] >= iou_thresh and (int(best_gt[0][i]), int(best_gt[1][j])) in gt_corr:
                    stats[k]["tp"] += 1
    out = {}
    for k in keys:
        tp, npred = stats[k]["tp"], stats[k]["pred"]
        prec = tp / npred if npred else 0.0
        rec = tp / gt_total if gt_total else 0.0
        out[k] = {"precision": prec, "recall": rec, "F-score": 2 * prec * rec / (prec + rec) if prec + rec > 0 else 0.0,
                  "TP": tp, "Pred. Num.": npred, "GT Num.": gt_total}
    return out


def optimized_dict(predictions: List[dict]) -> Dict[int, dict]:
    """`get_optimized_dict` (mp3d_evaluation.py:259-313): the `continuous.pkl` schema read by eval.py:1027-1038."""
    ret = {}
    for idx, p in enumerate(predictions):
        best = p["pred_assignment"].numpy()
        cam = p["camera"]
        ret[idx] = {
            "n_corr": best.sum(), "cost": 0.1,
            "best_camera": {"position": cam["pred"]["tran"], "rotation": cam["pred"]["rot"]},
            "gt_camera": {"position": cam["gts"]["tran"], "rotation": cam["gts"]["rot"]},
            "best_assignment": best,
            "plane_param_override": {"0": p["0"]["pred_plane"].cpu().numpy(), "1": p["1"]["pred_plane"].cpu().numpy()},
            "image_ids": {"0": p["0"]["image_id"], "1": p["1"]["image_id"]},
        }
    return ret


def dump_predictions(predictions: List[dict], output_dir: str) -> Dict[str, str]:
    """Rank-0 side of `eval_full_scene` (mp3d_evaluation.py:330-341).  With a process group the per-rank lists are
    concatenated in rank order first (comm.gather semantics, :317-319); non-zero ranks return {}."""
    dist = torch.distributed
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        parts = [None] * dist.get_world_size()
        dist.all_gather_object(parts, predictions)
        if dist.get_rank() != 0:
            return {}
        predictions = [p for part in parts for p in part]
    os.makedirs(output_dir, exist_ok=True)
    inst = os.path.join(output_dir, "NopeSAC_instances_predictions.pth")
    torch.save(predictions, inst)
    cont = os.path.join(output_dir, "continuous.pkl")
    with open(cont, "wb") as f:
        pickle.dump(optimized_dict(predictions), f)
    return {"instances_predictions": inst, "continuous": cont}


def create_small_table(d: Dict[str, float]) -> str:
    keys, vals = list(d), ["%.4f" % d[k] for k in d]
    w = [max(len(k), len(v)) for k, v in zip(keys, vals)]
    return "\n".join(["| " + " | ".join(k.ljust(x) for k, x in zip(keys, w)) + " |",
                      "|" + "|".join("-" * (x + 2) for x in w) + "|",
                      "| " + " | ".join(v.ljust(x) for v, x in zip(vals, w)) + " |"])
