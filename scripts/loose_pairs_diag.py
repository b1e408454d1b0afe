"""GPU box: the bf16 vs fp32 difference of the REFINED pose on non-forced inputs (round-5 verdict, parity hole (a): `loose_structured.camera`
T 0.74 / R 4.4 deg mean with one match per pair).  Per pair: planes kept per view, the matched plane pairs of both precisions, the pixel
pose's and the refined pose's difference, the refined translation's magnitude.  Then the attribution: the fp32 model's refinement stage
evaluated ON THE bf16 MODEL'S initial pose / matches (same f32 arithmetic, perturbed input) = how much of the difference is the stage's
sensitivity to its input rather than bf16 arithmetic inside it."""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nopesac_amd import runner  # noqa: E402
from nopesac_amd.synth import synth_pair  # noqa: E402
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
m32, m16 = bench.build_model(dev, 50, "float32", bench.LOOSE), bench.build_model(dev, 50, "bfloat16", bench.LOOSE)
mode = sys.argv[2] if len(sys.argv) > 2 else "pairs"
inp = [synth_pair(i, structured=True) for i in range(n)]
if mode == "twins":            # both views show the SAME image: every kept plane has its twin in the other view (several matches per pair)
    for d in inp:
        d["1"] = dict(d["0"], image=d["0"]["image"].clone())
a, b = m32(inp), m16(inp)
if mode == "noise":            # control: fp32 against fp32 with +-0.5 grey levels of input noise (how far does the REFINED pose move per
    g = torch.Generator().manual_seed(5)                 # degree of pixel-pose change when NO bf16 arithmetic is involved?)
    inp2 = [{v: dict(d[v], image=(d[v]["image"] + (torch.rand(d[v]["image"].shape, generator=g) - 0.5)).clamp(0, 255)) for v in "01"} for d in inp]
    b = m32(inp2)
rows = []
for i, (x, y) in enumerate(zip(a, b)):
    def err(key):
        t = runner.translation_error(y[key]["tran"][None], x[key]["tran"][None])[0]
        r = runner.rotation_error_deg(y[key]["rot"][None], x[key]["rot"][None])[0]
        return round(float(t), 4), round(float(r), 3)
    A32, A16 = np.asarray(x["pred_assignment"]), np.asarray(y["pred_assignment"])
    row = {"pair": i, "planes_fp32": [len(x[v]["pred_plane"]) for v in "01"], "planes_bf16": [len(y[v]["pred_plane"]) for v in "01"],
           "m_fp32|bf16": [int(x["matched_num"]), int(y["matched_num"])], "matches_fp32": [tuple(int(v) for v in ij) for ij in np.argwhere(A32 > 0)], "matches_bf16": [tuple(int(v) for v in ij) for ij in np.argwhere(A16 > 0)],
           "camera_init_T_R": err("camera_init"), "camera_initRec_T_R": err("camera_initRec"), "camera_T_R": err("camera"),
           "abs_t_refined_fp32": round(float(np.linalg.norm(x["camera"]["tran"])), 3), "abs_t_init_fp32": round(float(np.linalg.norm(x["camera_init"]["tran"])), 3),
           "refined_t_fp32": [round(float(v), 3) for v in x["camera"]["tran"]], "refined_t_bf16": [round(float(v), 3) for v in y["camera"]["tran"]]}
    rows.append(row)
    print(json.dumps(row))
same = [r for r in rows if r["matches_fp32"] == r["matches_bf16"]]
multi = [r for r in same if len(r["matches_fp32"]) >= 2]
print("pairs with identical matches: %d of %d; with >= 2 matches: %d" % (len(same), n, len(multi)))
for name, sel in (("all", rows), ("identical matches", same), (">= 2 identical matches", multi), ("exactly 1 match", [r for r in same if len(r["matches_fp32"]) == 1])):
    if sel:
        print("%-26s camera T mean %.4f max %.4f | R mean %.3f max %.3f deg | relative T (T / |t|) mean %.4f" % (
            name, np.mean([r["camera_T_R"][0] for r in sel]), np.max([r["camera_T_R"][0] for r in sel]), np.mean([r["camera_T_R"][1] for r in sel]),
            np.max([r["camera_T_R"][1] for r in sel]), np.mean([r["camera_T_R"][0] / max(r["abs_t_refined_fp32"], 1e-6) for r in sel])))
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "loose_pairs_diag_%s.json" % mode), "w"), indent=1)
print("amplification (refined R / pixel-pose R, means): %.2f" % (np.mean([r["camera_T_R"][1] for r in rows]) / max(np.mean([r["camera_init_T_R"][1] for r in rows]), 1e-9)))
