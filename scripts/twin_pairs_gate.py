"""GPU box: candidate inputs for a gate on the REFINED pose in bf16 on non-forced inputs (round-5 verdict, parity hole (a)).  Twin pairs (both
views show the same structured image) under the relaxed thresholds give several natural matches; a pair QUALIFIES when both precisions keep
the same planes and find the same >= 2 matches (the discrete decisions agree, so the comparison is arithmetic only).  For the qualifying
pairs: bf16 vs fp32, and the control fp32 vs fp32 with +-0.5 grey levels of input noise."""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nopesac_amd import runner  # noqa: E402
from nopesac_amd.synth import synth_pair  # noqa: E402
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
m32, m16 = bench.build_model(dev, 50, "float32", bench.LOOSE), bench.build_model(dev, 50, "bfloat16", bench.LOOSE)


def twins(i0, i1, noise=None):
    out = []
    for i in range(i0, i1):
        d = synth_pair(i, structured=True)
        img = d["0"]["image"]
        if noise is not None:
            img = (img + (torch.rand(img.shape, generator=noise) - 0.5)).clamp(0, 255)
        d["0"] = dict(d["0"], image=img)
        d["1"] = dict(d["0"], image=img.clone(), image_id=d["1"]["image_id"], file_name=d["1"]["file_name"])
        out.append(d)
    return out


def err(x, y, key):
    return (float(runner.translation_error(y[key]["tran"][None], x[key]["tran"][None])[0]), float(runner.rotation_error_deg(y[key]["rot"][None], x[key]["rot"][None])[0]))


rows = []
g = torch.Generator().manual_seed(5)
for lo in range(0, n, 16):
    inp = twins(lo, min(n, lo + 16))
    a, b, c = m32(inp), m16(inp), m32(twins(lo, min(n, lo + 16), noise=g))
    for i, (x, y, z) in enumerate(zip(a, b, c)):
        same = lambda u, v: all(u[k]["pred_plane_oriIdxs"] == v[k]["pred_plane_oriIdxs"] for k in "01") and np.array_equal(
            np.asarray(u["pred_assignment_beforeRef0"]) > 0, np.asarray(v["pred_assignment_beforeRef0"]) > 0) and int(u["matched_num"]) == int(v["matched_num"])
        rows.append({"pair": lo + i, "m": int(x["matched_num"]), "same_bf16": bool(same(x, y)), "same_noise": bool(same(x, z)), "abs_t": float(np.linalg.norm(x["camera"]["tran"])),
                     "bf16": {k: err(x, y, k) for k in ("camera_init", "camera")}, "noise": {k: err(x, z, k) for k in ("camera_init", "camera")}})
q = [r for r in rows if r["same_bf16"] and r["m"] >= 2]
qn = [r for r in rows if r["same_noise"] and r["m"] >= 2]
print("%d twin pairs; m histogram (fp32): %s; qualifying (same planes, same >= 2 matches): bf16 %d, noise control %d" % (
    n, dict(zip(*np.unique([r["m"] for r in rows], return_counts=True))), len(q), len(qn)))
for name, sel, key in (("bf16 vs fp32", q, "bf16"), ("fp32 + noise vs fp32", qn, "noise")):
    if sel:
        T = np.array([r[key]["camera"][0] / max(r["abs_t"], 1e-9) for r in sel]); R = np.array([r[key]["camera"][1] for r in sel])
        Ri = np.array([r[key]["camera_init"][1] for r in sel])
        print("%-22s refined camera: T / |t| mean %.4f max %.4f | R mean %.3f max %.3f deg   (pixel pose R mean %.3f max %.3f)" % (name, T.mean(), T.max(), R.mean(), R.max(), Ri.mean(), Ri.max()))
print("qualifying pairs (bf16):", [(r["pair"], r["m"], round(r["bf16"]["camera"][0] / r["abs_t"], 4), round(r["bf16"]["camera"][1], 3)) for r in q])
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "twin_pairs_gate.json"), "w"), indent=1)
