"""Diagnostic (GPU box): absolute errors of the fp32 HIP path against the CPU oracle on the forced-K bench workload and on the
default e2e pairs, and the pose error of the bf16 bench configuration (forced K = 32, B = 32) against the fp32 HIP path under
the same forced control.  Prints one JSON object."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nopesac_amd import runner  # noqa: E402
from nopesac_amd.synth import synth_pair, synth_state_dict  # noqa: E402
from oracle import nopesac_oracle as O  # noqa: E402
from tests.util import make_model  # noqa: E402

dev = torch.device("cuda:0")
out = {}
for K, nq in ((32, 50), (64, 64), (128, 128)):
    B = 2
    model = make_model(dev, nq=nq)
    inp = [synth_pair(20 + i) for i in range(B)]
    forced = bench.make_forced(B, K, nq, dev, 5)
    with torch.no_grad():
        d = model.forward_tensors(model.preprocess_image(inp), B, 480, 640, forced=forced)
    cam = d["cam"]
    cpu_forced = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in forced.items()}
    ref = O.inference(synth_state_dict(nq), inp, O.OracleConfig(num_queries=nq), forced=cpu_forced)
    rec = {}
    for key in ("camera_init", "camera_initRec", "camera_avgRef0", "camera"):
        t, r = cam["cameras"][key]
        et = max(float(np.abs(t[b].cpu().numpy() - ref[b][key]["tran"]).max()) for b in range(B))
        er = max(float(np.abs(r[b].cpu().numpy() - ref[b][key]["rot"]).max()) for b in range(B))
        rec[key] = {"abs_tran": et, "abs_rot": er, "max|tran|": max(float(np.abs(ref[b][key]["tran"]).max()) for b in range(B))}
    out["forced_K%d" % K] = rec
    print(K, rec, flush=True)

# default e2e
model = make_model(dev)
inp = [synth_pair(0), synth_pair(3), synth_pair(5)]
res = model(inp)
ref = O.inference(synth_state_dict(50), inp, O.OracleConfig())
rec = {}
for i, (a, b) in enumerate(zip(res, ref)):
    for key in ("camera_init", "camera_initRec", "camera"):
        rec["%d.%s" % (i, key)] = [float(np.abs(a[key]["tran"] - b[key]["tran"]).max()), float(np.abs(a[key]["rot"] - b[key]["rot"]).max()),
                                   float(np.abs(b[key]["tran"]).max())]
    for v in "01":
        rec["%d.plane%s" % (i, v)] = [float((a[v]["pred_plane"].cpu() - b[v]["pred_plane"]).abs().max()), float(b[v]["pred_plane"].abs().max())]
out["default_e2e"] = rec
print(rec, flush=True)

# bf16 bench configuration vs fp32 HIP path under the same forced control
B, K, nq = 32, 32, 50
m16 = bench.build_model(dev, nq, "bfloat16")
m32 = bench.build_model(dev, nq, "float32")
g = torch.Generator().manual_seed(1000)
raw = torch.randint(0, 256, (2 * B, 3, 480, 640), generator=g).float().to(dev)
forced = bench.make_forced(B, K, nq, dev, 7)
from nopesac_amd import ops  # noqa: E402
with torch.no_grad():
    a = m16.forward_tensors(None, B, 480, 640, forced=forced, raw_images=raw)["cam"]
    x = ops.preprocess(raw, m32.pixel_mean, m32.pixel_std, m32.backbone.STEM_CIN_PAD, torch.float32)
    b = m32.forward_tensors(x, B, 480, 640, forced=forced)["cam"]
rec = {"m16": a["m"].tolist()[:4], "m32": b["m"].tolist()[:4]}
for key in ("camera_init", "camera_initRec", "camera_avgRef0", "camera"):
    t16, q16 = [v.float().cpu().numpy() for v in a["cameras"][key]]
    t32, q32 = [v.float().cpu().numpy() for v in b["cameras"][key]]
    te, re = runner.translation_error(t16, t32), runner.rotation_error_deg(q16, q32)
    rec[key] = {"T_mean": float(te.mean()), "T_max": float(te.max()), "R_mean": float(re.mean()), "R_max": float(re.max()),
                "|t|": float(np.linalg.norm(t32, axis=-1).mean())}
out["bf16_vs_fp32_forced_K32_B32"] = rec
print(rec, flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "parity_report.json"), "w"), indent=1)
