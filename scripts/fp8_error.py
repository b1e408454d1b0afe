"""Pose error of the fp8 backbone mode (BASELINE configs[4]: fp8 3x3 backbone convs, K = 128, nq = 128) and of the plain bf16 mode against the
fp32 HIP path on the same inputs and K control - the numbers tests/test_e2e_gpu.py::test_config5_fp8_backbone_k128 gates.  usage: fp8_error.py [pairs]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nopesac_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
K, nq = 128, 128
dev = torch.device("cuda:0")
m32 = bench.build_model(dev, nq, "float32")
m16 = bench.build_model(dev, nq, "bfloat16")
m8 = bench.build_model(dev, nq, "bfloat16", ["MODEL.AMD.BACKBONE_FP8", True])
g = torch.Generator().manual_seed(1000)
raw = torch.randint(0, 256, (2 * B, 3, 480, 640), generator=g).float().to(dev)
forced = bench.make_forced(B, K, nq, dev, 7)
with torch.no_grad():
    m8.backbone.calibrate_fp8(ops.preprocess(raw[:4], m8.pixel_mean, m8.pixel_std, m8.backbone.STEM_CIN_PAD, m8.compute_dtype))
out = {}
for name, m in (("bf16", m16), ("fp8", m8)):
    e = bench.bench_workload_pose_error(m, m32, dev, B, K, nq, raw=raw, forced=forced)
    out[name] = {k: e[k] for k in ("camera_init", "camera_initRec", "camera")}
    print(name, json.dumps(out[name]), flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "fp8_error.json"), "w"), indent=1)
