"""Where does the bf16 pose error of the benchmark configuration come from?  

The bench workload (B pairs, K matched planes forced) runs through the fp32 HIP path (the 1e-4 parity path) and through the bf16
configuration with ONE stage at a time switched back to f32 operands; the table is the pose error of each variant against the
fp32 path (formulas mp3d_evaluation.py:389-465) and what the variant costs (ms per single-stream step).

    backbone   ResNet-50 in fp32 (features rounded to bf16 once, at the end)
    decoder    pixel decoder (GN) + mask_features + the six convs_backbone layers in f32
    branches   the affinity volume stays f32, the 2 x 6 strided convs run on f32 operands
    fc         fc_trans / fc_rots + the trans / rots regressors with f32 weights
    aim        AIM re-embedding MLPs with f32 weights
    refine     the RANSAC stage's MLP stacks with f32 weights

Usage (GPU box):  python scripts/bf16_attribution.py [--pairs 32] [--out gpurun_out/bf16_attribution.json]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nopesac_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=32)
ap.add_argument("--k", type=int, default=32)
ap.add_argument("--only", default="", help="substring filter on the variant names")
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "bf16_attribution.json"))
args = ap.parse_args()
dev = torch.device("cuda:0")
B, K = args.pairs, args.k
nq = 50 if K <= 50 else K
m16, m32 = bench.build_model(dev, nq, "bfloat16"), bench.build_model(dev, nq, "float32")
g = torch.Generator().manual_seed(1000)
raw = torch.randint(0, 256, (2 * B, 3, 480, 640), generator=g).float().to(dev)
forced = bench.make_forced(B, K, nq, dev, 7)
head = m16.camera_head_list[0]
orig_backbone_forward = m16.backbone.forward


def fp32_backbone(x, raw=None, until="res5"):
    """ResNet-50 in fp32 up to and including stage `until` ("stem", "res2" .. "res5"), bf16 kernels behind it."""
    xin = ops.preprocess(raw[0], m32.pixel_mean, m32.pixel_std, m32.backbone.STEM_CIN_PAD, torch.float32) if x is None else x.float()
    if until == "res5":
        return {k: v.to(torch.bfloat16) for k, v in m32.backbone(xin).items()}
    mid = m32.backbone(xin, stop_after=until)["x"].to(torch.bfloat16)
    feats = orig_backbone_forward(None, resume=(until, mid))
    if until in ("res2", "res3", "res4"):            # the stages' own outputs that were computed in fp32
        full = m32.backbone(xin)
        for k in ("res2", "res3", "res4"):
            if k in full and ["res2", "res3", "res4"].index(k) <= ["res2", "res3", "res4"].index(until):
                feats[k] = full[k].to(torch.bfloat16)
    return feats


def ms_per_step(m, n=3):
    def one():
        with torch.no_grad():
            m.forward_tensors(None, B, 480, 640, forced=forced, raw_images=raw)
    one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        one()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


def stem_rounding_variant(x, raw=None, round_input=False, round_weights=False):
    """Which operand rounding of the bf16 stem matters?  (round 4)  The fp32 stem with ONLY its input or ONLY its weights rounded to
    bf16, its output rounded to bf16 once, the bf16 kernels behind it."""
    xin = ops.preprocess(raw[0], m32.pixel_mean, m32.pixel_std, m32.backbone.STEM_CIN_PAD, torch.float32) if x is None else x.float()
    if round_input:
        xin = xin.bfloat16().float()
    c = m32.backbone.packed["stem"]
    w = c.w(torch.float32)
    if round_weights:
        w = w.bfloat16().float()
    y = ops.maxpool(ops.conv2d(xin, w, c.scale, c.bias, stride=2, pad=3, act=ops.ACT_RELU), 3, 2, 1)
    return orig_backbone_forward(None, resume=("stem", y.to(torch.bfloat16)))


import functools  # noqa: E402
STEM_VARIANTS = {"stem: bf16 input, f32 weights": dict(round_input=True), "stem: f32 input, bf16 weights": dict(round_weights=True),
                 "stem: bf16 input and weights, f32 kernel": dict(round_input=True, round_weights=True)}
variants = [("bf16 (as timed)", (), False)] + [(p, (p,), False) for p in ("decoder", "branches", "fc", "aim", "refine")] + \
           [("backbone", (), True), ("decoder+branches+fc", ("decoder", "branches", "fc"), False), ("fc+aim+refine", ("fc", "aim", "refine"), False),
            ("branches+fc", ("branches", "fc"), False), ("all head parts", ("decoder", "branches", "fc", "aim", "refine"), False),
            ("all head parts + backbone", ("decoder", "branches", "fc", "aim", "refine"), True)] + \
           [("backbone fp32 through " + st, (), st) for st in ("stem", "res2", "res3", "res4")] + [(k, (), k) for k in STEM_VARIANTS]
table = {}
for name, parts, bb in variants:
    if args.only and args.only not in name and name != "bf16 (as timed)":
        continue
    head.fp32_parts = frozenset(parts)
    if isinstance(bb, str) and bb in STEM_VARIANTS:
        m16.backbone.forward = functools.partial(stem_rounding_variant, **STEM_VARIANTS[bb])
    else:
        m16.backbone.forward = (functools.partial(fp32_backbone, until=bb) if isinstance(bb, str) else fp32_backbone) if bb else orig_backbone_forward
    err = bench.bench_workload_pose_error(m16, m32, dev, B, K, nq, raw=raw, forced=forced)
    row = {k: {kk: err[k][kk] for kk in ("T_err_mean", "T_err_max", "R_err_deg_mean", "R_err_deg_max")} for k in ("camera_init", "camera_initRec", "camera")}
    row["ms_per_step_single_stream"] = round(ms_per_step(m16), 2)
    table[name] = row
    print("%-28s init R %.2f/%.2f T %.4f | initRec R %.2f/%.2f | camera R %.2f/%.2f T %.4f | %.2f ms" % (
        name, row["camera_init"]["R_err_deg_mean"], row["camera_init"]["R_err_deg_max"], row["camera_init"]["T_err_max"],
        row["camera_initRec"]["R_err_deg_mean"], row["camera_initRec"]["R_err_deg_max"], row["camera"]["R_err_deg_mean"],
        row["camera"]["R_err_deg_max"], row["camera"]["T_err_max"], row["ms_per_step_single_stream"]), flush=True)
head.fp32_parts = frozenset()
m16.backbone.forward = orig_backbone_forward
os.makedirs(os.path.dirname(args.out), exist_ok=True)
json.dump({"pairs": B, "K": K, "reference": "fp32 HIP path, same forced K control", "variants": table}, open(args.out, "w"), indent=1)
