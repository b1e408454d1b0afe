"""GPU box: pairs/s of the CLI runner's batch loop (nopesac_amd.run.inference_on_dataset) on synthetic pairs, by --inflight."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nopesac_amd import run  # noqa: E402
from nopesac_amd.config import get_cfg  # noqa: E402
from nopesac_amd.evaluation import PoseEvaluator  # noqa: E402
from nopesac_amd.registry import build_model  # noqa: E402
from nopesac_amd.synth import synth_pair, synth_state_dict  # noqa: E402
B, N = int(os.environ.get("B", "32")), int(os.environ.get("N", "1024"))
cfg = get_cfg()
cfg.merge_from_file(os.path.join(ROOT, "configs", "inference_mp3d.yaml"))
cfg.merge_from_list(["MODEL.DEVICE", "cuda", "MODEL.AMD.COMPUTE_DTYPE", "bfloat16", "MODEL.AMD.ROUTING_FILE", os.path.join(ROOT, "profiles", "routing_r5.json")])
model = build_model(cfg).eval()
model.load_state_dict(synth_state_dict(50))
run.tune_kernels(model, cfg, B)
base = [synth_pair(i) for i in range(B)]
for p in base:
    for v in "01":
        p[v]["image"] = p[v]["image"].pin_memory()
pairs = [base[i % B] for i in range(N)]
import gc
if os.environ.get("FREEZE", "1") == "1":
    gc.collect(); gc.freeze()
for depth in (1, 2, 4):
    ev = PoseEvaluator()
    run.inference_on_dataset(model, pairs[:2 * B], ev, B, inflight=depth)          # warm-up
    t = run.inference_on_dataset(model, pairs, PoseEvaluator(), B, inflight=depth)
    print("runner B=%d inflight=%d: %.1f pairs/s (%.2f ms per batch; compute %.2f ms per batch)" % (B, depth, t["pairs"] / t["total_s"], 1e3 * t["total_s"] * B / t["pairs"], 1e3 * t["compute_s"] * B / t["pairs"]), flush=True)
