"""End-to-end rate of the CLI on a Matterport3D-style split ON DISK (PNG frames): json -> data.LazyPairs (batch PNG decode on the host,
pinned batch buffers) -> uint8 images -> four batches in flight (bf16) -> package() -> evaluator.  The split is synthetic (16 distinct
480 x 640 frames, Pillow-encoded, referenced by N pairs; random-init weights): what is measured is the pipeline, two runs of different
length so that start-up (model build, routing file, the first batch's full synchronisation) cancels.
usage: python scripts/mp3d_e2e_rate.py [pairs_short=512] [pairs_long=4096]"""
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from PIL import Image  # noqa: E402

from nopesac_amd import run, runner  # noqa: E402

tape = "--tape" in sys.argv                 # MODEL.AMD.USE_HIP_GRAPH: every batch's forward replayed through the launch tape
scannet = "--scannet" in sys.argv           # 968 x 1296 JPEG frames (GPU decode + GPU resize) instead of 480 x 640 PNG frames (host decode)
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
n_short = int(argv[0]) if len(argv) > 0 else 512
n_long = int(argv[1]) if len(argv) > 1 else 4096
rng = np.random.default_rng(5)
FH, FW = (968, 1296) if scannet else (480, 640)
yy, xx = np.mgrid[0:FH, 0:FW].astype(np.float32)
with tempfile.TemporaryDirectory() as td:
    root = os.path.join(td, "datasets", "scannet_dataset" if scannet else "mp3d_dataset")
    jdir = os.path.join(root, "scannet_json" if scannet else "mp3d_planercnn_json")
    os.makedirs(jdir)
    files = []
    for i in range(16):
        a = np.stack([128 + 90 * np.sin(xx / (20 + i) + yy / 45), 128 + 70 * np.cos(yy / (17 + i)) * np.sin(xx / 70), 120 + 100 * ((xx // 80 + yy // 60) % 2)], -1)
        f = os.path.join(root, "frame_%02d.%s" % (i, "jpg" if scannet else "png"))
        im = Image.fromarray(np.clip(a + rng.normal(0, 3.0, a.shape), 0, 255).astype(np.uint8))
        if scannet:
            im.save(f, format="JPEG", quality=90, subsampling=2)
        else:
            im.save(f)
        files.append(f)
    entries = [{"rel_pose": {"position": [0.1, 0.2, 0.3], "rotation": [1.0, 0.0, 0.0, 0.0]},
                "0": {"file_name": files[(2 * k) % 16], "image_id": "h_%d_0" % k, "height": 480, "width": 640},
                "1": {"file_name": files[(2 * k + 1) % 16], "image_id": "h_%d_1" % k, "height": 480, "width": 640}} for k in range(n_long)]
    json.dump({"categories": [], "data": entries}, open(os.path.join(jdir, "cached_set_testV2.json" if scannet else "cached_set_test.json"), "w"))
    out = {"split": "scannet-style 968x1296 JPEG (GPU decode + resize)" if scannet else "mp3d-style 480x640 PNG (host decode)",
           "cpu_budget": runner.cpu_budget(), "file_kbytes": os.path.getsize(files[0]) // 1024}
    res = {}
    for label, n in (("warm-up", 64), ("short", n_short), ("long", n_long)):
        t0 = time.perf_counter()
        r = run.main(["--config-file", os.path.join(ROOT, "configs", "inference_scannet.yaml" if scannet else "inference_mp3d.yaml"), "--eval-only",
                      "--synthetic-weights", "--dataset", "scannet_test" if scannet else "mp3d_test",
                      "--datasets-dir", os.path.join(td, "datasets"), "--limit", str(n), "--pairs-per-batch", "32", "--inflight", "4", "--uint8-images",
                      "MODEL.AMD.COMPUTE_DTYPE", "bfloat16", "MODEL.AMD.ROUTING_FILE", os.path.join(ROOT, "profiles", "routing_r5.json"), "MODEL.AMD.AUTOTUNE", False,
                      "MODEL.AMD.USE_HIP_GRAPH", str(bool(tape))])
        res[label] = {"pairs": r["timing(rank0)"]["pairs"], "loop_s": round(r["timing(rank0)"]["total_s"], 3), "wall_s": round(time.perf_counter() - t0, 2)}
    d_pairs = res["long"]["pairs"] - res["short"]["pairs"]
    d_t = res["long"]["loop_s"] - res["short"]["loop_s"]
    out["launch_tape"] = tape
    out.update(res)
    out["steady_pairs_per_s"] = round(d_pairs / d_t, 1)
    out["note"] = "steady rate = (pairs_long - pairs_short) / (loop seconds long - short): inference_on_dataset's own clock around its batch loop"
    print(json.dumps(out, indent=1))
