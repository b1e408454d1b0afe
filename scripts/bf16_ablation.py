"""bf16-vs-fp32 agreement on structured synthetic pairs under relaxed thresholds, with the bf16-only fused kernels toggled
(analysis aid: tells rounding-level differences from a broken fused kernel)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from nopesac_amd import runner  # noqa: E402
from nopesac_amd.synth import synth_pair  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    inp = [synth_pair(i, structured=True) for i in range(n)]
    m32 = bench.build_model(dev, 50, "float32", bench.LOOSE)
    ref = m32(inp)
    del m32
    m16 = bench.build_model(dev, 50, "bfloat16", bench.LOOSE)
    configs = {"all fused": {}, "no fused gnn": {"gnn": False}, "no encoder tail": {"enc": False}, "no bottleneck tail": {"tail": False},
               "no fused stem": {"stem": False}, "none fused": {"gnn": False, "enc": False, "tail": False, "stem": False}}
    for name, c in configs.items():
        m16.matching_head.fused_gnn = c.get("gnn", True)
        m16.sem_seg_head.fused_encoder_tail = c.get("enc", True)
        m16.backbone.fused_tail = c.get("tail", True)
        m16.backbone.fused_stem = c.get("stem", True)
        out = m16(inp)
        same = [i for i in range(n) if all(a[v]["pred_plane_oriIdxs"] == b[v]["pred_plane_oriIdxs"] for v in "01" for a, b in [(ref[i], out[i])])
                and torch.equal(ref[i]["pred_assignment"], out[i]["pred_assignment"])]
        t_err = runner.translation_error(np.stack([o["camera"]["tran"] for o in out]), np.stack([o["camera"]["tran"] for o in ref]))
        r_err = runner.rotation_error_deg(np.stack([o["camera"]["rot"] for o in out]), np.stack([o["camera"]["rot"] for o in ref]))
        ti = runner.translation_error(np.stack([o["camera_init"]["tran"] for o in out]), np.stack([o["camera_init"]["tran"] for o in ref]))
        print("%-20s identical plane sets+matches: %2d/%d | camera T err median %.4f mean-on-identical %.4f | R err median %.2f mean-on-identical %.2f | init T %.4f"
              % (name, len(same), n, float(np.median(t_err)), float(t_err[same].mean()) if same else float("nan"), float(np.median(r_err)),
                 float(r_err[same].mean()) if same else float("nan"), float(ti.mean())))


if __name__ == "__main__":
    main()
