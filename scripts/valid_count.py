"""How many queries pass the score test on the bench inputs (sizes the post-selection pixel loop)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
model = bench.build_model(dev, 50, "bfloat16")
g = torch.Generator().manual_seed(1000)
raw = torch.randint(0, 256, (16, 3, 480, 640), generator=g).float().to(dev)
from nopesac_amd import ops
with torch.no_grad():
    x = ops.preprocess(raw, model.pixel_mean, model.pixel_std, model.backbone.STEM_CIN_PAD, model.compute_dtype)
    feats = model.backbone(x)
    head_out, qf = model.sem_seg_head(feats)
    lg = head_out["pred_logits"].float()
    p = torch.softmax(lg, -1)
    valid = (p[..., 0] > p[..., 1]) & (p[..., 0] > 0.6)
    print("valid queries per image:", valid.sum(1).tolist())
