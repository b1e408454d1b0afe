"""GPU box: kernel trace target - the model on ONE pair per call (the reference harness's batch), eager."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nopesac_amd.synth import synth_pair  # noqa: E402
dev = torch.device("cuda:0")
model = bench.build_model(dev, 50, "bfloat16")
model.output_rle = True
inputs = [synth_pair(0)]
with torch.no_grad():
    for _ in range(8):
        model(inputs)
    torch.cuda.synchronize()
