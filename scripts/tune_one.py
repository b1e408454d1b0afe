"""Print the autotuner's timing table for one GEMM / conv shape: python scripts/tune_one.py M K N [out_dtype bf16|f32] [res]."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nopesac_amd import ops
M, K, N = [int(v) for v in sys.argv[1:4]]
od = torch.bfloat16 if (len(sys.argv) < 5 or sys.argv[4] == "bf16") else torch.float32
dev = torch.device("cuda:0")
x = torch.randn(M, K, device=dev).bfloat16()
w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
b = torch.zeros(N, device=dev)
res = torch.randn(M, N, device=dev).to(od) if "res" in sys.argv else None
ops.TUNER.measuring = True
y = ops.linear(x, w, b, residual=res, out_dtype=od)
ops.TUNER.measuring = False
for key, cfg, times in ops.TUNER.log:
    print("chosen", cfg, {k: round(v * 1000, 1) for k, v in times.items()})
