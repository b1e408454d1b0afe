"""GPU box: where does a workgroup of mask_head_kernel<64> spend its time?  In-kernel cycle stamps (tuning instantiation)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nopesac_amd import _lib, ops  # noqa: E402
dev = torch.device("cuda:0")
L = _lib.load()
L.nps_mask_head_debug_buffer.argtypes = [ctypes.c_void_p]
L.nps_mask_head_debug_buffer.restype = None
B, nq = 64, 50
c1 = (0.5 * torch.randn(B, 120, 160, 256, device=dev)).bfloat16()
t1 = (0.5 * torch.randn(B, 60, 80, 256, device=dev)).bfloat16()
wl = ops.mfma_fragment_major((torch.randn(256, 256, device=dev) / 16).bfloat16())
sc, bi = torch.ones(256, device=dev), torch.zeros(256, device=dev)
mw, mb = torch.randn(B, nq, 256, device=dev) / 16, torch.randn(B, nq, device=dev)
run = lambda: ops.mask_head(c1, t1, wl, sc, bi, mw, mb, pipe=False)
nwg = B * 120 * 160 // 128
buf = torch.zeros(nwg * 8 * 16, dtype=torch.int64, device=dev)
for _ in range(3):
    run()
L.nps_mask_head_debug_buffer(buf.data_ptr())
for _ in range(3):
    run()
torch.cuda.synchronize()
L.nps_mask_head_debug_buffer(None)
t = buf.view(nwg, 8, 16).cpu().double()
names = ["weights / c1 loads issued, c1 -> LDS", "barrier", "lateral GEMM (64 MFMA per wave)", "barrier", "BN + ReLU -> tile", "barrier", "bilinear taps, blend, add -> p1",
         "barrier", "mask GEMM (16 MFMA per wave)", "barrier", "bias + sigmoid -> f32 staging", "barrier", "probability store"]
d = t[:, :, 1:14] - t[:, :, 0:13]
print("%d workgroups of 8 waves (two per CU); cycles per phase, mean | max" % nwg)
for i, n in enumerate(names):
    print("  %-40s %8.0f | %8.0f" % (n, float(d[:, :, i].mean()), float(d[:, :, i].max())))
tot = t[:, :, 13] - t[:, :, 0]
print("  %-40s %8.0f | %8.0f" % ("total", float(tot.mean()), float(tot.max())))
