"""GPU box: where does a workgroup of conv3x3_c64_kernel spend its time?  In-kernel cycle stamps at the phase boundaries (tuning instantiation)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nopesac_amd import _lib, ops  # noqa: E402
dev = torch.device("cuda:0")
L = _lib.load()
L.nps_c64_debug_buffer.argtypes = [ctypes.c_void_p]
L.nps_c64_debug_buffer.restype = None
x = torch.randn(64, 120, 160, 64, device=dev).bfloat16().relu()
w = (torch.randn(64, 3, 3, 64, device=dev) / 24).bfloat16()
sc, bi = torch.ones(64, device=dev), torch.zeros(64, device=dev)
run = lambda: ops.conv3x3_c64(x, w, sc, bi)
nwg = 10 * 8 * 64
buf = torch.zeros(nwg * 4 * 8, dtype=torch.int64, device=dev)
for _ in range(3):
    run()
L.nps_c64_debug_buffer(buf.data_ptr())
for _ in range(3):
    run()
torch.cuda.synchronize()
L.nps_c64_debug_buffer(None)
t = buf.view(nwg, 4, 8).cpu().double()
names = ["weight ring issue, halo loads, ds_write", "barrier", "implicit GEMM (144 MFMA per wave)", "barrier", "BN + act -> staging (LDS)", "barrier", "stores"]
d = t[:, :, 1:8] - t[:, :, 0:7]
print("%d workgroups of 4 waves (three per CU); cycles per phase, mean | max" % nwg)
for i, n in enumerate(names):
    print("  %-42s %8.0f | %8.0f" % (n, float(d[:, :, i].mean()), float(d[:, :, i].max())))
tot = t[:, :, 7] - t[:, :, 0]
print("  %-42s %8.0f | %8.0f   (MFMA floor with three workgroups per CU: 144 x 32 x 3 = 13824 per SIMD)" % ("total", float(tot.mean()), float(tot.max())))
