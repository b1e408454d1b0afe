"""GPU box: where does a workgroup of the fused stem spend its time?  In-kernel cycle stamps at the phase boundaries (tuning instantiation)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nopesac_amd import _lib, ops  # noqa: E402
dev = torch.device("cuda:0")
L = _lib.load()
L.nps_stem_debug_buffer.argtypes = [ctypes.c_void_p]
L.nps_stem_debug_buffer.restype = None
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
w = (torch.randn(64, 224, device=dev) / 12).bfloat16()
sc, bi = torch.ones(64, device=dev), torch.zeros(64, device=dev)
img = torch.randint(0, 256, (B, 3, 480, 640), device=dev).float()
pad3 = torch.tensor([123.675, 116.28, 103.53], device=dev) - 128.0
run = lambda: ops.stem_fused_raw_shifted(img, pad3, w, sc, bi)
nwg = 8 * 30 * B
buf = torch.zeros(nwg * 4 * 16, dtype=torch.int64, device=dev)
for _ in range(3):
    run()
L.nps_stem_debug_buffer(buf.data_ptr())
for _ in range(3):
    run()
torch.cuda.synchronize()
L.nps_stem_debug_buffer(None)
t = buf.view(nwg, 4, 16).cpu().double()
names = ["patch: loads, convert, ds_write", "weights: loads, ds_write", "barrier", "implicit GEMM (84 MFMA per wave)", "barrier", "BN + ReLU -> conv tile (LDS)", "barrier",
         "max-pool from LDS + stores"]
d = t[:, :, 1:9] - t[:, :, 0:8]
print("%d images, %d workgroups of 4 waves (three per CU); cycles per phase, mean | max" % (B, nwg))
for i, n in enumerate(names):
    print("  %-36s %8.0f | %8.0f" % (n, float(d[:, :, i].mean()), float(d[:, :, i].max())))
tot = t[:, :, 8] - t[:, :, 0]
print("  %-36s %8.0f | %8.0f   (MFMA floor with three workgroups per CU: 84 x 32 x 3 = 8064 per SIMD)" % ("total", float(tot.mean()), float(tot.max())))
