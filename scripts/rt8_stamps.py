"""GPU box: where does a workgroup of pw_chain_rt8_kernel (res3's edge tails) spend its time?  In-kernel cycle stamps (tuning instantiation).
usage: rt8_stamps.py [proj|cn256]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nopesac_amd import _lib, ops  # noqa: E402
dev = torch.device("cuda:0")
L = _lib.load()
L.nps_rt8_debug_buffer.argtypes = [ctypes.c_void_p]
L.nps_rt8_debug_buffer.restype = None
mode = sys.argv[1] if len(sys.argv) > 1 else "proj"
B, OH, OW, C, C4 = 64, 60, 80, 128, 512
CN, C2, stride = (128, 256, 2) if mode == "proj" else (256, 0, 1)
rn = lambda *s, k=1.0: (torch.randn(*s, device=dev) * k).bfloat16()
b = rn(B, OH, OW, C)
w3 = ops.mfma_fragment_major(rn(C4, C, k=C ** -0.5))
s3, b3 = torch.ones(C4, device=dev), torch.zeros(C4, device=dev)
kw = {}
if C2:
    kw.update(x2=rn(B, OH * stride, OW * stride, C2), wsc=ops.mfma_fragment_major(rn(C4, C2, k=C2 ** -0.5)), ssc=s3, bsc=b3, stride=stride)
else:
    kw.update(residual=rn(B, OH, OW, C4))
kw.update(w1=ops.mfma_fragment_major(rn(CN, C4, k=C4 ** -0.5)), s1=torch.ones(CN, device=dev), b1=torch.zeros(CN, device=dev))
run = lambda: ops.bottleneck_tail(b, w3, s3, b3, **kw)
nwg = B * OH * OW // 128
buf = torch.zeros(nwg * 8 * 24, dtype=torch.int64, device=dev)
for _ in range(3):
    run()
L.nps_rt8_debug_buffer(buf.data_ptr())
for _ in range(3):
    run()
torch.cuda.synchronize()
L.nps_rt8_debug_buffer(None)
t = buf.view(nwg, 8, 24).cpu().double()
names = ["prologue: address math + loads issued", "scale / shift + operand tiles -> LDS", "barrier"]
for c in range(4):
    names += ["chunk %d: GEMM 1 + epilogue -> y chunk (LDS)" % c, "   barrier (y chunk complete)", "   y store + GEMM 2", "   barrier (+ residual park)"]
names += ["a' epilogue + store"]
d = t[:, :, 1:len(names) + 1] - t[:, :, 0:len(names)]
print("%s: %d workgroups of 8 waves, one per CU; cycles per phase, mean | max" % (mode, nwg))
for i, n in enumerate(names):
    print("  %-46s %8.0f | %8.0f" % (n, float(d[:, :, i].mean()), float(d[:, :, i].max())))
tot = t[:, :, len(names)] - t[:, :, 0]
print("  %-46s %8.0f | %8.0f" % ("total", float(tot.mean()), float(tot.max())))
