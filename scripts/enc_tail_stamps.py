"""GPU box: where does a workgroup of enc_tail128_kernel spend its time?  In-kernel cycle stamps (tuning instantiation) at the phase
boundaries, per (workgroup, wave); prints the mean / max interval per phase in shader cycles and the spread of workgroup start times."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nopesac_amd import _lib, ops  # noqa: E402
dev = torch.device("cuda:0")
L = _lib.load()
L.nps_enc_tail_debug_buffer.argtypes = [ctypes.c_void_p]
L.nps_enc_tail_debug_buffer.restype = None
M = int(sys.argv[1]) if len(sys.argv) > 1 else 19200
g = torch.Generator(device=dev).manual_seed(0)
attn = torch.randn(M, 256, device=dev, generator=g).bfloat16()
src = torch.randn(M, 256, device=dev, generator=g)
fm = lambda n, k: ops.mfma_fragment_major((torch.randn(n, k, device=dev, generator=g) / k ** 0.5).bfloat16())
v = lambda n: 0.1 * torch.randn(n, device=dev, generator=g)
W = {"wo": fm(256, 256), "bo": v(256), "ga": 1 + v(256), "bea": v(256), "w1": fm(1024, 256), "b1": v(1024), "w2": fm(256, 1024), "b2": v(256),
     "gb": 1 + v(256), "beb": v(256)}
pos = torch.randn(300, 256, device=dev, generator=g)
pp, pj = (fm(512, 256), v(512), 512), (fm(256, 256), v(256), 256)
f = lambda: ops.transformer_tail(attn, src, W, pre_norm=False, pos=pos, want=("y",), proj_pos=pp, proj=pj)
rows = int(os.environ.get('NOPESAC_ENC_TAIL_ROWS', '3' if (M + 127) // 128 <= 160 and (M + 95) // 96 <= 256 else '4'))
nwg = (M + 32 * rows - 1) // (32 * rows)
buf = torch.zeros(nwg * 8 * 16, dtype=torch.int64, device=dev)
for _ in range(3):
    f()
L.nps_enc_tail_debug_buffer(buf.data_ptr())
for _ in range(5):
    f()
torch.cuda.synchronize()
L.nps_enc_tail_debug_buffer(None)
t = buf.view(nwg, 8, 16).cpu().double()
names = ["issue src/attn loads + ds_write", "barrier (attn visible)", "GEMM 1 (out-proj, 64 MFMA)", "LN 1", "bf16(y1) -> LDS + barrier", "quarter 0", "quarter 1",
         "quarter 2", "quarter 3", "LN 2", "f32 staging, chunk pass (y stores, pos), bf16 tiles -> LDS", "barrier", "projection rounds (192 MFMA) + staged stores", "store drain"]
d = t[:, :, 1:15] - t[:, :, 0:14]
print("M = %d, %d workgroups of %d tokens; cycles per phase, mean over (workgroup, wave) | max" % (M, nwg, 32 * rows))
for i, n in enumerate(names):
    print("  %-40s %8.0f | %8.0f" % (n, float(d[:, :, i].mean()), float(d[:, :, i].max())))
tot = t[:, :, 14] - t[:, :, 0]
print("  %-40s %8.0f | %8.0f   (MFMA floor: %d MFMAs x 32 cycles x 2 waves per SIMD = %d)" % ("total", float(tot.mean()), float(tot.max()), 192 * rows, 192 * rows * 64))
st = t[:, :, 0].min(dim=1).values
print("  workgroup start spread: %.0f cycles; end spread %.0f" % (float(st.max() - st.min()), float(t[:, :, 14].max(dim=1).values.max() - t[:, :, 14].max(dim=1).values.min())))
