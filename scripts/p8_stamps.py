"""GPU box: prologue / K-loop / epilogue cycle split of the p8 kernel from in-kernel cycle stamps (variant 24 = full kernel,
31 = no DMA + no fragment reads), per workgroup; also the spread of workgroup start times (rounds)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nopesac_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.load()
L.nps_p8_debug_buffer.argtypes = [ctypes.c_void_p]
L.nps_p8_debug_buffer.restype = None
L.nps_p8_debug_tile.argtypes = [ctypes.c_int]
L.nps_p8_debug_tile.restype = None
st = torch.cuda.current_stream().cuda_stream
shape = [int(v) for v in sys.argv[1:8]] if len(sys.argv) > 7 else [64, 60, 80, 256, 256, 3, 1]
B, H, W, Cin, Cout, k, s = shape
pad = k // 2
OH, OW = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
x = torch.randn(B, H, W, Cin, device=dev).bfloat16()
w = (torch.randn(Cout, k, k, Cin, device=dev) / (Cin * k * k) ** 0.5).bfloat16()
sc, bi = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
y = torch.empty(B, OH, OW, Cout, device=dev, dtype=torch.bfloat16)
nwg = ((B * OH * OW + 255) // 256) * (Cout // 256)
nk = k * k * Cin // 64
for variant, which in ((24, 1),):
    buf = torch.zeros(nwg * 8 * 16, dtype=torch.int64, device=dev)
    L.nps_p8_debug_buffer(buf.data_ptr())
    L.nps_p8_debug_tile(which)
    for _ in range(12):
        rc = L.nopesac_conv2d_nhwc_p8(x.data_ptr(), w.data_ptr(), sc.data_ptr(), bi.data_ptr(), None, y.data_ptr(), B, H, W, Cin, Cout, k, k, s, pad,
                                      Cin, Cout, 0, 1, 1, variant, st)  # act relu, bf16 out -> EPI 1
        assert rc == 0
    torch.cuda.synchronize()
    t = buf.view(nwg, 8, 16).cpu().double()
    live = t[:, 0, 0] > 0                       # persistent grid: only the first min(tiles, CUs) workgroup slots exist
    t = t[live]
    pro, loop, epi = (t[:, :, 1] - t[:, :, 0]), (t[:, :, 2] - t[:, :, 1]), (t[:, :, 3] - t[:, :, 2])
    print("variant %d: %d tiles on %d persistent workgroups, %d K-tiles; tile #%d of every workgroup: prologue %.0f  loop %.0f (= %.0f per K-tile, "
          "%.0f per interval)  epilogue incl. store drain %.0f cycles (mean over waves)" % (
              variant, nwg, int(live.sum()), nk, which, pro.mean(), loop.mean(), loop.mean() / nk, loop.mean() / nk / 8, epi.mean()))
    e = t[:, :, 4:16]
    ph = e[:, :, 0:9]                                  # K-tile 10: phase starts / barrier-1 crossings
    for grp, sl in (("leading waves 0-3", slice(0, 4)), ("lagging waves 4-7", slice(4, 8))):
        d = (ph[:, sl, 1:] - ph[:, sl, :-1]).mean(dim=(0, 1))
        print("   K-tile 10, %s: [load | mfma] intervals per phase: %s   (sum %.0f)" % (
            grp, "  ".join("%.0f | %.0f" % (float(d[2 * k]), float(d[2 * k + 1])) for k in range(4)), float(d.sum())))
    it = e[:, :, 9:12] - t[:, :, 2:3]
    print("   after the K loop's end (mean cycles): prefetch issued %.0f, row table + addresses %.0f, DMAs issued %.0f" % tuple(float(it[:, :, k].mean()) for k in range(3)))
    print("   per-wave loop cycles of workgroup 0:", [int(v) for v in loop[0].tolist()])
L.nps_p8_debug_buffer(None)
