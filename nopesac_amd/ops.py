"""Thin typed wrappers: torch tensors (device memory + the current HIP stream) -> C-ABI calls.

torch is plumbing here (allocation, streams); every computation below runs in libnopesac_hip.so.
All wrappers enqueue on torch's current stream and never synchronise.
"""
from __future__ import annotations

from typing import Optional

import ctypes
import os

import torch

from . import _lib

F32, BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_SIGMOID = 0, 1, 2, 3
ACT_RES_AFTER = 0x100      # OR-able: residual is added after the activation
ACT_BIAS_BATCHED = 0x200   # OR-able (set by conv2d): with batched weights, bias is [B, Cout]
FP8 = 3                    # NPS_DT_FP8: OCP e4m3fn bytes
_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.float8_e4m3fn: FP8}


class OpsArgumentError(AssertionError, ValueError):
    """A wrapper's argument check failed.  Raised explicitly (not `assert`): the checks stay in force under `python -O`.
    (Subclasses AssertionError so that callers written against the first version keep working.)"""


def _require(cond, msg="argument check failed"):
    if not cond:
        raise OpsArgumentError(msg if isinstance(msg, str) else repr(msg))


def _L():
    return _lib.load()


try:                                            # torch.cuda.current_stream() builds a Stream object through four layers of Python
    _raw_stream, _raw_device = torch._C._cuda_getCurrentRawStream, torch._C._cuda_getDevice   # (8 us per launch, a fifth of the submit time)
except AttributeError:                          # a torch build without the private accessors
    _raw_stream = _raw_device = None


def _stream():
    """hipStream_t (as an int) of torch's current stream on the current device."""
    if _raw_stream is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return None
    _require(t.is_cuda, "nopesac_amd ops need device tensors (there is no CPU path)")
    return t.data_ptr()


def _chk(t: torch.Tensor, dtype=None, contiguous=True):
    _require(t.is_cuda, "nopesac_amd ops need device tensors (there is no CPU path)")
    if dtype is not None:
        _require(t.dtype == dtype, (t.dtype, dtype))
    if contiguous:
        _require(t.is_contiguous(), 'argument check failed: t.is_contiguous()')
    return t


class ConvTuner:
    """Load-time autotuner: for every distinct conv/GEMM problem signature, time the kernel configurations the
    library offers (they all compute the same result) and remember the fastest.  Off by default; a model turns
    it on for one dedicated, single-stream pass (PlaneTR_NopeSAC.autotune) and then freezes it."""
    CANDIDATES = (0, 1, 2, 3, 4)

    def __init__(self):
        self.best = {}
        self.loaded = {}          # key_str -> cfg from a routing file (ConvTuner.load)
        self.measuring = False
        self.log = []

    def choose(self, key, launch, extra=()):
        cfg = self.best.get(key)
        if cfg is None and self.loaded:
            cfg = self.loaded.get(self.key_str(key))
            if cfg is not None:
                self.best[key] = cfg
        cands = tuple(self.CANDIDATES) + tuple(extra)
        if cfg is not None or not self.measuring:
            return cfg if (cfg is not None and cfg in cands) else 0     # a remembered cfg this call is not eligible for -> heuristic
        times = {}
        ok = []
        for c in cands:
            try:
                launch(c)                                 # warm (first-touch, icache)
                ok.append(c)
            except _lib.HipKernelError:
                # the C side rejects this shape for this configuration (an eligibility test above that does not mirror every
                # check of the entry point, or a stale routing entry): the candidate is dropped, tuning goes on.  Configuration 0
                # (the library's own heuristic) must work - its failure is the caller's error
                if c == 0:
                    raise
        cands = tuple(ok)
        for rnd in range(3):                              # three interleaved rounds, keep each candidate's best: robust to
            for c in cands:                               # a noisy neighbour / clock ramp during one candidate's window
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(6):
                    launch(c)
                e1.record()
                e1.synchronize()
                t = e0.elapsed_time(e1) / 6
                times[c] = min(times.get(c, t), t)
        cfg = min(times, key=times.get)
        if times[cfg] > 0.97 * times[0]:      # keep the heuristic unless a candidate is clearly faster
            cfg = 0
        self.best[key] = cfg
        self.log.append((key, cfg, times))
        return cfg

    # ---- persistent routing: the decisions of one tuning pass as a JSON file, so that the benchmark, the PMC / rocprofv3 scripts
    # and the driver's runs all launch IDENTICAL kernels (profiles/routing_*.json; `bench.py --routing`)
    @staticmethod
    def key_str(key) -> str:
        return "|".join(str(v).replace("torch.", "") for v in key)

    def save(self, path: str, meta: dict = None):
        import json
        rows = {self.key_str(k): int(v) for k, v in self.best.items()}
        with open(path, "w") as f:
            json.dump({"format": "nopesac_amd.ConvTuner/1", "meta": meta or {}, "kernels": {str(k): v for k, v in CONV_CFG_KERNEL.items()},
                       "routing": dict(sorted(rows.items()))}, f, indent=1)

    def load(self, path: str) -> int:
        """Install a saved routing: shapes it lists are never re-measured (shapes it does not list fall back to the built-in
        heuristic unless `measuring` is switched on again).  Returns the number of entries."""
        import json
        with open(path) as f:
            doc = json.load(f)
        _require(doc.get("format") == "nopesac_amd.ConvTuner/1", "not a ConvTuner routing file")
        self.loaded = {k: int(v) for k, v in doc["routing"].items()}
        return len(self.loaded)


TUNER = ConvTuner()
CFG_BFRAG3, CFG_BFRAG32 = 7, 8         # tuner-only configurations: nopesac_conv2d_nhwc_bfrag, K-tile 64 / 32
CFG_HALO16, CFG_HALO8 = 9, 10          # tuner-only: nopesac_conv3x3_halo_bf16, 16x16 / 16x8 pixel tiles
CFG_P8 = 11                            # tuner-only: nopesac_conv2d_nhwc_p8 (256x256x64 tiles, phase-interleaved 8-wave schedule)
CFG_P8N, CFG_P8N_TAP = 13, 14          # tuner-only: nopesac_conv2d_nhwc_p8n (256x128 tiles: the Cout % 128 == 0 layers, round 5), channel- / tap-major K order
CFG_P8N_SPLIT = 15                     # tuner-only: nopesac_conv2d_nhwc_p8n_splitk (few tiles x long K: the pose net's first conv; round 6)
CFG_P8_SK = 12                         # tuner-only: nopesac_conv2d_nhwc_p8_sk (the same kernel with stream-K work distribution, round 5)
# NOPESAC_P8_CAP_1X1=n (experiment, round 5): persistent workgroups of the p8 kernel on 1x1 layers (the HBM-bound ones) capped at n
P8_CAP_1X1 = [int(os.environ.get("NOPESAC_P8_CAP_1X1", "0"))]
P8_VARIANT = [32]                      # variant handed to nopesac_conv2d_nhwc_p8: 32 = channel-major K order (better L2 reuse of the taps)
# added to nopesac_conv2d_nhwc_bfrag's variant for stride-1 KxK convs: channel-major K order (round 4: same time in isolation, 95 instead
# of 149 MB read from HBM per launch on res3's 3x3 layers; stride 2 measured slower and stays tap-major).  NOPESAC_BFRAG_KMAJOR=0: A/B runs
BFRAG_KMAJOR = [0 if os.environ.get("NOPESAC_BFRAG_KMAJOR") == "0" else 256]
LAST_CONV_CFG = [0]                    # kernel configuration of the most recent conv2d launch (0 = the library's heuristic)
CONV_CFG_KERNEL = {1: "conv_igemm_kernel<128x128>", 2: "conv_igemm_kernel<64x64>", 3: "conv_igemm_glds_kernel<BK=64>", 4: "conv_igemm_glds_kernel<BK=32>",
                   7: "conv_igemm_bfrag_kernel<3, 64, false>", 8: "conv_igemm_bfrag_kernel<4, 32, false>", 9: "conv3x3_halo_kernel<16, 16>",
                   10: "conv3x3_halo_kernel<16, 8>", 11: "conv_igemm_p8_kernel", 12: "conv_igemm_p8_kernel<stream-K>",
                   13: "conv_igemm_p8n_kernel", 14: "conv_igemm_p8n_kernel<tap-major>", 15: "conv_igemm_p8n_kernel<split-K>"}
P8N_TUNABLE = [os.environ.get("NOPESAC_P8N", "1") != "0"]           # NOPESAC_P8N=0: the tuner never offers the 256x128-tile kernel (A/B runs)
P8_SK_TUNABLE = [os.environ.get("NOPESAC_P8_SK", "0") == "1"]      # NOPESAC_P8_SK=1: the tuner may pick the stream-K form (wins isolated launches, loses 1.2 % in the four-in-flight loop: profiles/r5_b_*)
_P8_SK_WS = {}                         # (device index, stream handle) -> workspace tensor of the stream-K conv


def p8_sk_workspace(device) -> torch.Tensor:
    """The stream-K conv's workspace for the CURRENT stream of `device` (arrival counters + partial-tile slabs, 128 MB): launches of one
    stream are ordered, so they share one; every stream gets its own.  Zeroed once - the kernel leaves the counters zero.  Inside a
    graph capture the allocation (and its zero fill, harmlessly replayed) belongs to the capture's private pool."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), _stream(), torch.cuda.is_current_stream_capturing())
    ws = _P8_SK_WS.get(key)
    if ws is None:
        n = int(_L().nopesac_conv2d_p8_sk_workspace_bytes())
        ws = torch.empty(n, device=device, dtype=torch.uint8)
        ws[:16384].zero_()
        _P8_SK_WS[key] = ws
    return ws


def _frag_weights(w: torch.Tensor) -> torch.Tensor:
    """Fragment-major copy of a conv weight [Cout,KH,KW,Cin], made once and kept ON the weight tensor object (packed weights
    are static and long-lived; an address-keyed cache would hand out stale fragments when a freed weight's memory is reused).
    `linear()` passes a fresh 4-D view of the packed matrix on every call: the copy then lives on the view's base tensor."""
    holder = w
    base = w._base
    if base is not None and base.numel() == w.numel() and base.storage_offset() == w.storage_offset() and base.is_contiguous():
        holder = base
    f = getattr(holder, "_nps_frag", None)
    if f is None:
        f = mfma_fragment_major(w.reshape(w.shape[0], -1))
        holder._nps_frag = f
    return f


def conv2d(x: torch.Tensor, w: torch.Tensor, scale=None, bias=None, residual=None, *, stride=1, pad=0, act=ACT_NONE,
           out: Optional[torch.Tensor] = None, out_dtype=None, x_channels: Optional[int] = None,
           batched_weights: bool = False) -> torch.Tensor:
    """NHWC conv / linear.  x: [B,H,W,Cx] (a channel slice view of a wider buffer is allowed: only the
    last-dim stride may exceed the channel count); w: [Cout,KH,KW,Cin] or [B,Cout,KH,KW,Cin] when
    `batched_weights`.  `out` may be a channel-slice view of a wider NHWC buffer."""
    _require(x.dim() == 4 and x.stride(3) == 1, 'argument check failed: x.dim() == 4 and x.stride(3) == 1')
    B, H, W, Cx = x.shape
    x_cs = x.stride(2)
    _require(x.stride(1) == W * x_cs and x.stride(0) == H * W * x_cs, "x must be pixel-dense NHWC")
    if batched_weights:
        _require(w.dim() == 5 and w.shape[0] == B and w.is_contiguous(), 'argument check failed: w.dim() == 5 and w.shape[0] == B and w.is_contiguous()')
        Cout, KH, KW, Cin = w.shape[1:]
        w_bs = Cout * KH * KW * Cin
    else:
        _require(w.dim() == 4 and w.is_contiguous(), 'argument check failed: w.dim() == 4 and w.is_contiguous()')
        Cout, KH, KW, Cin = w.shape
        w_bs = 0
    _require(Cin == (x_channels or Cx), (Cin, Cx))
    mixed = x.dtype == torch.float32 and w.dtype == torch.bfloat16     # f32 activations x bf16 weights
    _require(w.dtype == x.dtype or mixed, 'argument check failed: w.dtype == x.dtype or mixed')
    OH = (H + 2 * pad - KH) // stride + 1
    OW = (W + 2 * pad - KW) // stride + 1
    out_dtype = out_dtype or (out.dtype if out is not None else x.dtype)
    if out is None:
        out = torch.empty((B, OH, OW, Cout), device=x.device, dtype=out_dtype)
    _require(out.shape == (B, OH, OW, Cout) and out.stride(3) == 1 and out.dtype == out_dtype, 'argument check failed: out.shape == (B, OH, OW, Cout) and out.stride(3) == 1 and out.dtype == out_dtype')
    y_cs = out.stride(2)
    _require(out.stride(1) == OW * y_cs and out.stride(0) == OH * OW * y_cs, 'argument check failed: out.stride(1) == OW * y_cs and out.stride(0) == OH * OW * y_cs')
    r_cs = 0
    if residual is not None:
        _require(residual.shape == out.shape and residual.dtype == out_dtype and residual.stride(3) == 1, 'argument check failed: residual.shape == out.shape and residual.dtype == out_dtype and residual.stride(3) == 1')
        r_cs = residual.stride(2)
    if batched_weights and bias is not None and bias.numel() == B * Cout and B > 1:
        act |= ACT_BIAS_BATCHED                                   # per-batch bias rows [B, Cout]
    for v in (scale, bias):
        if v is not None:
            _chk(v, torch.float32)
            _require(v.numel() == Cout or (v is bias and (act & ACT_BIAS_BATCHED)), 'argument check failed: v.numel() == Cout or (v is bias and (act & ACT_BIAS_BATCHED))')
    # "A through LDS, B from L2" kernel (fragment-major weights, cached per weight tensor): only the autotuner selects it
    bfrag_ok = (x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and not batched_weights and Cin % 64 == 0 and Cout % 128 == 0
                and x_cs % 8 == 0 and KH * KW <= 32 and (act & ~(0xff | ACT_RES_AFTER)) == 0
                and (B * H * W + pad * W + pad) * x_cs * 2 < 2 ** 31)

    halo_ok = (bfrag_ok and KH == 3 and KW == 3 and stride == 1 and pad == 1 and residual is None and x_cs == Cin and y_cs == Cout
               and out_dtype == torch.bfloat16 and scale is not None and bias is not None and (act & ~0xff) == 0)

    p8_ok = (x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and not batched_weights and Cin % 64 == 0 and Cout % 256 == 0
             and x_cs % 8 == 0 and KH * KW <= 32 and (act & ~(0xff | ACT_RES_AFTER)) == 0
             and (B * H * W + pad * W + pad) * x_cs * 2 < 2 ** 31 and B * H * W < 2 ** 23 and x_cs < 2 ** 24 and KH * KW * Cin < 2 ** 24
             and Cout * KH * KW * Cin * 2 < 2 ** 31 and out_dtype in _DT and x_cs % 8 == 0
             and (y_cs % (4 if out_dtype == torch.float32 else 8) == 0)
             and (residual is None or (r_cs % (4 if out_dtype == torch.float32 else 8) == 0 and out_dtype != torch.float8_e4m3fn))
             and all(t is None or t.data_ptr() % 16 == 0 for t in (x, w, out, residual, scale, bias)))

    # stream-K only where whole rounds leave CUs idle: a few tiles per CU and a K loop long enough to cut
    p8_sk_ok = p8_ok and (-(-(B * OH * OW) // 256)) * (Cout // 256) <= 1024 and KH * KW * Cin >= 512

    p8n_ok = (x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and out_dtype == torch.bfloat16 and not batched_weights and residual is None
              and Cin % 64 == 0 and Cout % 128 == 0 and x_cs % 8 == 0 and y_cs % 8 == 0 and KH * KW <= 32 and act in (ACT_NONE, ACT_RELU, ACT_LEAKY)
              and (B * H * W + pad * W + pad) * x_cs * 2 < 2 ** 31 and B * H * W < 2 ** 23 and x_cs < 2 ** 24 and KH * KW * Cin < 2 ** 24
              and Cout * KH * KW * Cin * 2 < 2 ** 31 and (B * OH * OW + 256) * y_cs * 2 < 2 ** 31
              and all(t is None or t.data_ptr() % 16 == 0 for t in (x, w, out, scale, bias)))

    # split-K on the p8n structure: fewer tiles than half the CUs and a K loop of >= 64 K-tiles; slices = CUs // tiles (at most 8)
    p8n_tiles = (-(-(B * OH * OW) // 256)) * (Cout // 128) if Cout % 128 == 0 else 0
    p8n_splits = min(8, Cin // 64, 256 // p8n_tiles) if p8n_tiles else 0
    p8n_split_ok = p8n_ok and KH * KW * Cin >= 4096 and p8n_splits >= 2 and p8n_splits * B * OH * OW * Cout * 4 < 2 ** 31

    def launch(cfg):
        if cfg == CFG_P8N_SPLIT:
            ws = torch.empty(p8n_splits * B * OH * OW * Cout, device=x.device, dtype=torch.float32)    # (caching allocator: capture-safe)
            rc = _L().nopesac_conv2d_nhwc_p8n_splitk(_p(x), _p(w), _p(scale), _p(bias), _p(out), B, H, W, Cin, Cout, KH, KW, stride, pad, x_cs,
                                                     y_cs, act, 32, p8n_splits, _p(ws), ws.numel() * 4, _stream())
            _lib.check(rc, "nopesac_conv2d_nhwc_p8n_splitk")
            return
        if cfg in (CFG_P8N, CFG_P8N_TAP):
            rc = _L().nopesac_conv2d_nhwc_p8n(_p(x), _p(w), _p(scale), _p(bias), _p(out), B, H, W, Cin, Cout, KH, KW, stride, pad, x_cs, y_cs,
                                              act, 32 if cfg == CFG_P8N else 0, _stream())
            _lib.check(rc, "nopesac_conv2d_nhwc_p8n")
            return
        if cfg == CFG_P8_SK:
            ws = p8_sk_workspace(x.device)
            rc = _L().nopesac_conv2d_nhwc_p8_sk(_p(x), _p(w), _p(scale), _p(bias), _p(residual), _p(out), B, H, W, Cin, Cout, KH, KW, stride,
                                                pad, x_cs, y_cs, r_cs, act, _DT[out_dtype], P8_VARIANT[0], _p(ws), ws.numel(), _stream())
            _lib.check(rc, "nopesac_conv2d_nhwc_p8_sk")
            return
        if cfg == CFG_P8:
            rc = _L().nopesac_conv2d_nhwc_p8(_p(x), _p(w), _p(scale), _p(bias), _p(residual), _p(out), B, H, W, Cin, Cout, KH, KW, stride,
                                             pad, x_cs, y_cs, r_cs, act, _DT[out_dtype],
                                             P8_VARIANT[0] | ((P8_CAP_1X1[0] << 8) if KH * KW == 1 else 0), _stream())
            _lib.check(rc, "nopesac_conv2d_nhwc_p8")
            return
        if cfg in (CFG_HALO16, CFG_HALO8):
            rc = _L().nopesac_conv3x3_halo_bf16(_p(x), _p(_frag_weights(w)), _p(scale), _p(bias), _p(out), B, H, W, Cin, Cout, act,
                                                0 if cfg == CFG_HALO16 else 1, _stream())
            _lib.check(rc, "nopesac_conv3x3_halo_bf16")
            return
        if cfg in (CFG_BFRAG3, CFG_BFRAG32):
            rc = _L().nopesac_conv2d_nhwc_bfrag(_p(x), _p(_frag_weights(w)), _p(scale), _p(bias), _p(residual), _p(out), B, H, W, Cin, Cout,
                                                KH, KW, stride, pad, x_cs, y_cs, r_cs, act, _DT[out_dtype],
                                                (3 if cfg == CFG_BFRAG3 else 32) + (BFRAG_KMAJOR[0] if (KH * KW > 1 and stride == 1) else 0), _stream())
            _lib.check(rc, "nopesac_conv2d_nhwc_bfrag")
            return
        rc = _L().nopesac_conv2d_nhwc_ex(_p(x), _p(w), _p(scale), _p(bias), _p(residual), _p(out), B, H, W, Cin, Cout, KH, KW,
                                         stride, pad, x_cs, y_cs, r_cs, w_bs, act, 2 if mixed else _DT[x.dtype], _DT[out_dtype],
                                         cfg, _stream())
        _lib.check(rc, "nopesac_conv2d_nhwc_ex")

    cfg = 0
    if TUNER.measuring or TUNER.best or TUNER.loaded:
        # everything that decides which kernel configurations are eligible (bfrag_ok / halo_ok) is part of the key
        key = (x.dtype, w.dtype, out_dtype, B, H, W, Cin, Cout, KH, KW, stride, pad, residual is not None, x_cs, y_cs, w_bs != 0,
               scale is not None, bias is not None, act, bfrag_ok, halo_ok, p8_ok)
        cfg = TUNER.choose(key, launch, ((CFG_BFRAG3, CFG_BFRAG32) if bfrag_ok else ()) + ((CFG_HALO16, CFG_HALO8) if halo_ok else ())
                           + ((CFG_P8,) if p8_ok else ()) + ((CFG_P8_SK,) if (p8_sk_ok and P8_SK_TUNABLE[0]) else ())
                           + (((CFG_P8N,) + ((CFG_P8N_TAP,) if KH * KW > 1 else ())) if (p8n_ok and P8N_TUNABLE[0]) else ())
                           + ((CFG_P8N_SPLIT,) if (p8n_split_ok and P8N_TUNABLE[0]) else ()))
    try:
        launch(cfg)
    except _lib.HipKernelError:
        if cfg == 0:
            raise
        TUNER.best[key] = cfg = 0      # a stale routing entry the entry point rejects for this shape: heuristic from now on
        launch(0)
    LAST_CONV_CFG[0] = cfg             # read by bench.py's per-launch timer to attribute the launch to a kernel
    return out


def _rows4d(t: torch.Tensor, width: int) -> torch.Tensor:
    """View `t` ([..., width]: contiguous, or a column slice of a contiguous wider buffer) as a
    [1,1,rows,width] NHWC tensor whose pixel stride is the buffer's row stride."""
    _require(t.stride(-1) == 1 and t.shape[-1] == width, 'argument check failed: t.stride(-1) == 1 and t.shape[-1] == width')
    rows = t.numel() // width
    rs = width if t.is_contiguous() else t.stride(-2)
    if t.dim() > 2 and not t.is_contiguous():
        for d in range(t.dim() - 2):     # leading dims must be dense multiples of the row stride
            _require(t.stride(d) == t.stride(d + 1) * t.shape[d + 1], "unsupported row layout")
    return torch.as_strided(t, (1, 1, rows, width), (rows * rs, rows * rs, rs, 1), t.storage_offset())


def linear(x: torch.Tensor, w: torch.Tensor, bias=None, *, act=ACT_NONE, residual=None, out=None, scale=None,
           out_dtype=None) -> torch.Tensor:
    """y = act((x @ w.T) * scale + bias + residual).  x [..., K] / out [..., N] / residual [..., N] may be
    column slices of wider row-major buffers (free concatenation); w [N, K] contiguous."""
    K = x.shape[-1]
    N = w.shape[0]
    y = conv2d(_rows4d(x, K), w.view(N, 1, 1, K), scale, bias, None if residual is None else _rows4d(residual, N), act=act,
               out=None if out is None else _rows4d(out, N), out_dtype=out_dtype)
    return out if out is not None else y.view(*x.shape[:-1], N)


def preprocess(images_nchw: torch.Tensor, mean: torch.Tensor, std: torch.Tensor, cpad: int, out_dtype) -> torch.Tensor:
    x = _chk(images_nchw, torch.float32)
    B, C, H, W = x.shape
    y = torch.empty((B, H, W, cpad), device=x.device, dtype=out_dtype)
    rc = _L().nopesac_preprocess_nchw_to_nhwc(_p(x), _p(y), _p(_chk(mean, torch.float32)), _p(_chk(std, torch.float32)), B, C, H, W,
                                              cpad, _DT[out_dtype], _stream())
    _lib.check(rc, "nopesac_preprocess_nchw_to_nhwc")
    return y


def stem_fused(x: torch.Tensor, w224: torch.Tensor, scale: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """bf16 conv7x7/s2 + BN + ReLU + maxpool3x3/s2 in one kernel.  x [B,H,W,4] bf16, w224 [64,224] bf16."""
    _chk(x, torch.bfloat16); _chk(w224, torch.bfloat16); _chk(scale, torch.float32); _chk(bias, torch.float32)
    B, H, W, C = x.shape
    _require(C == 4 and w224.shape == (64, 224), 'argument check failed: C == 4 and w224.shape == (64, 224)')
    CH, CW = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    PH, PW = (CH + 2 - 3) // 2 + 1, (CW + 2 - 3) // 2 + 1
    y = torch.empty((B, PH, PW, 64), device=x.device, dtype=torch.bfloat16)
    _lib.check(_L().nopesac_stem_fused_bf16(_p(x), _p(w224), _p(scale), _p(bias), _p(y), B, H, W, _stream()), "nopesac_stem_fused_bf16")
    return y


def stem_fused_raw(images: torch.Tensor, mean: torch.Tensor, std: torch.Tensor, w224: torch.Tensor, scale: torch.Tensor,
                   bias: torch.Tensor) -> torch.Tensor:
    """`preprocess` + `stem_fused` in one kernel: images f32 NCHW [B,3,H,W] (0-255), per-channel mean / std f32[3]."""
    _chk(images, torch.float32); _chk(mean, torch.float32); _chk(std, torch.float32)
    _chk(w224, torch.bfloat16); _chk(scale, torch.float32); _chk(bias, torch.float32)
    B, C, H, W = images.shape
    _require(C == 3 and mean.numel() == 3 and std.numel() == 3 and w224.shape == (64, 224), 'argument check failed: C == 3 and mean.numel() == 3 and std.numel() == 3 and w224.shape == (64, 224)')
    CH, CW = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    PH, PW = (CH + 2 - 3) // 2 + 1, (CW + 2 - 3) // 2 + 1
    y = torch.empty((B, PH, PW, 64), device=images.device, dtype=torch.bfloat16)
    _lib.check(_L().nopesac_stem_fused_raw_bf16(_p(images), _p(mean), _p(std), _p(w224), _p(scale), _p(bias), _p(y), B, H, W, _stream()),
               "nopesac_stem_fused_raw_bf16")
    return y


def stem_fused_raw_shifted(images: torch.Tensor, pad3: torch.Tensor, w224_folded: torch.Tensor, scale: torch.Tensor,
                           bias_folded: torch.Tensor) -> torch.Tensor:
    """The raw-image stem with the normalisation folded into weights / shift (see `fold_stem_normalisation`): the patch holds the
    pixel values minus 128, exact in bf16."""
    _chk(images, torch.float32); _chk(pad3, torch.float32); _chk(w224_folded, torch.bfloat16); _chk(scale, torch.float32); _chk(bias_folded, torch.float32)
    B, C, H, W = images.shape
    _require(C == 3 and pad3.numel() == 3 and w224_folded.shape == (64, 224), 'argument check failed: C == 3 and pad3.numel() == 3 and w224_folded.shape == (64, 224)')
    CH, CW = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    PH, PW = (CH + 2 - 3) // 2 + 1, (CW + 2 - 3) // 2 + 1
    y = torch.empty((B, PH, PW, 64), device=images.device, dtype=torch.bfloat16)
    _lib.check(_L().nopesac_stem_fused_raw_shifted_bf16(_p(images), _p(pad3), _p(w224_folded), _p(scale), _p(bias_folded), _p(y), B, H, W, _stream()),
               "nopesac_stem_fused_raw_shifted_bf16")
    return y


def fold_stem_normalisation(w_o773: torch.Tensor, scale: torch.Tensor, bias: torch.Tensor, mean: torch.Tensor, std: torch.Tensor):
    """Operands of `stem_fused_raw_shifted` from the stem's f32 weights [64,7,7,3] (o, kh, kw, c), its folded-BN scale / shift and the
    per-channel pixel mean / std:  sum_k w (v - mean) / std = sum_k (w / std) (v - 128) + sum_k (w / std) (128 - mean).
    -> (pad3 f32[3] = mean - 128, w224 bf16 [64,224] in the fused stem's (kh, kw padded to 8, c padded to 4) order, bias' f32[64])."""
    wd, m, s = w_o773.double(), mean.double().view(1, 1, 1, 3), std.double().view(1, 1, 1, 3)
    wf = wd / s
    const = (wf * (128.0 - m)).sum(dim=(1, 2, 3))
    w8 = w_o773.new_zeros(64, 7, 8, 4, dtype=torch.float32)
    w8[:, :, :7, :3] = wf.float()
    return ((mean.float() - 128.0).contiguous(), w8.reshape(64, 224).to(torch.bfloat16).contiguous(),
            (bias.double() + scale.double() * const).float().contiguous())


BOTTLENECK_TAIL_CONFIGS = {(64, 256, 0, 0), (64, 256, 64, 0), (64, 256, 128, 0), (64, 256, 0, 64), (64, 256, 64, 64),          # (C, C4, CN, C2)
                           (128, 512, 0, 0), (128, 512, 128, 0), (128, 512, 256, 0), (128, 512, 0, 256), (128, 512, 128, 256),
                           (256, 1024, 0, 0), (256, 1024, 256, 0), (256, 1024, 512, 0), (256, 1024, 0, 512), (256, 1024, 256, 512)}


def mfma_fragment_major(w2d: torch.Tensor) -> torch.Tensor:
    """[N,K] (N % 32 == 0, K % 16 == 0) -> same shape, re-ordered [N/32][K/16][2][32][8]: the order in which a wave's 64 lanes
    consume the matrix as v_mfma_f32_32x32x16_bf16 operands (lane = 32*(k%16 >= 8) + n%32 holds 8 consecutive k)."""
    N, K = w2d.shape
    _require(N % 32 == 0 and K % 16 == 0, 'argument check failed: N % 32 == 0 and K % 16 == 0')
    return w2d.view(N // 32, 32, K // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous().view(N, K)


def mlp_padded_k(N: int, K: int) -> int:
    """K as the chained-MLP kernel pads it (csrc/mlp_chain.hip: two K blocks of the layer's width class)."""
    q = 128 * (4 if N <= 256 else 2 if N <= 512 else 1)        # an even number of blocks of 64 * ksplit channels (ksplit = 4 / tiles per wave)
    return -(-K // q) * q


class MlpLayer:
    """One Linear of a chained MLP stack, packed for nopesac_mlp_chain_bf16: bf16 weights [Np, Kp] (zero padded, MFMA
    fragment-major), f32 bias [Np]."""
    __slots__ = ("w", "bias", "N", "K")

    def __init__(self, w2d: torch.Tensor, bias: Optional[torch.Tensor]):
        N, K = w2d.shape
        _require(N <= _lib.MLP_MAX_WIDTH, f"mlp_pack: N = {N} exceeds NOPESAC_MLP_MAX_WIDTH")
        Np, Kp = -(-N // 32) * 32, mlp_padded_k(N, K)
        wp = torch.zeros(Np, Kp, device=w2d.device, dtype=torch.bfloat16)
        wp[:N, :K] = w2d.to(torch.bfloat16)
        self.w = mfma_fragment_major(wp)
        self.bias = None
        if bias is not None:
            self.bias = torch.zeros(Np, device=w2d.device, dtype=torch.float32)
            self.bias[:N] = bias.float()
        self.N, self.K = N, K


def mlp_chain(x: torch.Tensor, layers, acts, outs, x_bcast: Optional[torch.Tensor] = None, rows_per: int = 1, restarts=None):
    """A stack of Linear(+bias)(+act) layers over the rows of x in ONE launch (bf16 MFMA operands, f32 activations: the rounding
    points of one ops.linear per layer).  x [rows, Kx] f32 (may be a column slice of a wider row-major buffer); x_bcast [P, Kb]:
    optional prefix columns shared by `rows_per` consecutive rows (row r reads x_bcast[r // rows_per]); layers: [MlpLayer];
    acts: [ACT_*] per layer; outs: per layer None or an f32 [rows, N] tensor (may be a column slice) that receives the layer's
    output - the last one is required.  restarts: per layer True if the layer starts a new stack over the SAME input rows."""
    _require(len(layers) == len(acts) == len(outs) and 1 <= len(layers) <= _lib.MLP_MAX_LAYERS, "mlp_chain: layers / acts / outs")
    _require(x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1, "mlp_chain: x must be f32 [rows, K] with unit column stride")
    rows = x.shape[0]
    c = _lib.MlpChain()
    c.x, c.x_ld, c.x_width = _p(x), x.stride(0), x.shape[1]
    c.xb, c.xb_ld, c.xb_width, c.xb_rows_per = None, 0, 0, 1
    if x_bcast is not None:
        _require(x_bcast.dtype == torch.float32 and x_bcast.dim() == 2 and x_bcast.stride(1) == 1 and rows_per >= 1
                 and x_bcast.shape[0] * rows_per >= rows, "mlp_chain: x_bcast")
        c.xb, c.xb_ld, c.xb_width, c.xb_rows_per = _p(x_bcast), x_bcast.stride(0), x_bcast.shape[1], rows_per
    c.rows, c.n_layers = rows, len(layers)
    for i, (l, a, o) in enumerate(zip(layers, acts, outs)):
        e = c.layers[i]
        e.w, e.bias, e.K, e.N, e.act = _p(l.w), _p(l.bias), l.K, l.N, a
        e.reserved = 1 if (restarts is not None and restarts[i]) else 0        # NOPESAC_MLP_RESTART: reads the chain input again
        e.out, e.out_ld = None, 0
        if o is not None:
            _require(o.dtype == torch.float32 and o.dim() == 2 and o.shape == (rows, l.N) and o.stride(1) == 1, "mlp_chain: out tensor")
            e.out, e.out_ld = _p(o), o.stride(0)
    _lib.check(_L().nopesac_mlp_chain_bf16(ctypes.byref(c), _stream()), "nopesac_mlp_chain_bf16")
    return outs[-1]


def mfma_fragment_major_fp8(w2d: torch.Tensor) -> torch.Tensor:
    """[N,K] one-byte elements (N % 32 == 0, K % 64 == 0) -> same shape, re-ordered [N/32][K/64][2][64][16]: the two 16-byte
    pieces h = 0, 1 that lane l = 32*half + n%32 feeds to v_mfma_f32_32x32x64_f8f6f4 hold k = kf*64 + 32*half + 16*h + 0..15."""
    N, K = w2d.shape
    _require(N % 32 == 0 and K % 64 == 0 and w2d.element_size() == 1, 'argument check failed: N % 32 == 0 and K % 64 == 0 and w2d.element_size() == 1')
    return w2d.view(N // 32, 32, K // 64, 2, 2, 16).permute(0, 2, 4, 3, 1, 5).contiguous().view(N, K)


FP8_MAX = 448.0            # largest finite e4m3fn


def quantize_weights_fp8(w: torch.Tensor):
    """Conv weight [Cout,KH,KW,Cin] (any float dtype) -> (fp8 fragment-major matrix [Cout, KH*KW*Cin], per-output-channel
    scale f32[Cout]) with w ~= w8 * scale[:, None]; symmetric, amax -> 448 per output channel."""
    w2 = w.reshape(w.shape[0], -1).float()
    sc = w2.abs().amax(dim=1).clamp_min(1e-12) / FP8_MAX
    w8 = (w2 / sc[:, None]).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn)
    return mfma_fragment_major_fp8(w8), sc.contiguous()


def conv2d_fp8(x: torch.Tensor, w8frag: torch.Tensor, scale: torch.Tensor, bias: torch.Tensor, *, ksize: int, stride=1, pad=0,
               act=ACT_NONE, out_dtype=torch.bfloat16, residual=None, variant: int = 0) -> torch.Tensor:
    """fp8 (e4m3fn) NHWC conv on the K = 64 fp8 MFMA.  x [B,H,W,Cin] float8_e4m3fn, w8frag from `quantize_weights_fp8`
    ([Cout, k*k*Cin]); `scale` carries the de-quantisation (bn_scale * w_scale * x_scale).  variant 0 = by Cin."""
    _require(x.dtype == torch.float8_e4m3fn and w8frag.dtype == torch.float8_e4m3fn and x.is_contiguous() and w8frag.is_contiguous(), 'argument check failed: x.dtype == torch.float8_e4m3fn and w8frag.dtype == torch.float8_e4m3fn and x.is_contiguous() and w8frag.is_contiguous()')
    B, H, W, Cin = x.shape
    Cout = w8frag.shape[0]
    _require(w8frag.shape[1] == ksize * ksize * Cin, 'argument check failed: w8frag.shape[1] == ksize * ksize * Cin')
    _chk(scale, torch.float32); _chk(bias, torch.float32)
    OH = (H + 2 * pad - ksize) // stride + 1
    OW = (W + 2 * pad - ksize) // stride + 1
    out = torch.empty((B, OH, OW, Cout), device=x.device, dtype=out_dtype)
    if residual is not None:
        _chk(residual, out_dtype)
        _require(residual.shape == out.shape, 'argument check failed: residual.shape == out.shape')
    if not variant:
        variant = 3 if Cin % 128 == 0 else 32
    rc = _L().nopesac_conv2d_nhwc_fp8(_p(x), _p(w8frag), _p(scale), _p(bias), _p(residual), _p(out), B, H, W, Cin, Cout, ksize, ksize,
                                      stride, pad, Cin, Cout, Cout if residual is not None else 0, act, _DT[out_dtype], variant, _stream())
    _lib.check(rc, "nopesac_conv2d_nhwc_fp8")
    return out


def bottleneck_tail(b, w3, s3, b3, *, residual=None, x2=None, wsc=None, ssc=None, bsc=None, stride=1, w1=None, s1=None, b1=None,
                    o_fp8: bool = False):
    """Fused conv3 + shortcut + ReLU (+ the next block's conv1) of a bf16 bottleneck; returns (y, o or None).
    b [B,OH,OW,C]; residual [B,OH,OW,C4] or projection source x2 [B,H2,W2,C2] with wsc [C4,C2]; w1 [CN,C4].
    The weight matrices are expected in `mfma_fragment_major` order.  o_fp8: o is written as float8_e4m3fn."""
    _chk(b, torch.bfloat16); _chk(w3, torch.bfloat16)
    B, OH, OW, C = b.shape
    C4 = w3.shape[0]
    CN = 0 if w1 is None else w1.shape[0]
    C2 = 0 if x2 is None else x2.shape[3]
    _require((C, C4, CN, C2) in BOTTLENECK_TAIL_CONFIGS, (C, C4, CN, C2))
    y = torch.empty((B, OH, OW, C4), device=b.device, dtype=torch.bfloat16)
    o = torch.empty((B, OH, OW, CN), device=b.device, dtype=torch.float8_e4m3fn if o_fp8 else torch.bfloat16) if CN else None
    if residual is not None:
        _chk(residual, torch.bfloat16)
        _require(residual.shape == y.shape, 'argument check failed: residual.shape == y.shape')
    if x2 is not None:
        _chk(x2, torch.bfloat16); _chk(wsc, torch.bfloat16)
    H2, W2 = (x2.shape[1], x2.shape[2]) if x2 is not None else (0, 0)
    rc = _L().nopesac_bottleneck_tail_bf16_ex(_p(b), _p(w3), _p(s3), _p(b3), _p(residual), _p(x2), _p(wsc), _p(ssc), _p(bsc), B, OH, OW,
                                              H2, W2, stride, C, C4, C2, _p(y), _p(w1), _p(s1), _p(b1), CN, _p(o),
                                              FP8 if o_fp8 else BF16, _stream())
    _lib.check(rc, "nopesac_bottleneck_tail_bf16_ex")
    return y, o


def maxpool(x: torch.Tensor, k: int, stride: int, pad: int) -> torch.Tensor:
    _chk(x)
    B, H, W, C = x.shape
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    y = torch.empty((B, OH, OW, C), device=x.device, dtype=x.dtype)
    _lib.check(_L().nopesac_maxpool_nhwc(_p(x), _p(y), B, H, W, C, k, stride, pad, _DT[x.dtype], _stream()), "nopesac_maxpool_nhwc")
    return y


def upsample2x_bilinear(x: torch.Tensor, addend=None, act=ACT_NONE) -> torch.Tensor:
    _chk(x)
    B, H, W, C = x.shape
    y = torch.empty((B, 2 * H, 2 * W, C), device=x.device, dtype=x.dtype)
    if addend is not None:
        _chk(addend, x.dtype)
        _require(addend.shape == y.shape, 'argument check failed: addend.shape == y.shape')
    _lib.check(_L().nopesac_upsample2x_bilinear_nhwc(_p(x), _p(addend), _p(y), B, H, W, C, act, _DT[x.dtype], _stream()),
               "nopesac_upsample2x_bilinear_nhwc")
    return y


def upsample2x_nearest_add(x: torch.Tensor, lateral: torch.Tensor) -> torch.Tensor:
    _chk(x); _chk(lateral, x.dtype)
    B, H, W, C = x.shape
    _require(lateral.shape == (B, 2 * H, 2 * W, C), 'argument check failed: lateral.shape == (B, 2 * H, 2 * W, C)')
    y = torch.empty_like(lateral)
    _lib.check(_L().nopesac_upsample2x_nearest_add_nhwc(_p(x), _p(lateral), _p(y), B, H, W, C, _DT[x.dtype], _stream()),
               "nopesac_upsample2x_nearest_add_nhwc")
    return y


def groupnorm(x: torch.Tensor, gamma, beta, groups: int, eps: float, act=ACT_NONE) -> torch.Tensor:
    _chk(x)
    B, H, W, C = x.shape
    y = torch.empty_like(x)
    ws = torch.empty(B * 16 * groups * 2, device=x.device, dtype=torch.float32)
    _lib.check(_L().nopesac_groupnorm_nhwc(_p(x), _p(_chk(gamma, torch.float32)), _p(_chk(beta, torch.float32)), _p(y), B, H * W, C,
                                           groups, eps, act, _DT[x.dtype], _p(ws), _stream()), "nopesac_groupnorm_nhwc")
    return y


def layernorm(x: torch.Tensor, gamma, beta, res=None, addend=None, eps=1e-5):
    """-> y (and y + addend if `addend` [rows_a, D] is given; row index taken modulo rows_a)."""
    _chk(x, torch.float32)
    D = x.shape[-1]
    rows = x.numel() // D
    y = torch.empty_like(x)
    y2 = torch.empty_like(x) if addend is not None else None
    if res is not None:
        _chk(res, torch.float32)
        _require(res.shape == x.shape, 'argument check failed: res.shape == x.shape')
    a_rows = 0 if addend is None else _chk(addend, torch.float32).numel() // D
    _lib.check(_L().nopesac_layernorm(_p(x), _p(res), _p(_chk(gamma, torch.float32)), _p(_chk(beta, torch.float32)), _p(y),
                                      _p(addend), a_rows, _p(y2), rows, D, eps, _stream()), "nopesac_layernorm")
    return (y, y2) if addend is not None else y


def layernorm_ex(x: torch.Tensor, gamma, beta, res=None, addend=None, eps=1e-5, want=("y",)):
    """LayerNorm with a chosen set of outputs: "y" (f32), "y16" (bf16 copy), "y2" (y + addend, f32), "y2_16" (bf16).
    Returns a dict with exactly the requested tensors (one launch)."""
    _chk(x, torch.float32)
    D = x.shape[-1]
    rows = x.numel() // D
    out = {k: torch.empty(x.shape, device=x.device, dtype=torch.bfloat16 if k.endswith("16") else torch.float32) for k in want}
    _require(set(want) <= {"y", "y16", "y2", "y2_16"} and want, 'argument check failed: set(want) <= {"y", "y16", "y2", "y2_16"} and want')
    if res is not None:
        _chk(res, torch.float32)
        _require(res.shape == x.shape, 'argument check failed: res.shape == x.shape')
    a_rows = 0 if addend is None else _chk(addend, torch.float32).numel() // D
    _require(addend is not None or not ({"y2", "y2_16"} & set(want)), 'argument check failed: addend is not None or not ({"y2", "y2_16"} & set(want))')
    _lib.check(_L().nopesac_layernorm_ex(_p(x), _p(res), _p(_chk(gamma, torch.float32)), _p(_chk(beta, torch.float32)), _p(out.get("y")),
                                         _p(addend), a_rows, _p(out.get("y2")), _p(out.get("y16")), _p(out.get("y2_16")), rows, D, eps,
                                         _stream()), "nopesac_layernorm_ex")
    return out


def add_rows(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    _chk(a, torch.float32); _chk(b, torch.float32)
    D = a.shape[-1]
    out = torch.empty_like(a)
    _lib.check(_L().nopesac_add_rows(_p(a), _p(b), _p(out), a.numel() // D, D, b.numel() // D, _stream()), "nopesac_add_rows")
    return out


def softmax_rows(x: torch.Tensor, out_dtype=None, pad_to: int = 0) -> torch.Tensor:
    """softmax over the last dim; with out_dtype (float32 / bfloat16) and / or pad_to > D the result is written into rows of
    max(D, pad_to) elements of that type, zero beyond D (one launch instead of softmax + zero fill + strided cast copy)."""
    _chk(x, torch.float32)
    D = x.shape[-1]
    out_dtype = out_dtype or torch.float32
    if out_dtype == torch.float32 and pad_to <= D:
        y = torch.empty_like(x)
        _lib.check(_L().nopesac_softmax_rows(_p(x), _p(y), x.numel() // D, D, _stream()), "nopesac_softmax_rows")
        return y
    ld = max(D, pad_to)
    y = torch.empty(x.shape[:-1] + (ld,), device=x.device, dtype=out_dtype)
    _lib.check(_L().nopesac_softmax_rows_pad(_p(x), _p(y), x.numel() // D, D, ld, _DT[out_dtype], _stream()), "nopesac_softmax_rows_pad")
    return y


def cached_constant(cache: dict, key, make):
    """A read-only device tensor built once per key and shared by every later call ON ANY STREAM: the stream that builds it is waited
    for before the tensor is published, so that a batch on another stream can never read it half-written (the kernels that fill it
    are not ordered against other streams by anything else)."""
    t = cache.get(key)
    if t is None:
        t = make()
        first = t[0] if isinstance(t, tuple) else t
        if first.is_cuda and not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream(first.device).synchronize()
        cache[key] = t
    return t


def metric_rows(trans, rot, n1, n2, m, pair_idx0: int, t_err=None, r_err=None, nonfinite=None) -> torch.Tensor:
    """[B,16] f32 result rows of the runner (runner.metric_rows on a GPU): one launch."""
    _chk(trans, torch.float32); _chk(rot, torch.float32); _chk(n1, torch.int32); _chk(n2, torch.int32); _chk(m, torch.int32)
    B = trans.shape[0]
    _require(trans.shape == (B, 3) and rot.shape == (B, 4) and n1.numel() == B and n2.numel() == B and m.numel() == B, "metric_rows: shapes")
    for v in (t_err, r_err):
        if v is not None:
            _chk(v, torch.float32)
    if nonfinite is not None:
        _chk(nonfinite, torch.int32)
    rows = torch.empty(B, 16, device=trans.device, dtype=torch.float32)
    _lib.check(_L().nopesac_metric_rows(_p(trans), _p(rot), _p(n1), _p(n2), _p(m), _p(t_err), _p(r_err), _p(nonfinite), int(pair_idx0), _p(rows),
                                        B, _stream()), "nopesac_metric_rows")
    return rows


def concat_cols(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """[a | b] along the last dim of two contiguous f32 row tensors."""
    _chk(a, torch.float32); _chk(b, torch.float32)
    rows, Da, Db = a.shape[0], a.shape[-1], b.shape[-1]
    _require(a.dim() == 2 and b.dim() == 2 and b.shape[0] == rows, "concat_cols: [rows, Da], [rows, Db]")
    out = torch.empty(rows, Da + Db, device=a.device, dtype=torch.float32)
    _lib.check(_L().nopesac_concat_cols(_p(a), Da, _p(b), Db, _p(out), rows, _stream()), "nopesac_concat_cols")
    return out


def add_rows_bf16(a: torch.Tensor, b: torch.Tensor):
    """(a as bf16, a + b as bf16), b broadcast over blocks of b.shape[0] rows; one launch."""
    _chk(a, torch.float32); _chk(b, torch.float32)
    D = a.shape[-1]
    a16 = torch.empty(a.shape, device=a.device, dtype=torch.bfloat16)
    ab16 = torch.empty(a.shape, device=a.device, dtype=torch.bfloat16)
    _lib.check(_L().nopesac_add_rows_bf16(_p(a), _p(b), _p(a16), _p(ab16), a.numel() // D, D, b.numel() // D, _stream()), "nopesac_add_rows_bf16")
    return a16, ab16


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, B: int, Lq: int, Lk: int, heads: int, scale: float,
              qlen=None, klen=None, mfma_bf16: bool = False) -> torch.Tensor:
    """q [B*Lq, >=heads*32] etc. as (possibly column-sliced) row-major matrices -> o [B*Lq, heads*32]."""
    io16 = q.dtype == torch.bfloat16          # bf16 q/k/v in memory -> bf16 o (MFMA kernel only)
    for t in (q, k, v):
        _require(t.is_cuda and t.dtype == q.dtype and t.dim() == 2 and t.stride(1) == 1, 'argument check failed: t.is_cuda and t.dtype == q.dtype and t.dim() == 2 and t.stride(1) == 1')
    _require(q.dtype == torch.float32 or (io16 and mfma_bf16), 'argument check failed: q.dtype == torch.float32 or (io16 and mfma_bf16)')
    _require(q.shape[0] == B * Lq and k.shape[0] == B * Lk and v.shape[0] == B * Lk, 'argument check failed: q.shape[0] == B * Lq and k.shape[0] == B * Lk and v.shape[0] == B * Lk')
    o = torch.empty((B * Lq, heads * 32), device=q.device, dtype=q.dtype)
    fn = _L().nopesac_attention_small_bf16io if io16 else (_L().nopesac_attention_small_bf16 if mfma_bf16 else _L().nopesac_attention_small)
    rc = fn(_p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(o), o.stride(0), B, Lq, Lk,
            heads, scale, _p(qlen), _p(klen), _stream())
    _lib.check(rc, "nopesac_attention_small" + ("_bf16" if mfma_bf16 else ""))
    return o


def transpose_hw_rows(x: torch.Tensor, H: int, W: int) -> torch.Tensor:
    _chk(x, torch.float32)
    B, C = x.shape[0], x.shape[-1]
    y = torch.empty_like(x)
    _lib.check(_L().nopesac_transpose_hw_rows(_p(x), _p(y), B, H, W, C, _stream()), "nopesac_transpose_hw_rows")
    return y


def count_nonfinite(tensors, counter: Optional[torch.Tensor] = None) -> torch.Tensor:
    """int32[1] device counter += number of Inf / NaN values in the given f32 tensors (no host synchronisation)."""
    if counter is None:
        counter = torch.zeros(1, device=tensors[0].device, dtype=torch.int32)
    tensors = [t for t in tensors if t.numel()]
    for i in range(0, len(tensors), 16):                      # NOPESAC_NONFINITE_MAX_TENSORS per launch
        grp = tensors[i:i + 16]
        for t in grp:
            _chk(t, torch.float32)
        ptrs = (ctypes.c_void_p * len(grp))(*[_p(t) for t in grp])
        cnts = (ctypes.c_int64 * len(grp))(*[t.numel() for t in grp])
        _lib.check(_L().nopesac_count_nonfinite_batch(ptrs, cnts, len(grp), _p(counter), _stream()), "nopesac_count_nonfinite_batch")
    return counter


class HostFetch:
    """A batch of device tensors on its way to the host, on the stream that was current at construction.  Up to 16 MB: ONE kernel per
    32 tensors (`nopesac_gather_bytes`) stores their bytes straight into one pinned, device-mapped host buffer (16-byte aligned
    segments) - no concatenation, no pad fills, no copy node, so a captured graph / launch tape can hold the fetch.  Larger sets: one
    torch.cat on the device + one copy.  `views()` hands out host tensors of the original dtype / shape (views of that buffer) once the caller
    knows the fetch has completed - after `wait()`, or after an event it recorded behind the fetch on the same stream.
    private_views = True (a fetch recorded in a graph: the buffer is written again by every replay): views() snapshots the buffer.
    dynamic = {name: int64[1] device tensor}: only that many leading BYTES of the named tensor are valid (known on the device only); the
    rest of its host view is undefined.  Such a tensor does not count towards the 16 MB bound of the kernel path."""
    KERNEL_MAX_BYTES, KERNEL_MAX_SEGMENTS, ALIGN = 16 << 20, 32, 16
    _zero_pad = {}

    def __init__(self, tensors: dict, private_views: bool = False, dynamic: Optional[dict] = None, host: Optional[torch.Tensor] = None):
        dynamic = dynamic or {}
        dev = [(k, t.detach().contiguous()) for k, t in tensors.items() if t.is_cuda]
        self.passthrough = {k: t.detach() for k, t in tensors.items() if not t.is_cuda}
        self.spans, self.host, self.stream, self.private_views = {}, None, None, bool(private_views)
        if not dev:
            return
        off, segs, fixed = 0, [], 0
        for k, t in dev:
            nb = t.numel() * t.element_size()
            self.spans[k] = (off, nb, t.dtype, tuple(t.shape))
            if nb:
                segs.append((t, nb, off, dynamic.get(k)))
                fixed += 0 if k in dynamic else nb
            off += nb + (-nb) % self.ALIGN
        self.stream = torch.cuda.current_stream()
        if not segs:
            self.host = torch.empty(0, dtype=torch.uint8)
            return
        if fixed <= self.KERNEL_MAX_BYTES:
            # (the bytes between segments are never read; `host`: a pinned uint8 buffer the caller allocated - inside a stream capture
            #  nothing may be allocated on the host)
            if host is not None:
                _require(host.dtype == torch.uint8 and host.is_pinned() and host.numel() >= off, "HostFetch: host buffer too small / not pinned")
            self.host = host[:off] if host is not None else torch.empty(off, dtype=torch.uint8, pin_memory=True)
            for i in range(0, len(segs), self.KERNEL_MAX_SEGMENTS):             # NOPESAC_GATHER_MAX_SEGMENTS per launch
                grp = segs[i:i + self.KERNEL_MAX_SEGMENTS]
                n = len(grp)
                for _, _, _, dyn in grp:
                    if dyn is not None:
                        _chk(dyn, torch.int64)
                ptrs = (ctypes.c_void_p * n)(*[_p(t) for t, _, _, _ in grp])
                sizes = (ctypes.c_int64 * n)(*[nb for _, nb, _, _ in grp])
                dyns = (ctypes.c_void_p * n)(*[_p(dyn) for _, _, _, dyn in grp])
                offs = (ctypes.c_int64 * n)(*[o for _, _, o, _ in grp])
                _lib.check(_L().nopesac_gather_bytes(ptrs, sizes, dyns, offs, n, self.host.data_ptr(), _stream()), "nopesac_gather_bytes")
            self._keep = [(t, dyn) for t, _, _, dyn in segs]                     # sources stay allocated until the object goes
            return
        _require(host is None, "HostFetch: more than KERNEL_MAX_BYTES of fixed-size tensors cannot go into a caller-provided buffer")
        d0 = dev[0][1].device
        z = HostFetch._zero_pad.get(d0)
        if z is None:
            z = HostFetch._zero_pad[d0] = torch.zeros(self.ALIGN, device=d0, dtype=torch.uint8)
        parts = []
        for k, t in dev:
            nb = t.numel() * t.element_size()
            if nb:
                parts.append(t.view(-1).view(torch.uint8))
            if (-nb) % self.ALIGN:
                parts.append(z[:(-nb) % self.ALIGN])
        flat = torch.cat(parts)
        self.host = torch.empty(flat.numel(), dtype=torch.uint8, pin_memory=True)
        self.host.copy_(flat, non_blocking=True)

    def wait(self) -> "HostFetch":
        if self.stream is not None:
            self.stream.synchronize()
        return self

    def host_bytes(self) -> int:
        return 0 if self.host is None else int(self.host.numel())

    def views(self) -> dict:
        out = dict(self.passthrough)
        host = self.host
        if self.private_views and host is not None:
            # (a plain memcpy into pageable memory: .clone() of a pinned tensor allocates PINNED memory - a hipHostMalloc of ~10 ms
            #  whenever the results of earlier calls still hold the allocator's cached blocks - and torch's copy_ from a pinned source
            #  measured 1-30 ms per call where numpy's copy of the same bytes takes 50 us)
            host = torch.from_numpy(self.host.numpy().copy())
        for k, (o, n, dt, shape) in self.spans.items():
            out[k] = host[o:o + n].view(dt).view(shape)
        return out


def gather_to_host(tensors: dict) -> dict:
    """{name: tensor} -> {name: host tensor of the same dtype / shape}.  Device tensors travel together: one concatenation on the
    device, one copy into a fresh pinned host buffer, one wait; the returned tensors are views of that buffer."""
    return HostFetch(tensors).wait().views()


def u8_to_f32(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """uint8 tensor -> float32 (exact), e.g. the image planes of a batch after an 8-bit host-to-device copy."""
    _chk(x, torch.uint8)
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    _chk(out, torch.float32)
    _require(out.numel() == x.numel(), "u8_to_f32: sizes differ")
    _lib.check(_L().nopesac_u8_to_f32(_p(x), _p(out), x.numel(), _stream()), "nopesac_u8_to_f32")
    return out


def clock_probe(spin_cycles: int = 400000, stream=None) -> torch.Tensor:
    """Enqueue the engine-clock probe (one wave, ~0.2 ms) on `stream` (default: current); returns an int64[2] device tensor
    (shader cycles, 100 MHz ticks) valid once the stream has passed it: MHz = 100 * t[0] / t[1]."""
    out = torch.zeros(2, device="cuda", dtype=torch.int64)
    _lib.check(_L().nopesac_clock_probe(_p(out), int(spin_cycles), stream.cuda_stream if stream is not None else _stream()), "nopesac_clock_probe")
    return out


def normalize_rows(x: torch.Tensor, canonical_sign: bool = False) -> torch.Tensor:
    _chk(x, torch.float32)
    D = x.shape[-1]
    y = torch.empty_like(x)
    _lib.check(_L().nopesac_normalize_rows(_p(x), _p(y), x.numel() // D, D, int(canonical_sign), _stream()), "nopesac_normalize_rows")
    return y


def postselect_planes(cls_logits, mask_prob, params, query_feat, H, W, score_thr, mask_thr, overlap_thr, planar: bool = False) -> dict:
    """mask_prob: [B,h,w,nq], or [B,nq,h,w] when `planar`."""
    _chk(cls_logits, torch.float32); _chk(mask_prob, torch.float32); _chk(params, torch.float32); _chk(query_feat, torch.float32)
    B, nq, _ = cls_logits.shape
    if planar:
        _, nq2, h, w = mask_prob.shape
    else:
        _, h, w, nq2 = mask_prob.shape
    _require(nq2 == nq, 'argument check failed: nq2 == nq')
    D = query_feat.shape[-1]
    dev = cls_logits.device
    i32 = dict(device=dev, dtype=torch.int32)
    f32 = dict(device=dev, dtype=torch.float32)
    out = {
        "n_kept": torch.empty(B, **i32), "kept_idx": torch.empty(B, nq, **i32), "planes": torch.empty(B, nq, 3, **f32),
        "feats": torch.empty(B, nq, D, **f32), "scores": torch.empty(B, nq, **f32), "areas": torch.empty(B, nq, **i32),
        "centers": torch.empty(B, nq, 2, **f32), "winner": torch.zeros(B, H, W, device=dev, dtype=torch.uint8),
        "flags": torch.empty(B, **i32),
    }
    work = torch.empty(B, 9 * nq + 8, **i32)
    rc = _L().nopesac_postselect_planes_ex(_p(cls_logits), _p(mask_prob), _p(params), _p(query_feat), B, nq, D, h, w, H, W,
                                           score_thr, mask_thr, overlap_thr, _p(out["n_kept"]), _p(out["kept_idx"]),
                                           _p(out["planes"]), _p(out["feats"]), _p(out["scores"]), _p(out["areas"]),
                                           _p(out["centers"]), _p(out["winner"]), _p(out["flags"]), _p(work), int(planar), _stream())
    _lib.check(rc, "nopesac_postselect_planes_ex")
    return out


def matcher_sinkhorn(desc_dot, planes1, planes2, cam7, n1, n2, bin_score, offset_mult, normal_mult, iters, match_thr):
    B, nq, _ = desc_dot.shape
    for t in (desc_dot, planes1, planes2, cam7, bin_score):
        _chk(t, torch.float32)
    _chk(n1, torch.int32); _chk(n2, torch.int32)
    log_scores = torch.empty(B, nq + 1, nq + 1, device=desc_dot.device, dtype=torch.float32)
    assignment = torch.empty(B, nq, nq, device=desc_dot.device, dtype=torch.float32)
    rc = _L().nopesac_matcher_sinkhorn(_p(desc_dot), _p(planes1), _p(planes2), _p(cam7), _p(n1), _p(n2), _p(bin_score),
                                       offset_mult, normal_mult, iters, match_thr, B, nq, _p(log_scores), _p(assignment), _stream())
    _lib.check(rc, "nopesac_matcher_sinkhorn")
    return log_scores, assignment


def geo_sequence(assignment, planes1, planes2, n1, n2, init_trans, init_rot, warp_in_ref=True):
    B, nq, _ = assignment.shape
    for t in (assignment, planes1, planes2, init_trans, init_rot):
        _chk(t, torch.float32)
    dev = assignment.device
    f32 = dict(device=dev, dtype=torch.float32)
    geo_local, geo_global = torch.empty(B, nq, 6, **f32), torch.empty(B, nq, 6, **f32)
    sig, geo_enc = torch.empty(B, nq, **f32), torch.empty(B, nq, 8, **f32)
    m = torch.empty(B, device=dev, dtype=torch.int32)
    rc = _L().nopesac_geo_sequence(_p(assignment), _p(planes1), _p(planes2), _p(n1), _p(n2), _p(init_trans), _p(init_rot), B, nq,
                                   int(warp_in_ref), _p(geo_local), _p(geo_global), _p(sig), _p(geo_enc), _p(m), _stream())
    _lib.check(rc, "nopesac_geo_sequence")
    return geo_local, geo_global, sig, geo_enc, m


def ransac_score_maps(geo_local, rot_raw, trans_raw, init_rot, init_trans, m, diagnostics=True):
    B, nq, _ = geo_local.shape
    for t in (geo_local, rot_raw, trans_raw, init_rot, init_trans):
        _chk(t, torch.float32)
    dev = geo_local.device
    f32 = dict(device=dev, dtype=torch.float32)
    out = {"rots_all": torch.empty(B, nq + 1, 4, **f32), "trans_all": torch.empty(B, nq + 1, 3, **f32),
           "normal_score": torch.empty(B, nq + 1, nq, **f32), "param_score": torch.empty(B, nq + 1, nq, **f32),
           "dn_sum": torch.empty(B, nq + 1, **f32), "dl2_sum": torch.empty(B, nq + 1, **f32)}
    if diagnostics:
        for k in ("l2_dist", "normal_angle", "offset_dist"):
            out[k] = torch.empty(B, nq + 1, nq, **f32)
    rc = _L().nopesac_ransac_score_maps(_p(geo_local), _p(rot_raw), _p(trans_raw), _p(init_rot), _p(init_trans), _p(m), B, nq,
                                        _p(out["rots_all"]), _p(out["trans_all"]), _p(out["normal_score"]), _p(out["param_score"]),
                                        _p(out.get("l2_dist")), _p(out.get("normal_angle")), _p(out.get("offset_dist")),
                                        _p(out["dn_sum"]), _p(out["dl2_sum"]), _stream())
    _lib.check(rc, "nopesac_ransac_score_maps")
    return out


def ransac_soft_vote(sf_rot, sf_trans, reg_rot_w, reg_rot_b, reg_trans_w, reg_trans_b, init_rot_feat, init_trans_feat,
                     fused_rot, fused_trans, rots_w, rots_b, trans_w, trans_b, maps, init_rot, init_trans, m, mode: int):
    B, NH, _ = sf_rot.shape
    nq = NH - 1
    dev = sf_rot.device
    f32 = dict(device=dev, dtype=torch.float32)
    out = {"pred_rot": torch.empty(B, 4, **f32), "pred_trans": torch.empty(B, 3, **f32), "avg_rot": torch.empty(B, 4, **f32),
           "avg_trans": torch.empty(B, 3, **f32), "score_rot": torch.empty(B, NH, **f32), "score_trans": torch.empty(B, NH, **f32)}
    args = [sf_rot, sf_trans, reg_rot_w, reg_rot_b, reg_trans_w, reg_trans_b, init_rot_feat, init_trans_feat, fused_rot,
            fused_trans, rots_w, rots_b, trans_w, trans_b, maps["rots_all"], maps["trans_all"], maps["dn_sum"], maps["dl2_sum"],
            init_rot, init_trans]
    for t in args:
        _chk(t, torch.float32)
    rc = _L().nopesac_ransac_soft_vote(*[_p(t) for t in args], _p(m), B, nq, mode, _p(out["pred_rot"]), _p(out["pred_trans"]),
                                       _p(out["avg_rot"]), _p(out["avg_trans"]), _p(out["score_rot"]), _p(out["score_trans"]), _stream())
    _lib.check(rc, "nopesac_ransac_soft_vote")
    return out


PLANE_CAM_REF_LOSS_NAMES = ("loss_tran_planeAvgReg", "loss_rot_planeAvgReg", "loss_tran_planeSoftReg", "loss_rot_planeSoftReg",
                            "loss_rotIdx", "loss_transIdx", "loss_paramL2_dist")


def plane_cam_ref_losses(vote: dict, maps: dict, m, gt_pose, weight: float = 1.0):
    """The seven refinement losses of the training-side twin (include/nopesac_hip.h: nopesac_plane_cam_ref_losses; reference
    camera_head.py:883-921) -> f32[7] in PLANE_CAM_REF_LOSS_NAMES order.  `vote` from ransac_soft_vote(mode | 16), `maps` from
    ransac_score_maps(diagnostics=True)."""
    _require("l2_dist" in maps, "plane_cam_ref_losses: ransac_score_maps must run with diagnostics=True (the parameter loss reads l2_dist)")
    B, NH, _ = maps["rots_all"].shape
    _chk(gt_pose, torch.float32)
    _require(tuple(gt_pose.shape) == (B, 7) and m.dtype == torch.int32 and m.numel() == B, "plane_cam_ref_losses: gt_pose [B,7], m int32[B]")
    args = [vote["pred_rot"], vote["pred_trans"], vote["avg_rot"], vote["avg_trans"], maps["rots_all"], maps["trans_all"],
            vote["score_rot"], vote["score_trans"], maps["l2_dist"]]
    for t in args:
        _chk(t, torch.float32)
    losses = torch.empty(7, device=gt_pose.device, dtype=torch.float32)
    rc = _L().nopesac_plane_cam_ref_losses(*[_p(t) for t in args], _p(m), _p(gt_pose), B, NH - 1, float(weight), _p(losses), _stream())
    _lib.check(rc, "nopesac_plane_cam_ref_losses")
    return losses


def camera_pose_loss(est_trans, est_rot, gt_trans, gt_rot, weight: float = 1.0, trans_eps: float = 0.0):
    """CameraPoseLoss / the AIM's reconstruction losses (include/nopesac_hip.h: nopesac_camera_pose_loss) -> f32[2] = (l_x, l_q) * weight.
    `gt_trans` / `gt_rot` may be column views of one [B,7] pose tensor (row strides are passed on)."""
    for t in (est_trans, est_rot):
        _chk(t, torch.float32)
    for t in (gt_trans, gt_rot):
        _chk(t, torch.float32, contiguous=False)
        _require(t.dim() == 2 and t.stride(1) == 1, "camera_pose_loss: gt rows must be dense")
    B = est_trans.shape[0]
    _require(tuple(est_trans.shape) == (B, 3) and tuple(est_rot.shape) == (B, 4) and tuple(gt_trans.shape) == (B, 3) and
             tuple(gt_rot.shape) == (B, 4), "camera_pose_loss: shapes")
    out = torch.empty(2, device=est_trans.device, dtype=torch.float32)
    rc = _L().nopesac_camera_pose_loss(_p(est_trans), _p(est_rot), _p(gt_trans), gt_trans.stride(0), _p(gt_rot), gt_rot.stride(0), B,
                                       float(trans_eps), float(weight), _p(out), _stream())
    _lib.check(rc, "nopesac_camera_pose_loss")
    return out


def refilter_assignment(assignment, planes1, planes2, n1, n2, rot, trans):
    B, nq, _ = assignment.shape
    out = torch.empty_like(assignment)
    rc = _L().nopesac_refilter_assignment(_p(_chk(assignment, torch.float32)), _p(planes1), _p(planes2), _p(n1), _p(n2), _p(_chk(rot, torch.float32)),
                                          _p(_chk(trans, torch.float32)), B, nq, _p(out), _stream())
    _lib.check(rc, "nopesac_refilter_assignment")
    return out


def force_k_select(logits: torch.Tensor, query_feat: torch.Tensor, perm: torch.Tensor, noise: torch.Tensor, B: int, K: int):
    """Benchmark-only K control in one launch (include/nopesac_hip.h: nopesac_force_k_select) -> feats f32 [2B,nq,D], n_kept int32 [2B]."""
    _chk(logits, torch.float32); _chk(query_feat, torch.float32); _chk(perm, torch.int64); _chk(noise, torch.float32)
    nq, n_cls, D = logits.shape[1], logits.shape[2], query_feat.shape[2]
    _require(logits.shape[0] >= B and query_feat.shape[0] >= B and query_feat.shape[1] == nq and perm.shape == (B, K) and noise.shape == (B, K, D),
             "force_k_select: shapes")
    feats = torch.empty(2 * B, nq, D, device=logits.device, dtype=torch.float32)
    n_kept = torch.empty(2 * B, device=logits.device, dtype=torch.int32)
    _lib.check(_L().nopesac_force_k_select(_p(logits), n_cls, _p(query_feat), _p(perm), _p(noise), B, nq, K, D, _p(feats), _p(n_kept), _stream()),
               "nopesac_force_k_select")
    return feats, n_kept


def rle_labels(winner: torch.Tensor, kept_idx: torch.Tensor, n_kept: torch.Tensor, flags: torch.Tensor) -> torch.Tensor:
    """winner uint8 [V,H,W] -> column-major uint8 [V,W,H] map of kept-plane ordinals (0xFF = none)."""
    _chk(winner, torch.uint8); _chk(kept_idx, torch.int32); _chk(n_kept, torch.int32); _chk(flags, torch.int32)
    V, H, W = winner.shape
    labels = torch.empty((V, W, H), device=winner.device, dtype=torch.uint8)
    _lib.check(_L().nopesac_rle_labels(_p(winner), _p(kept_idx), _p(n_kept), _p(flags), _p(labels), V, H, W, kept_idx.shape[1],
                                       _stream()), "nopesac_rle_labels")
    return labels


def rle_transitions(labels: torch.Tensor, n_kept: torch.Tensor, nq: int, offsets: torch.Tensor = None,
                    positions: torch.Tensor = None) -> torch.Tensor:
    """counts int32 [V,nq] of mask flips per (view, plane); with offsets (int64 [V,nq]) + positions (int32 buffer) also
    writes the ascending flip positions."""
    _chk(labels, torch.uint8); _chk(n_kept, torch.int32)
    V, W, H = labels.shape
    counts = torch.empty((V, nq), device=labels.device, dtype=torch.int32)
    if positions is not None:
        _chk(offsets, torch.int64); _chk(positions, torch.int32)
    _lib.check(_L().nopesac_rle_transitions(_p(labels), _p(n_kept), _p(offsets) if positions is not None else None, _p(counts),
                                            _p(positions) if positions is not None else None, V, W * H, nq, _stream()),
               "nopesac_rle_transitions")
    return counts


def decode_masks(winner: torch.Tensor, kept_idx: torch.Tensor, n_kept: torch.Tensor, flags: torch.Tensor, total: int) -> torch.Tensor:
    """bool [total, H, W]: the dense masks of every kept plane of every view, view after view (total = sum(n_kept), known to the
    caller); one launch for the whole batch."""
    _chk(winner, torch.uint8); _chk(kept_idx, torch.int32); _chk(n_kept, torch.int32); _chk(flags, torch.int32)
    V, H, W = winner.shape
    n64 = n_kept.to(torch.int64)
    offsets = (torch.cumsum(n64, 0) - n64).contiguous()
    masks = torch.empty((max(total, 1), H, W), device=winner.device, dtype=torch.uint8)
    _lib.check(_L().nopesac_decode_masks(_p(winner), _p(kept_idx), _p(n_kept), _p(flags), _p(offsets), _p(masks), V, H, W, kept_idx.shape[1],
                                         _stream()), "nopesac_decode_masks")
    return masks[:total].view(torch.bool)


def rle_compress(positions: torch.Tensor, offsets: torch.Tensor, counts: torch.Tensor, H: int, W: int):
    """Flip positions of n masks (device: positions int32 buffer, offsets int64 [n], counts int32 [n]) -> COCO counts strings on
    the device: (bytes uint8 [total], out_off int64 [n], lens int32 [n], bbox float64 [n,4]) - all device tensors; one host
    sync inside (the total string length sizes the byte buffer)."""
    _chk(positions, torch.int32); _chk(offsets, torch.int64); _chk(counts, torch.int32)
    n = counts.numel()
    dev = counts.device
    lens = torch.empty(n, device=dev, dtype=torch.int32)
    bbox = torch.empty(n, 4, device=dev, dtype=torch.float64)
    L = _L()
    _lib.check(L.nopesac_rle_compress_device(_p(positions), _p(offsets), _p(counts), n, H, W, _p(lens), _p(bbox), None, None, _stream()),
               "nopesac_rle_compress_device")
    ends = torch.cumsum(lens.to(torch.int64), 0)
    out_off = (ends - lens).contiguous()
    total = int(ends[-1].item())                                         # host sync
    out = torch.empty(max(total, 1), device=dev, dtype=torch.uint8)
    _lib.check(L.nopesac_rle_compress_device(_p(positions), _p(offsets), _p(counts), n, H, W, None, None, _p(out), _p(out_off), _stream()),
               "nopesac_rle_compress_device")
    return out[:total], out_off, lens, bbox


def rle_compress_capped(positions: torch.Tensor, offsets: torch.Tensor, counts: torch.Tensor, H: int, W: int, cap: int):
    """rle_compress without the host sync: the strings go into a byte buffer of FIXED capacity `cap`; a string that would cross it
    is not written.  Returns (bytes uint8 [cap], out_off int64 [n], lens int32 [n], bbox float64 [n,4], total int64 [1]) - device
    tensors; the caller checks out_off[-1] + lens[-1] (= total) <= cap once those have reached the host."""
    _chk(positions, torch.int32); _chk(offsets, torch.int64); _chk(counts, torch.int32)
    n = counts.numel()
    dev = counts.device
    lens = torch.empty(n, device=dev, dtype=torch.int32)
    bbox = torch.empty(n, 4, device=dev, dtype=torch.float64)
    L = _L()
    _lib.check(L.nopesac_rle_compress_device(_p(positions), _p(offsets), _p(counts), n, H, W, _p(lens), _p(bbox), None, None, _stream()),
               "nopesac_rle_compress_device")
    ends = torch.cumsum(lens.to(torch.int64), 0)
    out_off = (ends - lens).contiguous()
    out = torch.empty(int(cap), device=dev, dtype=torch.uint8)
    _lib.check(L.nopesac_rle_compress_device_capped(_p(positions), _p(offsets), _p(counts), n, H, W, _p(lens), _p(out), _p(out_off), int(cap),
                                                    _stream()), "nopesac_rle_compress_device_capped")
    return out, out_off, lens, bbox, ends[-1:]


GNN_PREFETCH = os.environ.get("NOPESAC_GNN_PREFETCH", "1") != "0"


def gnn_layer(x: torch.Tensor, x_off: int, src: torch.Tensor, src_off: int, out: torch.Tensor, out_off: int, n_sets: int, lens, W: dict,
              W_next: Optional[dict] = None, next_sets: int = 0):
    """One fused GNN layer (csrc/gnn_layer.hip): x/src/out f32 [sets, nq, 256]; lens int32 [sets] (indexed like x / src);
    W: fragment-major bf16 weights "wq" (pre-scaled), "wk", "wv", "wm", "w0", "w2" and f32 "g1", "b1", "g2", "b2".
    W_next / next_sets: the weights and the set count of the NEXT launch - with few plane sets (one pair per call) extra workgroups
    of this launch read them into the L2s the next launch will run on (nopesac_gnn_layer_bf16_pf)."""
    for t in (x, src, out):
        _chk(t, torch.float32)
        _require(t.dim() == 3 and t.shape[2] == 256, 'argument check failed: t.dim() == 3 and t.shape[2] == 256')
    nq = x.shape[1]
    _require(nq <= 128 and src.shape[1] == nq and out.shape[1] == nq, 'gnn_layer: nq <= 128, src / out with the same nq')
    _require(x_off + n_sets <= x.shape[0] and src_off + n_sets <= src.shape[0] and out_off + n_sets <= out.shape[0], 'argument check failed: x_off + n_sets <= x.shape[0] and src_off + n_sets <= src.shape[0] and out_off + n_sets <= out.shape[0]')
    if lens is not None:
        _chk(lens, torch.int32)
    nxt = None
    if W_next is not None and GNN_PREFETCH and next_sets > 0:
        nxt = (ctypes.c_void_p * 6)(*[_p(W_next[k]) for k in ("wq", "wk", "wv", "wm", "w0", "w2")])
    rc = _L().nopesac_gnn_layer_bf16_pf(_p(x), x_off, _p(src), src_off, _p(out), out_off, n_sets, nq, _p(lens), _p(lens),
                                        _p(W["wq"]), _p(W["wk"]), _p(W["wv"]), _p(W["wm"]), _p(W["w0"]), _p(W["w2"]),
                                        _p(W["g1"]), _p(W["b1"]), _p(W["g2"]), _p(W["b2"]), nxt, int(next_sets) if nxt is not None else 0, _stream())
    _lib.check(rc, "nopesac_gnn_layer_bf16_pf")
    return out


def encoder_tail(attn: torch.Tensor, src: torch.Tensor, W: dict, pos=None, want=("y", "y16", "ypos16")) -> dict:
    """Fused out-proj + residual + LN1 + FFN + residual + LN2 of a post-norm encoder layer (csrc/enc_tail.hip).
    attn bf16 [M,256], src f32 [M,256]; W: fragment-major bf16 "wo", "w1", "w2" + f32 "bo", "g1", "be1", "b1", "b2", "g2", "be2"."""
    _chk(attn, torch.bfloat16); _chk(src, torch.float32)
    M = src.shape[0]
    _require(attn.shape == (M, 256) and src.shape == (M, 256), 'argument check failed: attn.shape == (M, 256) and src.shape == (M, 256)')
    out = {k: torch.empty(M, 256, device=src.device, dtype=torch.float32 if k == "y" else torch.bfloat16) for k in want}
    if pos is not None:
        _chk(pos, torch.float32)
    rc = _L().nopesac_encoder_tail_bf16(_p(attn), _p(src), _p(W["wo"]), _p(W["bo"]), _p(W["g1"]), _p(W["be1"]), _p(W["w1"]), _p(W["b1"]),
                                        _p(W["w2"]), _p(W["b2"]), _p(W["g2"]), _p(W["be2"]), _p(pos), 0 if pos is None else pos.shape[0],
                                        _p(out.get("y")), _p(out.get("y16")), _p(out.get("ypos16")), M, _stream())
    _lib.check(rc, "nopesac_encoder_tail_bf16")
    return out


def resize_bilinear_u8(img: torch.Tensor, out_h: int, out_w: int) -> torch.Tensor:
    """uint8 [H,W,C] (device) -> uint8 [out_h,out_w,C], cv2 INTER_LINEAR fixed-point semantics."""
    _chk(img, torch.uint8)
    H, W, C = img.shape
    out = torch.empty((out_h, out_w, C), device=img.device, dtype=torch.uint8)
    _lib.check(_L().nopesac_resize_bilinear_u8(_p(img), H, W, C, _p(out), out_h, out_w, _stream()), "nopesac_resize_bilinear_u8")
    return out


def resize_bilinear_u8_batch(imgs, out_h: int, out_w: int, chw: bool = True) -> torch.Tensor:
    """n uint8 [H,W,C] device images of ONE size lying evenly spaced in one buffer (views of jpeg.decode_batch's output) -> uint8
    [n,C,out_h,out_w] (chw) or [n,out_h,out_w,C], one launch; None if the images are not laid out like that."""
    a = imgs[0]
    _chk(a, torch.uint8)
    H, W, C = a.shape
    n = len(imgs)
    stride = (imgs[1].data_ptr() - a.data_ptr()) if n > 1 else H * W * C
    st = a.untyped_storage().data_ptr()
    if stride < H * W * C or any(im.shape != a.shape or not im.is_contiguous() or im.data_ptr() != a.data_ptr() + k * stride
                                 or im.untyped_storage().data_ptr() != st for k, im in enumerate(imgs)):
        return None
    out = torch.empty((n, C, out_h, out_w) if chw else (n, out_h, out_w, C), device=a.device, dtype=torch.uint8)
    _lib.check(_L().nopesac_resize_bilinear_u8_batch(_p(a), n, stride, H, W, C, _p(out), out_h, out_w, 1 if chw else 0, _stream()),
               "nopesac_resize_bilinear_u8_batch")
    return out


def mask_head(c1: torch.Tensor, t1: torch.Tensor, w_lat_frag: torch.Tensor, scale, bias, mask_w: torch.Tensor, mask_b: torch.Tensor,
              sigmoid: bool = True, want_p1: bool = False, planar: bool = False, fold: Optional[torch.Tensor] = None, taps1: bool = False,
              pipe: bool = False):
    """Fused lateral conv + bilinear add + per-image mask GEMM (csrc/mask_head.hip).  c1 [B,H,W,256] / t1 [B,H/2,W/2,256] bf16;
    mask_w [B,nq,256] (any float dtype), mask_b [B,nq] f32 -> prob f32 [B,H,W,nq] ([B,nq,H,W] when `planar`) (and p1 bf16 if asked).
    `pipe`: the persistent software-pipelined kernel (round-5 experiment, slower; bit-identical) instead of two workgroups per CU."""
    _chk(c1, torch.bfloat16); _chk(t1, torch.bfloat16); _chk(w_lat_frag, torch.bfloat16)
    B, H, W, C = c1.shape
    nq = mask_w.shape[1] if fold is None else fold.shape[0] // B
    _require(C == 256 and t1.shape == (B, H // 2, W // 2, 256) and nq <= 128 and nq % 2 == 0 and (H * W) % 128 == 0,
             'mask_head: C == 256, t1 at half resolution, nq even and <= 128, H * W a multiple of 128')
    nqp = 64 if nq <= 64 else 128                                                       # planes padded to whole pairs of 32-wide MFMA tiles
    if fold is not None:          # both operands straight from the folded plane embeddings [B * nq, >= 257] f32, one launch
        _chk(fold, torch.float32)
        _require(fold.dim() == 2 and fold.shape[0] == B * nq and fold.shape[1] >= 257 and fold.is_contiguous(), "mask_head: fold [B * nq, >= 257]")
        mw = torch.empty(B, nqp // 32, 16, 2, 32, 8, device=c1.device, dtype=torch.bfloat16)
        mb = torch.empty(B, nqp, device=c1.device, dtype=torch.float32)
        _lib.check(_L().nopesac_mask_operands(_p(fold), fold.shape[1], _p(mw), _p(mb), B, nq, nqp, _stream()), "nopesac_mask_operands")
    else:
        mw = torch.zeros(B, nqp, 256, device=c1.device, dtype=torch.bfloat16)
        mw[:, :nq] = mask_w
        mw = mw.view(B, nqp // 32, 32, 16, 2, 8).permute(0, 1, 3, 4, 2, 5).contiguous()    # per-image MFMA fragment-major
        mb = torch.zeros(B, nqp, device=c1.device, dtype=torch.float32)
        mb[:, :nq] = mask_b
    prob = torch.empty((B, nq, H, W) if planar else (B, H, W, nq), device=c1.device, dtype=torch.float32)
    p1 = torch.empty(B, H, W, 256, device=c1.device, dtype=torch.bfloat16) if want_p1 else None
    rc = _L().nopesac_mask_head_bf16(_p(c1), _p(t1), _p(w_lat_frag), _p(scale), _p(bias), _p(mw), _p(mb), _p(prob), _p(p1), B, H, W, nq,
                                     int(sigmoid) | (2 if planar else 0) | (4 if taps1 else 0) | (8 if pipe else 0), _stream())
    _lib.check(rc, "nopesac_mask_head_bf16")
    return (prob, p1) if want_p1 else prob


def decoder_tail(attn: torch.Tensor, tgt: torch.Tensor, W: dict, pos=None, want=("y", "y16", "ypos16")) -> dict:
    """Pre-norm decoder tail (csrc/enc_tail.hip, pre_norm = 1): out-proj + residual, LN3, FFN + residual, next norm.
    W: fragment-major bf16 "wo", "w1", "w2"; f32 "bo", "b1", "b2", "g3", "be3" (norm3), "gn", "ben" (next norm).
    want: "y" (f32 residual stream), "y16" / "ypos16" (bf16 of the normalised result / + pos), "yn" (f32 normalised)."""
    _chk(attn, torch.bfloat16); _chk(tgt, torch.float32)
    M = tgt.shape[0]
    _require(attn.shape == (M, 256) and tgt.shape == (M, 256), 'argument check failed: attn.shape == (M, 256) and tgt.shape == (M, 256)')
    out = {k: torch.empty(M, 256, device=tgt.device, dtype=torch.float32 if k in ("y", "yn") else torch.bfloat16) for k in want}
    if pos is not None:
        _chk(pos, torch.float32)
    rc = _L().nopesac_decoder_tail_bf16(_p(attn), _p(tgt), _p(W["wo"]), _p(W["bo"]), _p(W["g3"]), _p(W["be3"]), _p(W["w1"]), _p(W["b1"]),
                                        _p(W["w2"]), _p(W["b2"]), _p(W["gn"]), _p(W["ben"]), _p(pos), 0 if pos is None else pos.shape[0],
                                        _p(out.get("y")), _p(out.get("y16")), _p(out.get("ypos16")), _p(out.get("yn")), M, _stream())
    _lib.check(rc, "nopesac_decoder_tail_bf16")
    return out


TAIL_PREFETCH = os.environ.get("NOPESAC_TAIL_PREFETCH", "1") != "0"


def transformer_tail(attn: torch.Tensor, src: torch.Tensor, W: dict, *, pre_norm: bool, skip_ffn: bool = False, pos=None,
                     want=("y",), proj_pos=None, proj=None, prefetch=None) -> dict:
    """The tail of a transformer layer + the input projections of the next attention in one launch (nopesac_transformer_tail_bf16).
    W: "wo", "bo", "ga", "bea" (first norm) and - unless skip_ffn - "w1", "b1", "w2", "b2", "gb", "beb" (second norm); fragment-major
    bf16 matrices, f32 vectors.  proj_pos / proj = (fragment-major weight, f32 bias or None, width): projections of the normalised
    result + pos / of the normalised result, returned as "proj_pos" / "proj" (bf16 [M, width]).  want: any of "y", "y16", "ypos16", "yn".
    prefetch = (tensors, workgroups): the weight tensors and the workgroup count of the NEXT tail launch - with few rows (one pair per
    call) extra workgroups of this launch read them into the L2s the next launch will run on (nopesac_transformer_tail_bf16_pf)."""
    _chk(attn, torch.bfloat16); _chk(src, torch.float32)
    M = src.shape[0]
    _require(attn.shape == (M, 256) and src.shape == (M, 256), "transformer_tail: attn / src [M, 256]")
    out = {k: torch.empty(M, 256, device=src.device, dtype=torch.float32 if k in ("y", "yn") else torch.bfloat16) for k in want}
    if pos is not None:
        _chk(pos, torch.float32)
    wa, ba, na = proj_pos if proj_pos is not None else (None, None, 0)
    wb, bb, nb = proj if proj is not None else (None, None, 0)
    if na:
        out["proj_pos"] = torch.empty(M, na, device=src.device, dtype=torch.bfloat16)
    if nb:
        out["proj"] = torch.empty(M, nb, device=src.device, dtype=torch.bfloat16)
    g = W.get
    nptr = nbytes = None
    n_next = next_wg = 0
    if prefetch is not None and TAIL_PREFETCH and M <= 64 * 32:
        tens = [t for t in prefetch[0] if t is not None][:8]
        if tens and prefetch[1] > 0:
            n_next, next_wg = len(tens), int(prefetch[1])
            nptr = (ctypes.c_void_p * n_next)(*[_p(t) for t in tens])
            nbytes = (ctypes.c_int64 * n_next)(*[t.numel() * t.element_size() for t in tens])
    rc = _L().nopesac_transformer_tail_bf16_pf(
        _p(attn), _p(src), _p(W["wo"]), _p(W["bo"]), _p(W["ga"]), _p(W["bea"]), _p(g("w1")), _p(g("b1")), _p(g("w2")), _p(g("b2")),
        _p(g("gb")), _p(g("beb")), _p(pos), 0 if pos is None else pos.shape[0], _p(out.get("y")), _p(out.get("y16")), _p(out.get("ypos16")),
        _p(out.get("yn")), int(pre_norm), int(skip_ffn), _p(wa), _p(ba), _p(out.get("proj_pos")), na, _p(wb), _p(bb), _p(out.get("proj")), nb, M,
        nptr, nbytes, n_next, next_wg, _stream())
    _lib.check(rc, "nopesac_transformer_tail_bf16_pf")
    return out


class PoseBranchTail:
    """Operands of nopesac_posenet_branch_tail_bf16, packed once: layers 1..5 of the two pose-net branches (ConvW objects with folded
    BatchNorm) as fragment-major bf16 weights + pointer tables (the ctypes arrays keep the tensors alive)."""

    def __init__(self, convs_trans, convs_rots):
        import ctypes
        self.keep = []
        ws, ss, bs = [], [], []
        for c in list(convs_trans) + list(convs_rots):
            _require(c.cout == 128 and c.cin == 128 and c.kh == 3 and c.kw == 3 and c.scale is not None and c.bias is not None,
                     "pose-net branch tail: 3x3, 128 -> 128 convs with folded BatchNorm")
            wf = mfma_fragment_major(c.w(torch.bfloat16).reshape(128, -1))
            self.keep += [wf, c.scale, c.bias]
            ws.append(wf.data_ptr()); ss.append(c.scale.data_ptr()); bs.append(c.bias.data_ptr())
        _require(len(ws) == 10, "pose-net branch tail: five layers per branch")
        arr = ctypes.c_void_p * 10
        self.w, self.s, self.b = arr(*ws), arr(*ss), arr(*bs)


def posenet_branch_tail(x_trans: torch.Tensor, x_rots: torch.Tensor, packed: PoseBranchTail):
    """Layers 1..5 of both pose-net branches in one launch.  x_*: layer-0 outputs [B,15,20,128] bf16 -> two [B,2,3,128] f32 tensors."""
    _chk(x_trans, torch.bfloat16); _chk(x_rots, torch.bfloat16)
    B, H, W, C = x_trans.shape
    _require(x_rots.shape == x_trans.shape, "pose-net branch tail: the two branches have the same shape")
    yt = torch.empty((B, 2, 3, 128), device=x_trans.device, dtype=torch.float32)
    yr = torch.empty_like(yt)
    _lib.check(_L().nopesac_posenet_branch_tail_bf16(_p(x_trans), _p(x_rots), packed.w, packed.s, packed.b, _p(yt), _p(yr), B, H, W, C, _stream()),
               "nopesac_posenet_branch_tail_bf16")
    return yt, yr


def conv3x3_c64(x: torch.Tensor, w: torch.Tensor, scale: torch.Tensor, bias: torch.Tensor, act: int = ACT_RELU) -> torch.Tensor:
    """bf16 3x3/s1/p1 conv 64 -> 64 + BN + act from an LDS halo tile (csrc/conv3x3_c64.hip).  x [B,H,W,64], w [64,3,3,64]."""
    _chk(x, torch.bfloat16); _chk(w, torch.bfloat16); _chk(scale, torch.float32); _chk(bias, torch.float32)
    B, H, W, C = x.shape
    _require(C == 64 and tuple(w.shape) == (64, 3, 3, 64), 'argument check failed: C == 64 and tuple(w.shape) == (64, 3, 3, 64)')
    y = torch.empty_like(x)
    _lib.check(_L().nopesac_conv3x3_c64_bf16(_p(x), _p(_frag_weights(w)), _p(scale), _p(bias), _p(y), B, H, W, act, _stream()),
               "nopesac_conv3x3_c64_bf16")
    return y
