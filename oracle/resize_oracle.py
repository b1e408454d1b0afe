"""TEST INFRASTRUCTURE ONLY - numpy restatement of cv2.resize(..., interpolation=INTER_LINEAR) for uint8 images, the call the
reference's ScanNet mapper makes (data/planercnn_transforms.py:314).  OpenCV is not under /root/reference and not installed
here: this follows the published algorithm (modules/imgproc/src/resize.cpp: 11-bit fixed-point HResizeLinear / VResizeLinear
with FixedPtCast).  PARITY UNPINNED (no cv2 in this image); pinned only by its invariants in tests/test_data_cpu.py."""
from __future__ import annotations

import numpy as np


def _coef(dst_n: int, src_n: int):
    scale = np.float32(src_n / dst_n)        # cv2 keeps the scale in double and the position in float
    d = np.arange(dst_n, dtype=np.float64)
    f = ((d + 0.5) * np.float64(scale) - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    lo, hi = s < 0, s >= src_n - 1
    f = np.where(lo | hi, np.float32(0), f)
    s = np.clip(s, 0, src_n - 1)
    i1 = np.minimum(s + 1, src_n - 1)
    c0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
    c1 = np.rint(f * np.float32(2048)).astype(np.int64)
    return s, i1, c0, c1


def resize_bilinear_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """img uint8 [H,W,C] -> uint8 [out_h,out_w,C]."""
    H, W, _ = img.shape
    x0, x1, a0, a1 = _coef(out_w, W)
    y0, y1, b0, b1 = _coef(out_h, H)
    s = img.astype(np.int64)
    rows = s[:, x0, :] * a0[None, :, None] + s[:, x1, :] * a1[None, :, None]          # [H, out_w, C]
    r0, r1 = rows[y0], rows[y1]
    v = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)
