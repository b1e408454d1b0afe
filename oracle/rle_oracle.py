"""TEST INFRASTRUCTURE ONLY - numpy restatement of the COCO mask RLE used by the reference's instance packaging.

The reference calls pycocotools.mask.encode / toBbox (meta_arch/siamese_planeTR.py:27,703-704,747-748).
pycocotools (pinned only as "pycocotools" in the reference's environment.yaml) is NOT under /root/reference and not
installed here, so this file restates its PUBLISHED algorithm (cocoapi common/maskApi.c: rleEncode, rleToString,
rleFrString, rleToBbox, rleDecode).  PARITY UNPINNED: there is no pycocotools build in this image to check against;
the encoder is pinned by (a) an independent decoder written from rleFrString/rleDecode (round trip) and (b) a few
hand-worked strings in tests/test_oracle_golden.py.
"""
from __future__ import annotations

import numpy as np


def run_lengths(mask: np.ndarray) -> list:
    """rleEncode: column-major scan, alternating runs starting with zeros."""
    flat = np.asarray(mask).astype(np.uint8).reshape(-1, order="F")
    flips = np.flatnonzero(np.diff(np.concatenate([[0], flat])) != 0)
    edges = np.concatenate([[0], flips, [flat.size]])
    return np.diff(edges).astype(np.int64).tolist()


def to_string(counts) -> bytes:
    """rleToString: 5 data bits per char, bit 5 = continuation, +48; counts[i>2] stored minus counts[i-2]."""
    out = bytearray()
    for i, c in enumerate(counts):
        x = int(c)
        if i > 2:
            x -= int(counts[i - 2])
        more = True
        while more:
            ch = x & 0x1F
            x >>= 5                                  # arithmetic shift, like C on a signed long
            more = (x != -1) if (ch & 0x10) else (x != 0)
            if more:
                ch |= 0x20
            out.append(ch + 48)
    return bytes(out)


def from_string(s: bytes) -> list:
    """rleFrString."""
    counts, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = s[p] - 48
            x |= (c & 0x1F) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    return counts


def encode(mask: np.ndarray) -> dict:
    """pycocotools.mask.encode for one HxW mask."""
    h, w = mask.shape
    return {"size": [int(h), int(w)], "counts": to_string(run_lengths(mask))}


def decode(rle: dict) -> np.ndarray:
    """pycocotools.mask.decode (rleDecode): bool [H,W]."""
    h, w = rle["size"]
    counts = from_string(rle["counts"])
    flat = np.zeros(h * w, np.uint8)
    pos, val = 0, 0
    for c in counts:
        flat[pos:pos + c] = val
        pos += c
        val ^= 1
    assert pos == h * w, "run lengths do not cover the image"
    return flat.reshape((h, w), order="F").astype(bool)


def to_bbox(rle: dict) -> np.ndarray:
    """rleToBbox: [x, y, w, h] float64, the tight box of the ones (zeros if < 2 runs)."""
    h, w = rle["size"]
    counts = from_string(rle["counts"])
    m = (len(counts) // 2) * 2
    if m == 0:
        return np.zeros(4)
    xs, ys, xe, ye, xp, cc = w, h, 0, 0, 0, 0
    for j in range(m):
        cc += counts[j]
        t = cc - (j % 2)
        y = t % h
        x = (t - y) // h
        if j % 2 == 0:
            xp = x
        elif xp < x:
            ys, ye = 0, h - 1
        xs, xe, ys, ye = min(xs, x), max(xe, x), min(ys, y), max(ye, y)
    return np.array([xs, ys, xe - xs + 1, ye - ys + 1], dtype=np.float64)
