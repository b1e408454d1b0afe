"""COCO RLE `instances` packaging from the post-selection winner map (replaces pycocotools.mask.encode / toBbox in
meta_arch/siamese_planeTR.py:685-720, 741-766).

Device: nopesac_rle_labels + nopesac_rle_transitions (csrc/rle.hip) produce, for every (view, kept plane), the
positions where the plane's mask flips along the column-major scan; no [n,H,W] mask tensor is ever built.
Host: nopesac_rle_compress_host (same library, plain C) turns one plane's flips into the compressed counts string and
the [x, y, w, h] box.  The only host sync is the size read between the count and the fill pass.
"""
from __future__ import annotations

import ctypes
from typing import List

import numpy as np
import torch

from . import _lib, ops


def flip_positions(winner: torch.Tensor, kept_idx: torch.Tensor, n_kept: torch.Tensor, flags: torch.Tensor):
    """winner uint8 [V,H,W], kept_idx int32 [V,nq], n_kept int32 [V], flags int32 [V] (device) ->
    (counts int32 [V,nq] cpu, offsets int64 [V,nq] cpu, positions uint32 numpy [total])."""
    V, H, W = winner.shape
    nq = kept_idx.shape[1]
    labels = ops.rle_labels(winner, kept_idx, n_kept, flags)
    counts = ops.rle_transitions(labels, n_kept, nq)
    offsets = torch.cumsum(counts.view(-1).to(torch.int64), 0) - counts.view(-1).to(torch.int64)
    total = int(counts.sum().item())                                     # host sync (sizes the positions buffer)
    pos = torch.empty(max(total, 1), device=winner.device, dtype=torch.int32)
    if total:
        ops.rle_transitions(labels, n_kept, nq, offsets=offsets.view(V, nq), positions=pos)
    return counts.cpu(), offsets.view(V, nq).cpu(), pos[:total].cpu().numpy().view(np.uint32)


def compress(positions: np.ndarray, H: int, W: int):
    """One mask's ascending flip positions -> (counts bytes, bbox [x,y,w,h] list of float)."""
    L = _lib.load()
    positions = np.ascontiguousarray(positions, dtype=np.uint32)
    cap = 6 * (len(positions) + 1)
    buf = ctypes.create_string_buffer(cap)
    bbox = (ctypes.c_double * 4)()
    n = L.nopesac_rle_compress_host(positions.ctypes.data if len(positions) else None, len(positions), H, W,
                                    ctypes.cast(buf, ctypes.c_void_p), cap, ctypes.cast(bbox, ctypes.c_void_p))
    if n < 0:
        _lib.check(n, "nopesac_rle_compress_host")
    return buf.raw[:n], [float(b) for b in bbox]


def encode_views(winner: torch.Tensor, kept_idx: torch.Tensor, n_kept: torch.Tensor, flags: torch.Tensor) -> List[List[dict]]:
    """Per view, per kept plane (query order): {"segmentation": {"size": [H,W], "counts": bytes}, "bbox": [x,y,w,h]}."""
    V, H, W = winner.shape
    counts, offsets, pos = flip_positions(winner, kept_idx, n_kept, flags)
    n_list = n_kept.cpu().tolist()
    out = []
    for v in range(V):
        planes = []
        for p in range(n_list[v]):
            o, c = int(offsets[v, p]), int(counts[v, p])
            s, bbox = compress(pos[o:o + c], H, W)
            planes.append({"segmentation": {"size": [H, W], "counts": s}, "bbox": bbox})
        out.append(planes)
    return out
