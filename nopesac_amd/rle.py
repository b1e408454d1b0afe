"""COCO RLE `instances` packaging from the post-selection winner map (replaces pycocotools.mask.encode / toBbox in
meta_arch/siamese_planeTR.py:685-720, 741-766).

Device: nopesac_rle_labels + nopesac_rle_transitions (csrc/rle.hip) produce, for every (view, kept plane), the
positions where the plane's mask flips along the column-major scan; no [n,H,W] mask tensor is ever built.
nopesac_rle_compress_device (a workgroup per mask) turns the flips into the compressed counts strings and the [x, y, w, h]
boxes, still on the device; the host receives finished byte strings.  Host syncs: two size reads (positions buffer, byte buffer).
nopesac_rle_compress_host / _batch_host (same library, plain C) are the host forms of the same encoder.
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


def compress_batch(positions: np.ndarray, offsets: np.ndarray, counts: np.ndarray, H: int, W: int):
    """All masks of a batch in ONE library call: -> (list of counts bytes, bbox float64 [n,4])."""
    L = _lib.load()
    n = int(len(counts))
    positions = np.ascontiguousarray(positions, dtype=np.uint32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    cap = 6 * (int(counts.sum()) + n) + 16
    buf = np.empty(cap, np.uint8)
    out_off = np.empty(n + 1, np.int64)
    bbox = np.empty((max(n, 1), 4), np.float64)
    r = L.nopesac_rle_compress_batch_host(positions.ctypes.data if positions.size else None, offsets.ctypes.data, counts.ctypes.data, n, H, W,
                                          buf.ctypes.data, cap, out_off.ctypes.data, bbox.ctypes.data)
    if r < 0:
        _lib.check(int(r), "nopesac_rle_compress_batch_host")
    raw = buf.tobytes()
    return [raw[out_off[i]:out_off[i + 1]] for i in range(n)], bbox[:n]


def encode_views(winner: torch.Tensor, kept_idx: torch.Tensor, n_kept: torch.Tensor, flags: torch.Tensor, n_kept_host=None) -> List[List[dict]]:
    """Per view, per kept plane (query order): {"segmentation": {"size": [H,W], "counts": bytes}, "bbox": [x,y,w,h]}.
    Flip positions AND the compressed strings are produced on the device (csrc/rle.hip); the host receives the finished byte
    strings (about 2 bytes per run instead of 4 per flip position) in one copy and only slices them.
    n_kept_host: n_kept as a Python list, when the caller has it already."""
    V, H, W = winner.shape
    nq = kept_idx.shape[1]
    labels = ops.rle_labels(winner, kept_idx, n_kept, flags)
    counts = ops.rle_transitions(labels, n_kept, nq)
    c64 = counts.view(-1).to(torch.int64)
    ends = torch.cumsum(c64, 0)
    offsets = (ends - c64).contiguous()
    total = int(ends[-1].item())                                         # host sync (sizes the positions buffer)
    pos = torch.empty(max(total, 1), device=winner.device, dtype=torch.int32)
    if total:
        ops.rle_transitions(labels, n_kept, nq, offsets=offsets.view(V, nq), positions=pos)
    data, out_off, lens, bbox = ops.rle_compress(pos, offsets, counts.view(-1).contiguous(), H, W)
    need = {"data": data, "out_off": out_off, "lens": lens, "bbox": bbox}
    if n_kept_host is None:
        need["n_kept"] = n_kept
    h = ops.gather_to_host(need)
    raw = h["data"].numpy().tobytes()
    out_off, lens, bbox = h["out_off"].tolist(), h["lens"].tolist(), h["bbox"].tolist()
    n_list = n_kept_host if n_kept_host is not None else h["n_kept"].tolist()
    out = []
    for v in range(V):
        row = []
        for p in range(n_list[v]):
            k = v * nq + p
            row.append({"segmentation": {"size": [H, W], "counts": raw[out_off[k]:out_off[k] + lens[k]]}, "bbox": bbox[k]})
        out.append(row)
    return out


class PendingRLE:
    """encode_views split in two so that a caller with several batches in flight never issues device work when it FETCHES results
    (a kernel enqueued behind other batches' launches waits for them; measured 6.8 ms per 32-pair step at the drop-in boundary):
    the constructor enqueues everything on the current stream - labels, flip positions (worst-case sized buffer: at most 2 flips
    per pixel and view), compressed strings into a byte buffer of fixed capacity, and the fetch of its filled part (at most `host_cap`
    bytes) plus the offset / length / box tables into pinned host memory; `finish()` (after the stream has passed the fetch) only slices.
    Totals beyond `host_cap` cost one more copy, beyond `cap` the synchronous encode_views."""

    def __init__(self, winner, kept_idx, n_kept, flags, cap: int = 64 << 20, host_cap: int = 6 << 20, host=None):
        V, H, W = winner.shape
        nq = kept_idx.shape[1]
        self.args, self.shape, self.cap, self.host_cap = (winner, kept_idx, n_kept, flags), (V, H, W, nq), int(cap), int(min(host_cap, cap))
        labels = ops.rle_labels(winner, kept_idx, n_kept, flags)
        counts = ops.rle_transitions(labels, n_kept, nq)
        c64 = counts.view(-1).to(torch.int64)
        ends = torch.cumsum(c64, 0)
        offsets = (ends - c64).contiguous()
        pos = torch.empty(V * 2 * H * W, device=winner.device, dtype=torch.int32)       # upper bound of the flips; only the used part is touched
        ops.rle_transitions(labels, n_kept, nq, offsets=offsets.view(V, nq), positions=pos)
        self.data, out_off, lens, bbox, total = ops.rle_compress_capped(pos, offsets, counts.view(-1).contiguous(), H, W, self.cap)
        # only the bytes the strings fill travel (one pair: ~20 KB of the 6 MB window - 0.11 ms of PCIe time per call before)
        # (finish() copies everything it hands out: the fetch may sit in a captured graph and be rewritten by the next replay)
        self.fetch = ops.HostFetch({"head": self.data[:self.host_cap], "out_off": out_off, "lens": lens, "bbox": bbox},
                                   dynamic={"head": total}, host=host)

    def finish(self, n_kept_host) -> List[List[dict]]:
        """Only after the stream the constructor ran on has passed the fetch (event / synchronize)."""
        V, H, W, nq = self.shape
        h = self.fetch.views()
        out_off, lens, bbox = h["out_off"].tolist(), h["lens"].tolist(), h["bbox"].tolist()
        if not out_off:                                        # no views / no plane slots: nothing was encoded
            return [[] for _ in range(V)]
        total = out_off[-1] + lens[-1]
        if total > self.cap:                                   # (never seen: 64 MB of run-length strings in one batch)
            return encode_views(*self.args, n_kept_host=n_kept_host)
        raw = h["head"].numpy()[:min(total, self.host_cap)].tobytes()
        if total > self.host_cap:
            raw += self.data[self.host_cap:total].cpu().numpy().tobytes()
        out = []
        for v in range(V):
            row = []
            for p in range(n_kept_host[v]):
                k = v * nq + p
                row.append({"segmentation": {"size": [H, W], "counts": raw[out_off[k]:out_off[k] + lens[k]]}, "bbox": bbox[k]})
            out.append(row)
        return out


# ---- reading side (evaluation): what pycocotools.mask.iou does for the matching evaluator (mp3d_evaluation.py:806-812)
def counts_of(rle: dict) -> np.ndarray:
    """Run lengths of a COCO RLE dict: `counts` is either the list of an uncompressed RLE or the compressed bytes / str
    (cocoapi rleFrString: 5 data bits per character + continuation bit, runs > 2 stored as differences)."""
    c = rle["counts"]
    if not isinstance(c, (bytes, str)):
        return np.asarray(c, dtype=np.int64)
    if isinstance(c, str):
        c = c.encode("ascii")
    out, p = [], 0
    while p < len(c):
        x, k, more = 0, 0, True
        while more:
            ch = c[p] - 48
            x |= (ch & 0x1F) << (5 * k)
            more = bool(ch & 0x20)
            p += 1
            k += 1
            if not more and (ch & 0x10):
                x |= -1 << (5 * k)
        if len(out) > 2:
            x += out[-2]
        out.append(x)
    return np.asarray(out, dtype=np.int64)


def decode(rle: dict) -> np.ndarray:
    """bool [H,W] mask of a COCO RLE (column-major runs, starting with zeros)."""
    h, w = rle["size"]
    counts = counts_of(rle)
    assert int(counts.sum()) == h * w, "RLE run lengths do not cover the image"
    vals = (np.arange(len(counts)) & 1).astype(np.uint8)
    return np.repeat(vals, counts).reshape((h, w), order="F").astype(bool)


def iou(dt: List[dict], gt: List[dict], iscrowd=None) -> np.ndarray:
    """[len(dt), len(gt)] float64 mask IoU (intersection / union; iscrowd[j] truthy: intersection / area(dt), as cocoapi rleIou)."""
    if len(dt) == 0 or len(gt) == 0:
        return np.zeros((len(dt), len(gt)), np.float64)
    D = np.stack([decode(r).reshape(-1) for r in dt]).astype(np.float64)
    G = np.stack([decode(r).reshape(-1) for r in gt]).astype(np.float64)
    inter = D @ G.T
    ad, ag = D.sum(1)[:, None], G.sum(1)[None, :]
    crowd = np.zeros(len(gt), bool) if iscrowd is None else np.asarray(iscrowd, bool)
    union = np.where(crowd[None, :], ad, ad + ag - inter)
    return np.where(union > 0, inter / np.maximum(union, 1e-300), 0.0)
