"""GPU decode of baseline JPEG files (the ScanNet colour frames of the reference's data mapper: detectron2 `utils.read_image` = PIL /
libjpeg-turbo decode, NopeSAC_Net/data/planercnn_transforms.py:210-227 and :306-314).

Host side (this file): marker walk, restart-interval split, removal of the byte stuffing, Huffman lookup tables - a few numpy calls
per file, no entropy decoding.  Device side (csrc/jpeg.hip, include/nopesac_hip.h `nopesac_jpeg_*`): Huffman decode (restart-free
files: self-synchronising, one lane per 2048-bit subsequence; files with restart markers and streams the lanes do not settle on: one
wave per restart interval on the scalar unit), dequantisation + libjpeg's 13-bit "islow" inverse DCT,
"fancy" (triangle) chroma upsampling and the 16-bit fixed-point YCbCr -> RGB conversion, bit for bit what libjpeg-turbo produces with
its default decompression parameters (tests/test_jpeg_gpu.py against Pillow-decoded fixtures and oracle/jpeg_oracle.py).

Supported: SOF0 / SOF1 (baseline / extended sequential Huffman), 8-bit, ONE scan with all components, gray or YCbCr with luma sampling
1x1 / 2x1 / 2x2 and 1x1 chroma.  Everything else (progressive, arithmetic, CMYK, multi-scan) raises JpegUnsupported: the caller
(data.PairMapper.decode_files) then decodes that file with PIL on the host, as the reference does for every file."""
from __future__ import annotations

import ctypes
import os
import re
import struct
import threading
from typing import List, Sequence

import numpy as np
import torch

from . import _lib

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55,
                   62, 63], dtype=np.int64)

# layout constants shared with csrc/jpeg.hip (include/nopesac_hip.h NOPESAC_JPEG_*)
LOOK_BITS = 9
HUFF_BYTES = 1536                 # one Huffman table: look u16[512] | maxcode i32[18] | valoffset i32[18] | huffval u8[256] | pad
TABLES_BYTES = 4 * HUFF_BYTES + 3 * 128          # DC0, DC1, AC0, AC1, then three quantisation tables u16[64] (natural order)
IMG_I32, IMG_I64, SEG_I32, SEG_I64 = 32, 8, 4, 2
SUB_WORDS, SYNC_PASSES = 64, 12                # NOPESAC_JPEG_SUB_WORDS / NOPESAC_JPEG_SYNC_PASSES
PARALLEL_MIN_BYTES = 4 * SUB_WORDS * 4         # shorter restart-free streams stay on the one-wave-per-interval kernel


_SCAN_END = re.compile(rb"\xff[^\x00\xd0-\xd7\xff]")        # (FF FF: fill bytes in front of a marker - the second FF decides)
_RST = re.compile(rb"\xff[\xd0-\xd7]")


class JpegUnsupported(ValueError):
    """The file is not a baseline JPEG this decoder handles."""


class JpegInfo:
    __slots__ = ("width", "height", "comps", "qt", "huff", "dri", "intervals", "hmax", "vmax", "mcux", "mcuy", "words", "seg_off", "seg_cnt",
                 "seg_bytes")


def _u16(b, p):
    return (b[p] << 8) | b[p + 1]


def parse(data: bytes, fast: bool = True) -> JpegInfo:
    """parse_markers() with every failure a JpegUnsupported: truncated or malformed files (a marker past the end, an empty segment,
    a short DQT / DHT, component ids a scan never names) surface from the marker walk as IndexError / KeyError / ValueError /
    struct.error - the callers (data.PairMapper, LazyPairs' reader threads) promise to fall back to PIL on JpegUnsupported only."""
    try:
        return parse_markers(data, fast)
    except JpegUnsupported:
        raise
    except (IndexError, KeyError, ValueError, TypeError, OverflowError, struct.error) as e:
        raise JpegUnsupported("malformed file: %s: %s" % (type(e).__name__, e)) from e


def parse_markers(data: bytes, fast: bool = True) -> JpegInfo:
    """Marker segments (ITU T.81 Annex B) of one file -> geometry, tables and the entropy-coded data split at the restart markers with
    the stuffed zero bytes removed.  fast (default): the pass over the entropy-coded bytes runs in the library
    (`nopesac_jpeg_prepare_scan`, host C++, interpreter lock released) and leaves the device word layout in info.words / seg_off /
    seg_cnt / seg_bytes; fast = False: the same in Python (info.intervals = the unstuffed bytes of every interval)."""
    if len(data) < 4 or data[0] != 0xFF or data[1] != 0xD8:
        raise JpegUnsupported("no SOI marker")
    info = JpegInfo()
    info.qt, info.huff, info.dri, info.comps, info.intervals, info.words = {}, {}, 0, None, None, None
    adobe, jfif = None, False
    p, n = 2, len(data)
    while p + 4 <= n:
        if data[p] != 0xFF:
            raise JpegUnsupported("marker expected at byte %d" % p)
        while p < n and data[p] == 0xFF:
            p += 1
        m = data[p]
        p += 1
        if m in (0xD8, 0x01) or 0xD0 <= m <= 0xD7:
            continue
        if m == 0xD9:
            break
        L = _u16(data, p)
        seg = data[p + 2:p + L]
        if m == 0xDB:
            s = 0
            while s < len(seg):
                pq, tq = seg[s] >> 4, seg[s] & 15
                s += 1
                if pq:
                    t = np.frombuffer(seg, dtype=">u2", count=64, offset=s).astype(np.uint16)
                    s += 128
                else:
                    t = np.frombuffer(seg, dtype=np.uint8, count=64, offset=s).astype(np.uint16)
                    s += 64
                nat = np.zeros(64, np.uint16)
                nat[ZIGZAG] = t
                info.qt[tq] = nat
        elif m == 0xC4:
            s = 0
            while s < len(seg):
                tc, th = seg[s] >> 4, seg[s] & 15
                cnt = sum(seg[s + 1:s + 17])
                info.huff[(tc, th)] = bytes(seg[s + 1:s + 17 + cnt])
                s += 17 + cnt
        elif m in (0xC0, 0xC1):
            if seg[0] != 8:
                raise JpegUnsupported("%d-bit samples" % seg[0])
            info.height, info.width = _u16(seg, 1), _u16(seg, 3)
            info.comps = [{"id": seg[6 + 3 * i], "h": seg[7 + 3 * i] >> 4, "v": seg[7 + 3 * i] & 15, "tq": seg[8 + 3 * i]} for i in range(seg[5])]
        elif 0xC2 <= m <= 0xCF and m not in (0xC4, 0xC8, 0xCC):
            raise JpegUnsupported("SOF%d (progressive / lossless / arithmetic coding)" % (m - 0xC0))
        elif m == 0xDD:
            info.dri = _u16(seg, 0)
        elif m == 0xEE and bytes(seg[:5]) == b"Adobe" and len(seg) >= 12:
            adobe = seg[11]
        elif m == 0xE0 and bytes(seg[:5]) == b"JFIF\0":
            jfif = True
        elif m == 0xDA:
            if info.comps is None:
                raise JpegUnsupported("SOS before SOF")
            ns = seg[0]
            if ns != len(info.comps):
                raise JpegUnsupported("more than one scan")
            for i in range(ns):
                c = next((c for c in info.comps if c["id"] == seg[1 + 2 * i]), None)
                if c is None:
                    raise JpegUnsupported("scan names an unknown component")
                c["td"], c["ta"] = seg[2 + 2 * i] >> 4, seg[2 + 2 * i] & 15
            if (seg[1 + 2 * ns], seg[2 + 2 * ns], seg[3 + 2 * ns]) != (0, 63, 0):
                raise JpegUnsupported("spectral selection / successive approximation")
            p += L
            if fast:
                _prepare_scan(info, data, p)
                break
            # the scan ends at the first FF that is neither stuffing nor RSTn - normally the EOI that ends the file: take the last EOI
            # and check with three memchr-speed counts that every FF before it is stuffing / a restart marker (else: regex scan)
            e = data.rfind(b"\xff\xd9")
            ecs = data[p:e] if e >= p else b""
            cuts = [m_.start() for m_ in _RST.finditer(ecs)] if info.dri else []
            if e < p or ecs.count(b"\xff") != ecs.count(b"\xff\x00") + len(cuts):
                end = _SCAN_END.search(data, p)
                if end is None:
                    raise JpegUnsupported("truncated file (no EOI)")
                ecs = data[p:end.start()]
                cuts = [m_.start() for m_ in _RST.finditer(ecs)] if info.dri else []
            # (after stuffing, FF D0..D7 inside the scan can only be a restart marker)
            parts = [ecs[a:b] for a, b in zip([0] + [c + 2 for c in cuts], cuts + [len(ecs)])]
            info.intervals = [q.replace(b"\xff\x00", b"\xff") for q in parts]
            break
        p += L
    if info.comps is None or (info.intervals is None and info.words is None):
        raise JpegUnsupported("no frame / scan")
    comps = info.comps
    if len(comps) == 1:
        comps[0]["h"] = comps[0]["v"] = 1                 # a one-component scan is never interleaved (T.81 A.2.2)
    elif len(comps) == 3:
        if adobe is not None and adobe != 1:
            raise JpegUnsupported("Adobe colour transform %d" % adobe)
        if adobe is None and not jfif and [c["id"] for c in comps] == [82, 71, 66]:
            # jdapimin.c default_decompress_parms: no JFIF / Adobe marker and component ids 'R','G','B' -> the data IS RGB
            raise JpegUnsupported("RGB-coded file (component ids R, G, B without a JFIF / Adobe marker)")
        if (comps[1]["h"], comps[1]["v"], comps[2]["h"], comps[2]["v"]) != (1, 1, 1, 1) or (comps[0]["h"], comps[0]["v"]) not in ((1, 1), (2, 1), (2, 2)):
            raise JpegUnsupported("sampling factors %r" % [(c["h"], c["v"]) for c in comps])
    else:
        raise JpegUnsupported("%d components" % len(comps))
    for c in comps:
        if "td" not in c:
            raise JpegUnsupported("the scan does not name component %d" % c["id"])
        if c["tq"] not in info.qt or (0, c["td"]) not in info.huff or (1, c["ta"]) not in info.huff or c["td"] > 1 or c["ta"] > 1:
            raise JpegUnsupported("missing / out-of-range table")
    if info.width <= 0 or info.height <= 0:
        raise JpegUnsupported("empty image")
    info.hmax, info.vmax = comps[0]["h"], comps[0]["v"]
    info.mcux, info.mcuy = -(-info.width // (8 * info.hmax)), -(-info.height // (8 * info.vmax))
    n_mcu = info.mcux * info.mcuy
    per = info.dri if info.dri else n_mcu
    n_int = len(info.seg_off) if info.words is not None else len(info.intervals)
    if n_int != -(-n_mcu // per):
        raise JpegUnsupported("%d restart intervals for %d MCUs of %d" % (n_int, n_mcu, per))
    for c in comps:
        c["bw"], c["bh"] = info.mcux * c["h"], info.mcuy * c["v"]
        c["dw"], c["dh"] = -(-info.width * c["h"] // info.hmax), -(-info.height * c["v"] // info.vmax)
    return info


def _prepare_scan(info: JpegInfo, data: bytes, p: int):
    """fast path of parse(): nopesac_jpeg_prepare_scan over data[p:]"""
    n = len(data) - p
    cap_segs = (n // 2 + 2) if info.dri else 1            # (an interval is at least its two marker bytes)
    cap_segs = min(cap_segs, 1 << 20)
    words = np.empty(n // 4 + 5 * cap_segs + 8, np.uint32)
    so, sc, sb = (np.empty(cap_segs, np.int64) for _ in range(3))
    consumed = ctypes.c_int64(0)
    base = ctypes.cast(ctypes.c_char_p(data), ctypes.c_void_p).value
    k = _lib.load().nopesac_jpeg_prepare_scan(ctypes.c_void_p(base + p), n, 1 if info.dri else 0, words.ctypes.data, words.size, so.ctypes.data,
                                              sc.ctypes.data, sb.ctypes.data, cap_segs, ctypes.byref(consumed))
    if k < 1:
        raise JpegUnsupported("scan data could not be prepared")
    if p + consumed.value + 1 >= len(data):
        raise JpegUnsupported("truncated file (no EOI)")
    info.seg_off, info.seg_cnt, info.seg_bytes = so[:k].tolist(), sc[:k].tolist(), sb[:k].tolist()
    info.words = words[:info.seg_off[-1] + info.seg_cnt[-1]]


_HUFF_CACHE = {}
_HUFF_LOCK = threading.Lock()                             # parse() / decode_batch run on LazyPairs' reader threads


def huffman_table_bytes(spec: bytes) -> np.ndarray:
    """BITS[16] + HUFFVAL of a DHT segment -> the HUFF_BYTES device layout (T.81 Annex C canonical codes; the 9-bit look-ahead table
    and the maxcode / valoffset arrays are the ones jdhuff.c's jpeg_make_d_derived_tbl builds)."""
    with _HUFF_LOCK:
        t = _HUFF_CACHE.get(spec)
    if t is not None:
        return t
    bits, vals = list(spec[:16]), list(spec[16:])
    if sum(bits) != len(vals) or len(vals) > 256:
        raise JpegUnsupported("bad Huffman table")
    look = np.zeros(1 << LOOK_BITS, np.uint16)
    maxcode = np.full(18, -1, np.int32)
    valoff = np.zeros(18, np.int32)
    code, k = 0, 0
    for ln in range(1, 17):
        if bits[ln - 1]:
            valoff[ln] = k - code
            for _ in range(bits[ln - 1]):
                if ln <= LOOK_BITS:
                    lo = code << (LOOK_BITS - ln)
                    look[lo:lo + (1 << (LOOK_BITS - ln))] = (ln << 8) | vals[k]
                code += 1
                k += 1
            maxcode[ln] = code - 1
        if code > (1 << ln):
            raise JpegUnsupported("bad Huffman code lengths")
        code <<= 1
    maxcode[17] = 0x7FFFFFFF                              # sentinel: the slow path always ends
    out = np.zeros(HUFF_BYTES, np.uint8)
    out[:1024] = look.view(np.uint8)
    out[1024:1096] = maxcode.view(np.uint8)
    out[1096:1168] = valoff.view(np.uint8)
    out[1168:1168 + len(vals)] = np.asarray(vals, np.uint8)
    with _HUFF_LOCK:
        _HUFF_CACHE[spec] = out
        while len(_HUFF_CACHE) > 256:
            _HUFF_CACHE.pop(next(iter(_HUFF_CACHE)))
    return out


def _words(intervals: Sequence[bytes]):
    """the restart intervals of one image -> (32-bit words whose MOST significant bit is the first bit of the stream, every interval
    padded to whole words + 4 zero words: the decoder reads ahead, and libjpeg also feeds zero bits past the end of a segment;
    word offset of every interval; its word count)"""
    lens = [len(q) + (-len(q)) % 4 + 16 for q in intervals]
    if len(intervals) == 1:
        buf = intervals[0] + b"\0" * (lens[0] - len(intervals[0]))
    else:
        buf = b"".join(q + b"\0" * (n - len(q)) for q, n in zip(intervals, lens))
    w = np.frombuffer(buf, dtype=">u4").astype(np.uint32)
    offs, o = [], 0
    for n in lens:
        offs.append(o)
        o += n // 4
    return w, offs, [n // 4 for n in lens]


class HostBatch:
    """The host half of decode_batch for one batch of parsed files: every array the launch chain needs, built WITHOUT touching the device -
    data.LazyPairs' reader threads build it next to the file reads, so that the thread which launches the model only copies and launches
    (the per-image table filling and the concatenation of a batch's 13 MB of scan words were ~2 ms per batch of 32 ScanNet pairs on that
    thread)."""
    __slots__ = ("infos", "img32", "img64", "tables", "seg32", "seg64", "words", "lane_img", "n", "n_lanes", "n_blocks", "max_px", "coef_off", "plane_off",
                 "out_off", "n_seg")


# NOPESAC_JPEG_PINNED=1: the batch's host arrays in pinned memory (asynchronous H2D DMAs).  Off by default: every pinned block that is not in
# torch's cache yet is a hipHostMalloc, which waits for the whole device - with several batches in flight the first rounds of a run (and every
# short run: bench.py's jpeg_decode leg, three rounds) serialise on it (measured: 68 instead of 11 ms per round of four batches).
_PIN = os.environ.get("NOPESAC_JPEG_PINNED", "0") == "1"


def _pinned(arr: np.ndarray) -> torch.Tensor:
    t = torch.from_numpy(np.ascontiguousarray(arr))
    if _PIN and torch.cuda.is_available():
        p = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        p.copy_(t)
        return p
    return t


def prepare_batch(infos: Sequence[JpegInfo], parallel: bool = True) -> HostBatch:
    """parse() results of a batch's files -> HostBatch (no device work)."""
    n = len(infos)
    hb = HostBatch()
    hb.infos, hb.n = list(infos), n
    img32, img64 = np.zeros((n, IMG_I32), np.int32), np.zeros((n, IMG_I64), np.int64)
    tables = np.zeros((n, TABLES_BYTES), np.uint8)
    seg32, seg64, words, lane_img = [], [], [], []
    coef_off = plane_off = out_off = word_off = n_lanes = 0
    n_blocks, max_px = 0, 0
    for i, f in enumerate(infos):
        nc = len(f.comps)
        a = img32[i]
        a[0:8] = (f.width, f.height, nc, f.hmax, f.vmax, f.mcux, f.mcuy, f.dri if f.dri else f.mcux * f.mcuy)
        for ci, c in enumerate(f.comps):
            a[8 + ci], a[11 + ci], a[14 + ci], a[17 + ci] = c["bw"], c["bh"], c["dw"], c["dh"]
            a[20 + ci], a[23 + ci] = c["td"], 2 + c["ta"]
            img64[i, ci] = coef_off
            img64[i, 3 + ci] = plane_off
            coef_off += c["bw"] * c["bh"] * 64
            plane_off += c["bw"] * c["bh"] * 64
            tables[i, 4 * HUFF_BYTES + 128 * ci:4 * HUFF_BYTES + 128 * (ci + 1)] = f.qt[c["tq"]].view(np.uint8)
        a[26] = n_blocks                                   # first block of the image in the batch-wide block list (IDCT kernel)
        nb = sum(c["bw"] * c["bh"] for c in f.comps)
        a[27] = nb
        n_blocks += nb
        img64[i, 6] = out_off
        out_off += -(-f.width * f.height * 3 // 16) * 16   # (16-byte aligned images: the colour kernel stores dwords)
        max_px = max(max_px, f.width * f.height)
        for (tc, th), spec in f.huff.items():
            if th <= 1 and tc <= 1:
                tables[i, (2 * tc + th) * HUFF_BYTES:(2 * tc + th + 1) * HUFF_BYTES] = huffman_table_bytes(spec)
        per, n_mcu = int(a[7]), f.mcux * f.mcuy
        if f.words is not None:
            w, offs, cnts, first_bytes = f.words, f.seg_off, f.seg_cnt, f.seg_bytes[0]
        else:
            w, offs, cnts = _words(f.intervals)
            first_bytes = len(f.intervals[0])
        img64[i, 7] = word_off
        if parallel and not f.dri and first_bytes >= PARALLEL_MIN_BYTES:
            nsub = -(-first_bytes * 8 // (SUB_WORDS * 32))
            a[28], a[29] = n_lanes, nsub
            lanes = -(-nsub // 64) * 64
            lane_img.append(np.full(lanes, i, np.int32))
            n_lanes += lanes
        for k in range(len(offs)):
            seg32.append((i, k * per, min(per, n_mcu - k * per), 0))
            seg64.append((word_off + offs[k], cnts[k]))
        words.append(w)
        word_off += len(w)
    hb.img32, hb.img64, hb.tables = _pinned(img32), _pinned(img64), _pinned(tables)
    hb.seg32, hb.seg64 = _pinned(np.asarray(seg32, np.int32).reshape(-1, 4)), _pinned(np.asarray(seg64, np.int64).reshape(-1, 2))
    hb.n_seg = len(seg32)
    if _PIN and torch.cuda.is_available():                 # the words of every file straight into ONE pinned buffer
        wt = torch.empty(word_off, dtype=torch.int32, pin_memory=True)
        wn, o = wt.numpy().view(np.uint32), 0
        for w in words:
            wn[o:o + len(w)] = w
            o += len(w)
        hb.words = wt
    else:
        hb.words = torch.from_numpy(np.concatenate(words).view(np.int32)) if words else torch.zeros(0, dtype=torch.int32)
    hb.lane_img = _pinned(np.concatenate(lane_img)) if n_lanes else None
    hb.n_lanes, hb.n_blocks, hb.max_px, hb.coef_off, hb.plane_off, hb.out_off = n_lanes, n_blocks, max_px, coef_off, plane_off, out_off
    return hb


class _Geometry:
    """what decode_batch needs of a file a HostBatch was built for natively"""
    __slots__ = ("width", "height")

    def __init__(self, height, width):
        self.height, self.width = int(height), int(width)


def prepare_files(paths: Sequence[str], threads: int = 4, parallel: bool = True):
    """The HostBatch of a batch of JPEG FILES built by ONE library call on `threads` threads of its own (csrc/jpeg_host.hip: read, marker
    walk, checks, stuffing removal, tables, launch arrays - parse() + prepare_batch() without the interpreter).  Returns (HostBatch, None)
    or (None, status list) when a file is outside what that path takes (the caller falls back to parse() per file, which decides what
    PIL must decode)."""
    L = _lib.load()
    n = len(paths)
    if n == 0:
        return None, []
    arr = (ctypes.c_char_p * n)(*[os.fsencode(p) for p in paths])
    status = (ctypes.c_int * n)()
    totals = (ctypes.c_int64 * 8)()
    h = L.nopesac_jpeg_batch_scan_host(arr, n, max(1, int(threads)), 1 if parallel else 0, status, totals)
    if not h:
        return None, [-1] * n
    try:
        if any(status):
            return None, list(status)
        n_words, n_segs, n_lanes, n_blocks, max_px, coef_off, plane_off, out_off = [int(v) for v in totals]
        hb = HostBatch()
        hb.n = n
        hb.img32, hb.img64 = torch.empty((n, IMG_I32), dtype=torch.int32), torch.empty((n, IMG_I64), dtype=torch.int64)
        hb.tables = torch.empty((n, TABLES_BYTES), dtype=torch.uint8)
        hb.seg32, hb.seg64 = torch.empty((n_segs, 4), dtype=torch.int32), torch.empty((n_segs, 2), dtype=torch.int64)
        hb.words = torch.empty(n_words, dtype=torch.int32)
        hb.lane_img = torch.empty(n_lanes, dtype=torch.int32) if n_lanes else None
        geo = torch.empty((n, 2), dtype=torch.int32)
        rc = L.nopesac_jpeg_batch_fill_host(h, hb.img32.data_ptr(), hb.img64.data_ptr(), hb.tables.data_ptr(), hb.seg32.data_ptr(), hb.seg64.data_ptr(),
                                            hb.words.data_ptr(), hb.lane_img.data_ptr() if n_lanes else None, geo.data_ptr())
        if rc != 0:
            return None, [rc] * n
        hb.infos = [_Geometry(hh, ww) for hh, ww in geo.tolist()]
        hb.n_seg, hb.n_lanes, hb.n_blocks, hb.max_px = n_segs, n_lanes, n_blocks, max_px
        hb.coef_off, hb.plane_off, hb.out_off = coef_off, plane_off, out_off
        return hb, None
    finally:
        L.nopesac_jpeg_batch_free_host(h)


def decode_batch(files: Sequence[bytes], device, bgr: bool = False, infos: Sequence[JpegInfo] = None, parallel: bool = True,
                 stats: dict = None, _force_unsettled: bool = False, host: HostBatch = None) -> List[torch.Tensor]:
    """JPEG files -> uint8 [H, W, 3] device tensors (RGB, or BGR for INPUT.FORMAT "BGR"), one launch chain for the whole batch on the
    current stream.  Raises JpegUnsupported if any file is outside the supported subset (nothing is decoded then).
    parallel: restart-free streams go through the self-synchronising decoder (nopesac_jpeg_huffman_parallel; images it does not
    settle fall through to the one-wave-per-interval kernel on the device, no host involvement).  stats: receives the device tensors
    "par_done" (int32 per image) and "changed" (lanes that moved per pass and image) for diagnostics.  host: the batch's HostBatch when
    the caller prepared it already (prepare_batch, e.g. on a reader thread)."""
    if host is None:
        infos = list(infos) if infos is not None else [parse(f) for f in files]
        if len(infos) == 0:
            return []
        host = prepare_batch(infos, parallel)
    infos, n = host.infos, host.n
    if n == 0:
        return []
    n_lanes, n_blocks, max_px = host.n_lanes, host.n_blocks, host.max_px
    dev = torch.device(device)
    up = lambda t: t.to(dev, non_blocking=True)
    t_img32, t_img64, t_tab = up(host.img32), up(host.img64), up(host.tables)
    t_seg32, t_seg64 = up(host.seg32), up(host.seg64)
    t_words = up(host.words)
    img64 = host.img64
    coef = torch.zeros(host.coef_off, device=dev, dtype=torch.int16)
    planes = torch.empty(host.plane_off, device=dev, dtype=torch.uint8)
    out = torch.empty(host.out_off, device=dev, dtype=torch.uint8)
    L = _lib.load()
    st = torch.cuda.current_stream(dev).cuda_stream
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    par_done, work = None, []
    if n_lanes:
        t_lane = up(host.lane_img)
        exit_state = torch.empty(n_lanes, device=dev, dtype=torch.int64)
        entry_used = torch.full((n_lanes,), -1, device=dev, dtype=torch.int64)
        n_blk = torch.zeros(n_lanes, device=dev, dtype=torch.int32)
        first_block = torch.zeros(n_lanes, device=dev, dtype=torch.int64)
        changed = torch.zeros(SYNC_PASSES * n, device=dev, dtype=torch.int32)
        par_done = torch.zeros(n, device=dev, dtype=torch.int32)
        if _force_unsettled:                              # test hook: pretend lanes still moved in the last pass -> the serial kernel decodes
            changed[(SYNC_PASSES - 1) * n:] = 1
        _lib.check(L.nopesac_jpeg_huffman_parallel(p(t_img32), p(t_img64), p(t_tab), n, p(t_lane), n_lanes, p(t_words), int(t_words.numel()),
                                                   p(exit_state), p(entry_used), p(n_blk), p(first_block), p(changed), p(par_done), p(coef), st),
                   "nopesac_jpeg_huffman_parallel")
        work = [t_lane, exit_state, entry_used, n_blk, first_block, changed, par_done]
        if stats is not None:
            stats["par_done"], stats["changed"] = par_done, changed.view(SYNC_PASSES, n)
    _lib.check(L.nopesac_jpeg_huffman(p(t_img32), p(t_img64), p(t_tab), p(t_seg32), p(t_seg64), host.n_seg, p(t_words), int(t_words.numel()), p(coef),
                                      p(par_done) if par_done is not None else None, st), "nopesac_jpeg_huffman")
    _lib.check(L.nopesac_jpeg_idct(p(t_img32), p(t_img64), p(t_tab), n, n_blocks, p(coef), p(planes), st), "nopesac_jpeg_idct")
    _lib.check(L.nopesac_jpeg_color(p(t_img32), p(t_img64), n, max_px, p(planes), p(out), 1 if bgr else 0, st), "nopesac_jpeg_color")
    res = []
    for i, f in enumerate(infos):
        o = int(img64[i, 6])
        res.append(out[o:o + f.width * f.height * 3].view(f.height, f.width, 3))
    for t in [t_img32, t_img64, t_tab, t_seg32, t_seg64, t_words, coef, planes] + work:
        t.record_stream(torch.cuda.current_stream(dev))
    return res
