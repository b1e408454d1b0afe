"""TEST INFRASTRUCTURE (CPU oracle; imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg).

Baseline-JPEG decode as the reference's image reader performs it: detectron2 `utils.read_image` -> PIL `Image.open(...).convert("RGB")`
(NopeSAC_Net/data/planercnn_transforms.py:210-227, :306-314), i.e. Pillow's bundled libjpeg-turbo (here Pillow 12.2.0 / libjpeg-turbo,
JPEG_LIB_VERSION 62) with its DEFAULT decompression parameters: dct_method JDCT_ISLOW, do_fancy_upsampling TRUE, out_color_space RGB.
libjpeg-turbo is a third-party dependency that is not vendored under /root/reference; this file restates its published algorithms
(ITU T.81 Huffman decoding as in jdhuff.c, the 13-bit fixed-point "islow" inverse DCT of jidctint.c, the triangle-filter "fancy"
chroma upsampling of jdsample.c incl. its edge rules, the 16-bit fixed-point YCbCr -> RGB tables of jdcolor.c) in numpy / plain
Python loops.  PINNED: tests/test_jpeg_cpu.py checks it bit for bit against Pillow (the reference's own decoder, importable in this
container) on the committed fixtures tests/golden/jpeg/*.jpg and their Pillow-decoded arrays (tests/golden/jpeg_decoded.npz, written
by oracle/gen_jpeg_golden.py), and - when Pillow is importable - on freshly encoded random images.

Supported (= what the HIP decoder supports): baseline / extended sequential Huffman (SOF0 / SOF1), 8-bit samples, one scan holding
all components, 1 component (gray) or 3 components (YCbCr) with luma sampling 1x1, 2x1 or 2x2 and 1x1 chroma, optional restart
intervals.  Anything else raises Unsupported."""
from __future__ import annotations

import numpy as np

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55,
                   62, 63], dtype=np.int32)          # jpeg_natural_order (jutils.c): zigzag position k -> natural (row-major) index


class Unsupported(ValueError):
    pass


def parse(data: bytes) -> dict:
    """Markers of a JFIF / EXIF baseline file (T.81 Annex B) -> geometry, tables and the entropy-coded segment split at restart
    markers with the byte stuffing removed."""
    if data[:2] != b"\xff\xd8":
        raise Unsupported("not a JPEG (no SOI)")
    p, n = 2, len(data)
    q, huff, frame, dri, adobe = {}, {}, None, 0, None
    while p < n:
        if data[p] != 0xFF:
            raise Unsupported("marker expected at %d" % p)
        while p < n and data[p] == 0xFF:
            p += 1
        m = data[p]
        p += 1
        if m == 0xD8 or 0xD0 <= m <= 0xD7 or m == 0x01:
            continue
        if m == 0xD9:
            break
        L = (data[p] << 8) | data[p + 1]
        seg = data[p + 2:p + L]
        if m == 0xDB:                                     # DQT
            s = 0
            while s < len(seg):
                pq, tq = seg[s] >> 4, seg[s] & 15
                s += 1
                if pq:
                    t = np.frombuffer(seg[s:s + 128], dtype=">u2").astype(np.int32)
                    s += 128
                else:
                    t = np.frombuffer(seg[s:s + 64], dtype=np.uint8).astype(np.int32)
                    s += 64
                nat = np.zeros(64, np.int32)
                nat[ZIGZAG] = t
                q[tq] = nat
        elif m == 0xC4:                                   # DHT
            s = 0
            while s < len(seg):
                tc, th = seg[s] >> 4, seg[s] & 15
                bits = np.frombuffer(seg[s + 1:s + 17], dtype=np.uint8).astype(np.int32)
                cnt = int(bits.sum())
                vals = np.frombuffer(seg[s + 17:s + 17 + cnt], dtype=np.uint8).astype(np.int32)
                huff[(tc, th)] = (bits, vals)
                s += 17 + cnt
        elif m in (0xC0, 0xC1):                           # SOF0 / SOF1
            if seg[0] != 8:
                raise Unsupported("sample precision %d" % seg[0])
            H, W, nc = (seg[1] << 8) | seg[2], (seg[3] << 8) | seg[4], seg[5]
            comps = [{"id": seg[6 + 3 * i], "h": seg[7 + 3 * i] >> 4, "v": seg[7 + 3 * i] & 15, "tq": seg[8 + 3 * i]} for i in range(nc)]
            frame = {"H": H, "W": W, "comps": comps}
        elif m in (0xC2, 0xC3, 0xC5, 0xC6, 0xC7, 0xC9, 0xCA, 0xCB, 0xCD, 0xCE, 0xCF):
            raise Unsupported("SOF marker 0x%02X (progressive / lossless / arithmetic)" % m)
        elif m == 0xDD:
            dri = (seg[0] << 8) | seg[1]
        elif m == 0xEE and seg[:5] == b"Adobe":
            adobe = seg[11]
        elif m == 0xDA:                                   # SOS: the scan follows
            if frame is None:
                raise Unsupported("SOS before SOF")
            ns = seg[0]
            if ns != len(frame["comps"]):
                raise Unsupported("more than one scan (%d of %d components)" % (ns, len(frame["comps"])))
            for i in range(ns):
                c = next((c for c in frame["comps"] if c["id"] == seg[1 + 2 * i]), None)
                if c is None:
                    raise Unsupported("scan component id")
                c["td"], c["ta"] = seg[2 + 2 * i] >> 4, seg[2 + 2 * i] & 15
            if (seg[1 + 2 * ns], seg[2 + 2 * ns], seg[3 + 2 * ns]) != (0, 63, 0):
                raise Unsupported("spectral selection / successive approximation")
            p += L
            e = p
            while True:                                   # end of the entropy-coded segment: FF followed by neither 00 nor RSTn
                e = data.find(b"\xff", e)
                if e < 0 or e + 1 >= n:
                    raise Unsupported("no EOI")
                if data[e + 1] == 0 or 0xD0 <= data[e + 1] <= 0xD7:
                    e += 2
                    continue
                break
            ecs = data[p:e]
            intervals, s = [], 0
            while True:                                   # split at RSTn (after stuffing, FF Dn can only be a marker)
                k = s
                while True:
                    k = ecs.find(b"\xff", k)
                    if k < 0 or (k + 1 < len(ecs) and 0xD0 <= ecs[k + 1] <= 0xD7):
                        break
                    k += 2
                part = ecs[s:k if k >= 0 else len(ecs)]
                intervals.append(part.replace(b"\xff\x00", b"\xff"))
                if k < 0:
                    break
                s = k + 2
            break
        p += L
    else:
        raise Unsupported("no SOS")
    if frame is None or not intervals:
        raise Unsupported("no frame / scan")
    comps = frame["comps"]
    if len(comps) == 1:
        comps[0]["h"] = comps[0]["v"] = 1                 # a single-component scan is never interleaved (T.81 A.2.2)
    elif len(comps) == 3:
        if adobe is not None and adobe != 1:
            raise Unsupported("Adobe transform %d (RGB / CMYK data)" % adobe)
        if (comps[1]["h"], comps[1]["v"], comps[2]["h"], comps[2]["v"]) != (1, 1, 1, 1) or (comps[0]["h"], comps[0]["v"]) not in ((1, 1), (2, 1), (2, 2)):
            raise Unsupported("sampling factors %r" % [(c["h"], c["v"]) for c in comps])
    else:
        raise Unsupported("%d components" % len(comps))
    for c in comps:
        if c["tq"] not in q or (0, c["td"]) not in huff or (1, c["ta"]) not in huff:
            raise Unsupported("missing table")
    frame.update(q=q, huff=huff, dri=dri, intervals=intervals)
    return frame


def huff_lookup(bits, vals):
    """(code length, symbol) for every 16-bit prefix (T.81 Annex C canonical codes); length 0 = invalid prefix."""
    look_len, look_sym = np.zeros(65536, np.uint8), np.zeros(65536, np.uint8)
    code, k = 0, 0
    for ln in range(1, 17):
        for _ in range(int(bits[ln - 1])):
            lo = code << (16 - ln)
            look_len[lo:lo + (1 << (16 - ln))] = ln
            look_sym[lo:lo + (1 << (16 - ln))] = vals[k]
            code += 1
            k += 1
        code <<= 1
    return look_len, look_sym


class _Bits:
    def __init__(self, data: bytes):
        self.d, self.p, self.acc, self.n = data, 0, 0, 0

    def peek16(self) -> int:
        while self.n < 16:
            b = self.d[self.p] if self.p < len(self.d) else 0      # libjpeg pads an exhausted segment with zero bits (after a warning)
            self.p += 1
            self.acc = ((self.acc << 8) | b) & 0xFFFFFFFFFF
            self.n += 8
        return (self.acc >> (self.n - 16)) & 0xFFFF

    def skip(self, k: int):
        self.n -= k

    def get(self, k: int) -> int:
        if k == 0:
            return 0
        self.peek16()
        v = (self.acc >> (self.n - k)) & ((1 << k) - 1)
        self.n -= k
        return v


def geometry(fr: dict) -> dict:
    comps = fr["comps"]
    hmax, vmax = max(c["h"] for c in comps), max(c["v"] for c in comps)
    mcux, mcuy = -(-fr["W"] // (8 * hmax)), -(-fr["H"] // (8 * vmax))
    for c in comps:
        c["bw"], c["bh"] = mcux * c["h"], mcuy * c["v"]                  # blocks per row / column of the component plane
        c["dw"], c["dh"] = -(-fr["W"] * c["h"] // hmax), -(-fr["H"] * c["v"] // vmax)      # downsampled_width / height
    return {"hmax": hmax, "vmax": vmax, "mcux": mcux, "mcuy": mcuy}


def decode_coefficients(fr: dict):
    """-> per component int32 [bh, bw, 64] quantised coefficients in natural order (jdhuff.c decode_mcu_slow semantics)."""
    g = geometry(fr)
    comps = fr["comps"]
    look = {k: huff_lookup(*v) for k, v in fr["huff"].items()}
    coef = [np.zeros((c["bh"], c["bw"], 64), np.int32) for c in comps]
    n_mcu = g["mcux"] * g["mcuy"]
    per = fr["dri"] if fr["dri"] else n_mcu
    mcu = 0
    for part in fr["intervals"]:
        br = _Bits(part)
        pred = [0] * len(comps)
        for _ in range(min(per, n_mcu - mcu)):
            my, mx = divmod(mcu, g["mcux"])
            for ci, c in enumerate(comps):
                dl, ds = look[(0, c["td"])]
                al, as_ = look[(1, c["ta"])]
                for v in range(c["v"]):
                    for h in range(c["h"]):
                        blk = coef[ci][my * c["v"] + v, mx * c["h"] + h]
                        w = br.peek16()
                        br.skip(int(dl[w]))
                        s = int(ds[w])
                        d = br.get(s)
                        if s and d < (1 << (s - 1)):
                            d -= (1 << s) - 1
                        pred[ci] += d
                        blk[0] = pred[ci]
                        k = 1
                        while k < 64:
                            w = br.peek16()
                            br.skip(int(al[w]))
                            rs = int(as_[w])
                            r, s = rs >> 4, rs & 15
                            if s:
                                k += r
                                d = br.get(s)
                                if d < (1 << (s - 1)):
                                    d -= (1 << s) - 1
                                blk[ZIGZAG[k & 63]] = d
                                k += 1
                            elif r == 15:
                                k += 16
                            else:
                                break
            mcu += 1
    return coef


_F = dict(c0298=2446, c0390=3196, c0541=4433, c0765=6270, c0899=7373, c1175=9633, c1501=12299, c1847=15137, c1961=16069, c2053=16819,
          c2562=20995, c3072=25172)


def _idct_1d(x, shift_in, descale):
    """one pass of jidctint.c jpeg_idct_islow over the LAST axis of x [..., 8] (int64); even part inputs scaled by 1 << 13."""
    F = _F
    z2, z3 = x[..., 2], x[..., 6]
    z1 = (z2 + z3) * F["c0541"]
    tmp2 = z1 - z3 * F["c1847"]
    tmp3 = z1 + z2 * F["c0765"]
    z2, z3 = x[..., 0], x[..., 4]
    tmp0 = (z2 + z3) << 13
    tmp1 = (z2 - z3) << 13
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    tmp0, tmp1, tmp2, tmp3 = x[..., 7], x[..., 5], x[..., 3], x[..., 1]
    z1, z2, z3, z4 = tmp0 + tmp3, tmp1 + tmp2, tmp0 + tmp2, tmp1 + tmp3
    z5 = (z3 + z4) * F["c1175"]
    tmp0, tmp1, tmp2, tmp3 = tmp0 * F["c0298"], tmp1 * F["c2053"], tmp2 * F["c3072"], tmp3 * F["c1501"]
    z1, z2, z3, z4 = -z1 * F["c0899"], -z2 * F["c2562"], -z3 * F["c1961"] + z5, -z4 * F["c0390"] + z5
    tmp0, tmp1, tmp2, tmp3 = tmp0 + z1 + z3, tmp1 + z2 + z4, tmp2 + z2 + z3, tmp3 + z1 + z4
    out = np.stack([tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0, tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3], -1)
    return (out + (1 << (descale - 1))) >> descale


def idct_islow(coef: np.ndarray, quant: np.ndarray) -> np.ndarray:
    """coef int [..., 64] (natural order), quant int [64] -> uint8 samples [..., 8, 8] (jidctint.c: CONST_BITS 13, PASS1_BITS 2)."""
    x = (coef.astype(np.int64) * quant.astype(np.int64)).reshape(coef.shape[:-1] + (8, 8))
    ws = _idct_1d(np.swapaxes(x, -1, -2), 0, 13 - 2)            # pass 1: columns
    ws = np.swapaxes(ws, -1, -2)
    y = _idct_1d(ws, 0, 13 + 2 + 3)                              # pass 2: rows
    y = y & 1023                                                 # range_limit[... & RANGE_MASK] with the table centred on 128
    y = np.where(y >= 512, y - 1024, y)
    return np.clip(y + 128, 0, 255).astype(np.uint8)


def planes(fr: dict, coef) -> list:
    out = []
    for c, cf in zip(fr["comps"], coef):
        s = idct_islow(cf, fr["q"][c["tq"]])                     # [bh, bw, 8, 8]
        out.append(s.transpose(0, 2, 1, 3).reshape(c["bh"] * 8, c["bw"] * 8))
    return out


def upsample_h2v1(pl: np.ndarray, dw: int, dh: int) -> np.ndarray:
    """jdsample.c h2v1_fancy_upsample (plain replication when downsampled_width <= 2)."""
    x = pl[:dh, :dw].astype(np.int32)
    if dw <= 2:
        return np.repeat(x, 2, 1).astype(np.uint8)
    left = np.concatenate([x[:, :1], x[:, :-1]], 1)
    right = np.concatenate([x[:, 1:], x[:, -1:]], 1)
    even = (3 * x + left + 1) >> 2
    odd = (3 * x + right + 2) >> 2
    even[:, 0], odd[:, -1] = x[:, 0], x[:, -1]
    out = np.empty((dh, 2 * dw), np.int32)
    out[:, 0::2], out[:, 1::2] = even, odd
    return out.astype(np.uint8)


def upsample_h2v2(pl: np.ndarray, dw: int, dh: int) -> np.ndarray:
    """jdsample.c h2v2_fancy_upsample with jdmainct.c's context rows (the rows above the first / below the last real row are copies
    of that row); plain 2x2 replication when downsampled_width <= 2."""
    x = pl[:dh, :dw].astype(np.int32)
    if dw <= 2:
        return np.repeat(np.repeat(x, 2, 0), 2, 1).astype(np.uint8)
    above = np.concatenate([x[:1], x[:-1]], 0)
    below = np.concatenate([x[1:], x[-1:]], 0)
    out = np.empty((2 * dh, 2 * dw), np.int32)
    for v, other in ((0, above), (1, below)):
        cs = 3 * x + other                                        # thiscolsum
        last = np.concatenate([cs[:, :1], cs[:, :-1]], 1)
        nxt = np.concatenate([cs[:, 1:], cs[:, -1:]], 1)
        even = (3 * cs + last + 8) >> 4
        odd = (3 * cs + nxt + 7) >> 4
        even[:, 0] = (4 * cs[:, 0] + 8) >> 4
        odd[:, -1] = (4 * cs[:, -1] + 7) >> 4
        out[v::2, 0::2], out[v::2, 1::2] = even, odd
    return out.astype(np.uint8)


def ycc_to_rgb(y, cb, cr) -> np.ndarray:
    """jdcolor.c build_ycc_rgb_table / ycc_rgb_convert (SCALEBITS 16)."""
    y, cb, cr = y.astype(np.int32), cb.astype(np.int32) - 128, cr.astype(np.int32) - 128
    r = y + ((91881 * cr + 32768) >> 16)
    g = y + ((-22554 * cb + 32768 - 46802 * cr) >> 16)
    b = y + ((116130 * cb + 32768) >> 16)
    return np.clip(np.stack([r, g, b], -1), 0, 255).astype(np.uint8)


def decode(data: bytes) -> np.ndarray:
    """-> uint8 [H, W, 3] RGB, what PIL's Image.open(...).convert("RGB") returns for the file."""
    fr = parse(data)
    coef = decode_coefficients(fr)
    return reconstruct(fr, coef)


def reconstruct(fr: dict, coef) -> np.ndarray:
    pl = planes(fr, coef)
    H, W, comps = fr["H"], fr["W"], fr["comps"]
    if len(comps) == 1:
        yy = pl[0][:H, :W]
        return np.stack([yy, yy, yy], -1)
    hv = (comps[0]["h"], comps[0]["v"])
    yy = pl[0][:H, :W]
    ch = []
    for c, p in zip(comps[1:], pl[1:]):
        if hv == (1, 1):
            u = p
        elif hv == (2, 1):
            u = upsample_h2v1(p, c["dw"], c["dh"])
        else:
            u = upsample_h2v2(p, c["dw"], c["dh"])
        ch.append(u[:H, :W])
    return ycc_to_rgb(yy, ch[0], ch[1])
