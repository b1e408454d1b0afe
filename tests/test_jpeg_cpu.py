"""JPEG decode, CPU side: the oracle (oracle/jpeg_oracle.py, a restatement of libjpeg-turbo's default decompression) against the
fixtures Pillow decoded (tests/golden/jpeg/*.jpg, jpeg_decoded.npz - oracle/gen_jpeg_golden.py) and, where Pillow is importable,
against Pillow itself on freshly encoded images; the product's host-side parser (nopesac_amd/jpeg.py) against the oracle's."""
import glob
import io
import os

import numpy as np
import pytest

from oracle import jpeg_oracle as J

GOLD = os.path.join(os.path.dirname(__file__), "golden")
FILES = sorted(glob.glob(os.path.join(GOLD, "jpeg", "*.jpg")))


def _name(p):
    return os.path.splitext(os.path.basename(p))[0]


def test_fixture_set_is_complete():
    dec = np.load(os.path.join(GOLD, "jpeg_decoded.npz"))
    assert len(FILES) == 16 and sorted(dec.files) == sorted(_name(p) for p in FILES)


@pytest.mark.parametrize("path", [p for p in FILES if "unsupported" not in p and "scannet_like" not in p], ids=_name)
def test_oracle_matches_the_pillow_decoded_fixture(path):
    dec = np.load(os.path.join(GOLD, "jpeg_decoded.npz"))
    out = J.decode(open(path, "rb").read())
    ref = dec[_name(path)]
    assert out.shape == ref.shape and out.dtype == np.uint8 and np.array_equal(out, ref)


def test_oracle_and_parser_refuse_progressive_files():
    from nopesac_amd import jpeg
    data = open(os.path.join(GOLD, "jpeg", "unsupported_progressive_40x56.jpg"), "rb").read()
    with pytest.raises(J.Unsupported):
        J.decode(data)
    with pytest.raises(jpeg.JpegUnsupported):
        jpeg.parse(data)
    with pytest.raises(jpeg.JpegUnsupported):
        jpeg.parse(b"\x89PNG\r\n\x1a\n" + b"\0" * 32)
    with pytest.raises(jpeg.JpegUnsupported):
        jpeg.parse(data[:200])                            # truncated


@pytest.mark.parametrize("path", [p for p in FILES if "unsupported" not in p], ids=_name)
def test_product_parser_agrees_with_the_oracle_parser(path):
    """geometry, tables and restart intervals of nopesac_amd/jpeg.py (what the device kernels are fed) = the oracle's reading"""
    from nopesac_amd import jpeg
    data = open(path, "rb").read()
    a, b = jpeg.parse(data, fast=False), J.parse(data)
    g = J.geometry(b)
    # the library's one-pass scan preparation (the default) lays out the words the Python path does
    fast = jpeg.parse(data)
    w, offs, cnts = jpeg._words(a.intervals)
    assert fast.intervals is None and np.array_equal(fast.words, w) and fast.seg_off == offs and fast.seg_cnt == cnts
    assert fast.seg_bytes == [len(q) for q in a.intervals]
    assert (fast.width, fast.height, fast.dri, fast.mcux, fast.mcuy) == (a.width, a.height, a.dri, a.mcux, a.mcuy)
    assert (a.width, a.height, a.mcux, a.mcuy, a.dri) == (b["W"], b["H"], g["mcux"], g["mcuy"], b["dri"])
    assert [(c["h"], c["v"], c["bw"], c["bh"], c["dw"], c["dh"], c["td"], c["ta"]) for c in a.comps] == \
           [(c["h"], c["v"], c["bw"], c["bh"], c["dw"], c["dh"], c["td"], c["ta"]) for c in b["comps"]]
    assert a.intervals == b["intervals"] and len(a.intervals) == (1 if not a.dri else -(-a.mcux * a.mcuy // a.dri))
    for c, d in zip(a.comps, b["comps"]):
        assert np.array_equal(a.qt[c["tq"]].astype(np.int32), b["q"][d["tq"]])
    # the device Huffman layout decodes every 16-bit prefix like the oracle's full table
    for key, spec in a.huff.items():
        t = jpeg.huffman_table_bytes(spec)
        look = t[:1024].view(np.uint16)
        maxcode, valoff, vals = t[1024:1096].view(np.int32), t[1096:1168].view(np.int32), t[1168:1424]
        ol, osym = J.huff_lookup(*b["huff"][key])
        for w in range(0, 65536, 7):
            e = int(look[w >> 7])
            if e >> 8:
                ln, sym = e >> 8, e & 255
            else:
                ln, w20 = 10, w << 4                         # (the kernel peeks up to 17 bits: four more zero bits here)
                while (w20 >> (20 - ln)) > maxcode[ln]:
                    ln += 1
                if ln > 16:
                    assert ol[w] == 0
                    continue
                sym = int(vals[((w20 >> (20 - ln)) + valoff[ln]) & 255])
            assert (ln, sym) == (int(ol[w]), int(osym[w])), (key, w)


def test_oracle_matches_pillow_on_fresh_encodings():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(5)
    n = 0
    for (h, w) in [(8, 8), (1, 1), (2, 3), (17, 5), (5, 33), (33, 4), (50, 70)]:
        for sub in (0, 1, 2):
            for q in (25, 90):
                for extra in ({}, {"restart_marker_blocks": 3}, {"optimize": True}):
                    a = (rng.random((h, w, 3)) * 255).astype(np.uint8) if (h + w + q) % 2 else \
                        np.broadcast_to(np.linspace(0, 255, w).astype(np.uint8)[None, :, None], (h, w, 3)).copy()
                    b = io.BytesIO()
                    Image.fromarray(a).save(b, format="JPEG", quality=q, subsampling=sub, **extra)
                    ref = np.asarray(Image.open(io.BytesIO(b.getvalue())).convert("RGB"))
                    assert np.array_equal(J.decode(b.getvalue()), ref), (h, w, sub, q, extra)
                    n += 1
    assert n == 126
