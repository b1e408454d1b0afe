"""JPEG decode on the device (nopesac_amd/jpeg.py + csrc/jpeg.hip) against Pillow-decoded fixtures, the CPU oracle and - where Pillow is
importable - Pillow itself at the ScanNet frame size; bit-exact in every case."""
import glob
import io
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
FILES = sorted(p for p in glob.glob(os.path.join(GOLD, "jpeg", "*.jpg")) if "unsupported" not in p)


def _name(p):
    return os.path.splitext(os.path.basename(p))[0]


def test_fixtures_decode_bit_exact_in_one_batch(device):
    from nopesac_amd import jpeg
    dec = np.load(os.path.join(GOLD, "jpeg_decoded.npz"))
    files = [open(p, "rb").read() for p in FILES]
    for bgr in (False, True):
        outs = jpeg.decode_batch(files, device, bgr=bgr)
        torch.cuda.synchronize()
        assert len(outs) == len(FILES) == 15
        for p, o in zip(FILES, outs):
            ref = dec[_name(p)]
            got = o.cpu().numpy()
            assert got.dtype == np.uint8 and got.shape == ref.shape, _name(p)
            assert np.array_equal(got[..., ::-1] if bgr else got, ref), _name(p)


@pytest.mark.parametrize("path", FILES, ids=_name)
def test_single_file_matches_the_oracle_coefficients_and_pixels(device, path):
    """the same file alone (other offsets in every buffer) against the CPU oracle"""
    from nopesac_amd import jpeg
    from oracle import jpeg_oracle as J
    data = open(path, "rb").read()
    out = jpeg.decode_batch([data], device)[0].cpu().numpy()
    if "scannet_like" in path:                              # (the pure-Python oracle needs ~10 s for this one: the fixture is the reference)
        ref = np.load(os.path.join(GOLD, "jpeg_decoded.npz"))[_name(path)]
    else:
        ref = J.decode(data)
    assert np.array_equal(out, ref)


def test_unsupported_file_raises_and_decodes_nothing(device):
    from nopesac_amd import jpeg
    good = open(FILES[0], "rb").read()
    bad = open(os.path.join(GOLD, "jpeg", "unsupported_progressive_40x56.jpg"), "rb").read()
    with pytest.raises(jpeg.JpegUnsupported):
        jpeg.decode_batch([good, bad], device)


def test_scannet_sized_frames_match_pillow(device):
    """968 x 1296, 4:2:0, no restart markers (one serial Huffman chain per image), with and without restart markers, two files per batch"""
    Image = pytest.importorskip("PIL.Image")
    from nopesac_amd import jpeg
    rng = np.random.default_rng(11)
    yy, xx = np.mgrid[0:968, 0:1296].astype(np.float32)
    files, refs = [], []
    for i, opt in enumerate((dict(quality=90, subsampling=2), dict(quality=75, subsampling=2, restart_marker_rows=1))):
        a = np.stack([128 + 100 * np.sin(xx / (9 + i) + yy / 31), 128 + 80 * np.cos(yy / 13) * np.sin(xx / 57), 255 * ((xx // 40 + yy // 24) % 2)], -1)
        a = np.clip(a + rng.normal(0, 8, a.shape), 0, 255).astype(np.uint8)
        b = io.BytesIO()
        Image.fromarray(a).save(b, format="JPEG", **opt)
        files.append(b.getvalue())
        refs.append(np.asarray(Image.open(io.BytesIO(b.getvalue())).convert("RGB")))
    outs = jpeg.decode_batch(files, device)
    for o, r in zip(outs, refs):
        assert np.array_equal(o.cpu().numpy(), r)
