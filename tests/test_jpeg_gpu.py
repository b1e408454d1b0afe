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


def _scannet_cfg():
    from nopesac_amd.config import get_cfg
    from tests.util import ROOT
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs", "inference_scannet.yaml"))
    return cfg


def test_scannet_mapper_decodes_jpeg_on_the_gpu_like_the_host_path(device, tmp_path):
    """data.PairMapper on JPEG frames: GPU decode + GPU resize = the host path (PIL decode, the reference's reader) + the same resize,
    bit for bit, as float32 / uint8, on the device / on the host; a progressive file goes through PIL and is counted."""
    import shutil
    from nopesac_amd import data
    cfg = _scannet_cfg()
    src = os.path.join(GOLD, "jpeg")
    names = ["s420_q88_scannet_like_242x324.jpg", "s444_q85_33x41.jpg", "unsupported_progressive_40x56.jpg", "gray_q80_45x31.jpg"]
    for n in names:
        shutil.copy(os.path.join(src, n), tmp_path / n)
    entries = [{"0": {"file_name": str(tmp_path / names[0]), "image_id": "a-0"}, "1": {"file_name": str(tmp_path / names[1]), "image_id": "a-1"}},
               {"0": {"file_name": str(tmp_path / names[2]), "image_id": "b-0"}, "1": {"file_name": str(tmp_path / names[3]), "image_id": "b-1"}}]
    for uint8 in (False, True):
        host = data.PairMapper(cfg, "scannet_test", uint8=uint8, gpu_jpeg=False)
        gpu = data.PairMapper(cfg, "scannet_test", uint8=uint8)
        gpu_dev = data.PairMapper(cfg, "scannet_test", device=device, uint8=uint8)
        assert gpu._use_gpu_jpeg() and gpu_dev._use_gpu_jpeg() and not host._use_gpu_jpeg()
        ref = [host(e) for e in entries]
        for m in (gpu, gpu_dev):
            got = [m(e) for e in entries]
            batch = m.map_batch(entries)
            for r, g, b in zip(ref, got, batch):
                for v in "01":
                    assert g[v]["image"].shape == (3, 480, 640) and g[v]["image"].dtype == r[v]["image"].dtype
                    assert torch.equal(g[v]["image"].cpu(), r[v]["image"]) and torch.equal(b[v]["image"].cpu(), r[v]["image"])
                    assert g[v]["image"].is_cuda == (m is gpu_dev)
            assert m.host_decoded == 2                    # the progressive file, once per pass


def test_lazy_pairs_decode_batches_ahead_on_the_gpu(device, tmp_path):
    """data.LazyPairs.iter_batches over a JPEG split: reader threads + one decode chain per batch on side streams; same mapped dicts, in
    order, as the host decoder gives."""
    import shutil
    from nopesac_amd import data
    cfg = _scannet_cfg()
    files = [p for p in FILES]
    entries = []
    for i in range(7):
        pair = {}
        for v in "01":
            srcp = files[(2 * i + int(v)) % len(files)]
            dst = tmp_path / ("%d_%s_%s" % (i, v, os.path.basename(srcp)))
            shutil.copy(srcp, dst)
            pair[v] = {"file_name": str(dst), "image_id": "%d-%s" % (i, v)}
        entries.append(pair)
    host = data.PairMapper(cfg, "scannet_test", uint8=True, gpu_jpeg=False)
    ref = [host(e) for e in entries]
    lazy = data.LazyPairs(entries, data.PairMapper(cfg, "scannet_test", device=device, uint8=True), workers=3, prefetch=8)
    got = [it for batch in lazy.iter_batches(3) for it in batch]
    assert len(got) == 7
    torch.cuda.synchronize()
    for r, g in zip(ref, got):
        for v in "01":
            assert g[v]["image_id"] == r[v]["image_id"] and g[v]["image"].is_cuda and torch.equal(g[v]["image"].cpu(), r[v]["image"])


def test_same_size_batch_is_resized_in_one_launch_and_stays_in_hbm(device, tmp_path):
    """A ScanNet batch (all frames one size): LazyPairs' GPU path resizes and transposes the whole batch in ONE launch
    (nopesac_resize_bilinear_u8_batch) into one [n,3,480,640] tensor in the model's order (view 0 of every pair, then view 1), hands out
    views of it that STAY on the device even for a mapper built without one (the host round trip was most of a batch's mapping time),
    and the model takes the batch without a copy - same pixels as the host path (PIL decode + per-image resize), uint8 and float32."""
    from PIL import Image
    from nopesac_amd import data, ops
    from nopesac_amd.modeling.meta_arch import PlaneTR_NopeSAC as M
    cfg = _scannet_cfg()
    rng = np.random.default_rng(9)
    yy, xx = np.mgrid[0:242, 0:324].astype(np.float32)
    entries = []
    for i in range(5):
        pair = {}
        for v in "01":
            a = np.stack([128 + 90 * np.sin(xx / (9 + i) + yy / 14), 128 + 70 * np.cos(yy / (7 + int(v))) * np.sin(xx / 20), 120 + 100 * ((xx // 16 + yy // 12) % 2)], -1)
            f = tmp_path / ("%d_%s.jpg" % (i, v))
            Image.fromarray(np.clip(a + rng.normal(0, 4.0, a.shape), 0, 255).astype(np.uint8)).save(f, format="JPEG", quality=88, subsampling=2)
            pair[v] = {"file_name": str(f), "image_id": "%d-%s" % (i, v)}
        entries.append(pair)
    for uint8 in (True, False):
        host = data.PairMapper(cfg, "scannet_test", uint8=uint8, gpu_jpeg=False)
        ref = [host(e) for e in entries]
        lazy = data.LazyPairs(entries, data.PairMapper(cfg, "scannet_test", uint8=uint8), workers=2, prefetch=8)
        batches = list(lazy.iter_batches(3))
        assert [len(b) for b in batches] == [3, 2]
        k = 0
        for b in batches:
            imgs = [p["0"]["image"] for p in b] + [p["1"]["image"] for p in b]
            assert all(t.is_cuda and t.shape == (3, 480, 640) for t in imgs)
            whole = M._as_one_batch(imgs)
            assert whole is not None and whole.shape == (2 * len(b), 3, 480, 640)
            for p in b:
                for v in "01":
                    assert p[v]["image_id"] == ref[k][v]["image_id"] and torch.equal(p[v]["image"].cpu(), ref[k][v]["image"]), (uint8, k, v)
                k += 1
    # the batched kernel against the per-image one, both layouts
    dec = [torch.from_numpy(rng.integers(0, 256, (5, 37, 53, 3), dtype=np.uint8)).to(device)]
    views = [dec[0][i] for i in range(5)]
    one = torch.stack([ops.resize_bilinear_u8(v, 24, 40) for v in views])
    assert torch.equal(ops.resize_bilinear_u8_batch(views, 24, 40, chw=False), one)
    assert torch.equal(ops.resize_bilinear_u8_batch(views, 24, 40, chw=True), one.permute(0, 3, 1, 2).contiguous())
    assert ops.resize_bilinear_u8_batch([views[1], views[0]], 24, 40) is None


def test_self_synchronising_decoder_settles_and_matches_the_serial_kernel(device):
    """restart-free files of >= 4 KB go through nopesac_jpeg_huffman_parallel: every image must settle (par_done) and give the bytes the
    one-wave-per-interval kernel gives (parallel=False)"""
    from nopesac_amd import jpeg
    big = [p for p in FILES if os.path.getsize(p) >= 8192 and "rst" not in p]
    assert len(big) >= 2
    files = [open(p, "rb").read() for p in big] * 3
    st = {}
    a = jpeg.decode_batch(files, device, stats=st)
    b = jpeg.decode_batch(files, device, parallel=False)
    assert st and int(st["par_done"].sum()) == len(files), (st["par_done"].tolist(), st["changed"].tolist())
    assert int(st["changed"][-1].sum()) == 0
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_unsettled_images_fall_through_to_the_serial_kernel_on_the_device(device):
    """an image whose lanes still moved in the last pass keeps par_done = 0: nopesac_jpeg_huffman decodes it (no host involvement)"""
    from nopesac_amd import jpeg
    dec = np.load(os.path.join(GOLD, "jpeg_decoded.npz"))
    files = [open(p, "rb").read() for p in FILES]
    st = {}
    outs = jpeg.decode_batch(files, device, stats=st, _force_unsettled=True)
    assert st and int(st["par_done"].sum()) == 0
    for p, o in zip(FILES, outs):
        assert np.array_equal(o.cpu().numpy(), dec[_name(p)]), _name(p)


def test_parallel_decoder_on_every_sampling_mode(device):
    """gray, 4:4:4, 4:2:2 and 4:2:0 streams long enough for the self-synchronising path (odd sizes, optimised Huffman tables as well):
    settled, equal to the serial kernel and to Pillow"""
    Image = pytest.importorskip("PIL.Image")
    from nopesac_amd import jpeg
    rng = np.random.default_rng(21)
    yy, xx = np.mgrid[0:203, 0:317].astype(np.float32)
    base = np.stack([128 + 100 * np.sin(xx / 11 + yy / 23), 128 + 90 * np.cos(yy / 7) * np.sin(xx / 19), 255 * ((xx // 9 + yy // 6) % 2)], -1)
    img = np.clip(base + rng.normal(0, 10, base.shape), 0, 255).astype(np.uint8)
    files, refs = [], []
    for opt in (dict(subsampling=0, quality=92), dict(subsampling=1, quality=90), dict(subsampling=2, quality=95), dict(quality=93),
                dict(subsampling=2, quality=85, optimize=True)):
        a = img[..., 0] if "subsampling" not in opt else img
        b = io.BytesIO()
        Image.fromarray(a).save(b, format="JPEG", **opt)
        files.append(b.getvalue())
        refs.append(np.asarray(Image.open(io.BytesIO(b.getvalue())).convert("RGB")))
    assert all(len(f) >= 8192 for f in files), [len(f) for f in files]
    st = {}
    a = jpeg.decode_batch(files, device, stats=st)
    b = jpeg.decode_batch(files, device, parallel=False)
    assert int(st["par_done"].sum()) == len(files), st["par_done"].tolist()
    for x, y, r in zip(a, b, refs):
        assert torch.equal(x, y) and np.array_equal(x.cpu().numpy(), r)
