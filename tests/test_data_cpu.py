"""Dataset plumbing on the host (no GPU): pair json loading, the Matterport3D mapper path (PIL decode, channel order, root
replacement), and the invariants of the cv2-INTER_LINEAR restatement used for the ScanNet path."""
import json
import os

import numpy as np
import pytest
import torch

from tests.util import ROOT


def _cfg(extra=()):
    from nopesac_amd.config import get_cfg
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs", "inference_mp3d.yaml"))
    cfg.merge_from_list(list(extra))
    return cfg


def _write_pair(tmp_path, shape=(480, 640)):
    from PIL import Image
    rng = np.random.default_rng(0)
    files = []
    for i in range(2):
        arr = rng.integers(0, 256, (*shape, 3), dtype=np.uint8)
        f = tmp_path / f"img{i}.png"
        Image.fromarray(arr).save(f)
        files.append((str(f), arr))
    entry = {"0": {"file_name": files[0][0], "image_id": "house_0_1", "height": shape[0], "width": shape[1]},
             "1": {"file_name": files[1][0], "image_id": "house_0_2", "height": shape[0], "width": shape[1]},
             "rel_pose": {"position": [0.1, 0.2, 0.3], "rotation": [1.0, 0.0, 0.0, 0.0]}}
    jf = tmp_path / "cached_set_test.json"
    json.dump({"categories": [{"id": 1, "name": "plane"}], "data": [entry]}, open(jf, "w"))
    return str(jf), files


def test_splits_and_json(tmp_path):
    from nopesac_amd import data
    assert data.dataset_json("mp3d_test").endswith("mp3d_dataset/mp3d_planercnn_json/cached_set_test.json")
    assert data.dataset_json("scannet_test", "/d").startswith("/d/scannet_dataset/scannet_json/")
    with pytest.raises(KeyError):
        data.dataset_json("coco")
    jf, _ = _write_pair(tmp_path)
    pairs = data.load_pairs_json(jf)
    assert len(pairs) == 1 and pairs[0]["rel_pose"]["rotation"][0] == 1.0
    bad = tmp_path / "bad.json"
    json.dump({"images": []}, open(bad, "w"))
    with pytest.raises(ValueError):
        data.load_pairs_json(str(bad))


def test_mp3d_mapper(tmp_path):
    from nopesac_amd import data
    jf, files = _write_pair(tmp_path)
    entry = data.load_pairs_json(jf)[0]
    out = data.PairMapper(_cfg(), "mp3d_test")(entry)                       # configs/Base.yaml: INPUT.FORMAT = RGB
    for v in "01":
        img = out[v]["image"]
        assert img.dtype == torch.float32 and img.shape == (3, 480, 640) and not img.is_cuda
        assert torch.equal(img, torch.from_numpy(files[int(v)][1].transpose(2, 0, 1).astype("float32")))
    assert "image" not in entry["0"] and out["rel_pose"] == entry["rel_pose"]            # deep copy, pose passed through
    u8 = data.PairMapper(_cfg(), "mp3d_test", uint8=True)(entry)            # 8-bit hand-over: the same samples, a quarter of the bytes
    for v in "01":
        assert u8[v]["image"].dtype == torch.uint8 and u8[v]["image"].is_contiguous()
        assert torch.equal(u8[v]["image"].float(), out[v]["image"])
    bgr = data.PairMapper(_cfg(["INPUT.FORMAT", "BGR"]), "mp3d_test")(entry)
    assert torch.equal(bgr["0"]["image"], out["0"]["image"].flip(0))
    moved = dict(entry, **{"0": dict(entry["0"], file_name=data.MP3D_ORIGINAL_ROOT + "x.png")})
    with pytest.raises(FileNotFoundError):
        data.PairMapper(_cfg(["DATASETS.ROOT_DIR", str(tmp_path) + "/"]), "mp3d_test")(moved)   # prefix replaced -> tmp_path/x.png
    wrong = dict(entry, **{"0": dict(entry["0"], height=100)})
    with pytest.raises(ValueError):
        data.PairMapper(_cfg(), "mp3d_test")(wrong)


def test_resize_oracle_invariants():
    from oracle.resize_oracle import resize_bilinear_u8
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (31, 45, 3), dtype=np.uint8)
    assert np.array_equal(resize_bilinear_u8(img, 31, 45), img)                          # same size: identity
    const = np.full((20, 30, 3), 137, np.uint8)
    assert np.array_equal(resize_bilinear_u8(const, 48, 64), const[:1, :1].repeat(48, 0).repeat(64, 1))
    up = resize_bilinear_u8(img, 62, 90)                                                 # exact 2x: each output is a 1/4-3/4 blend
    assert up.shape == (62, 90, 3) and int(up.min()) >= int(img.min()) and int(up.max()) <= int(img.max())
    ramp = np.tile(np.arange(0, 200, 4, dtype=np.uint8)[None, :, None], (8, 1, 3))       # horizontal ramp, 50 px -> 100 px
    r2 = resize_bilinear_u8(ramp, 8, 100).astype(int)
    assert np.all(np.diff(r2[0, :, 0]) >= 0) and abs(int(r2[0, 50, 0]) - 99) <= 2        # monotone, midpoint preserved
    half = resize_bilinear_u8(np.array([[[0], [100]], [[200], [60]]], np.uint8).repeat(1, 2), 1, 1)
    assert int(half[0, 0, 0]) == 90                                                      # 2x2 -> 1x1: the mean of the four


def test_lazy_pairs_decode_ahead_in_order(tmp_path):
    """data.LazyPairs (what the runner iterates for a real split): same mapped dicts as the eager mapper, in order, through the
    thread pool; slicing = rank shards; a ragged last batch."""
    from nopesac_amd import data
    from PIL import Image
    rng = np.random.default_rng(3)
    entries = []
    for k in range(7):
        pair = {}
        for v in "01":
            arr = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
            f = tmp_path / f"p{k}_{v}.png"
            Image.fromarray(arr).save(f)
            pair[v] = {"file_name": str(f), "image_id": f"h_{k}_{v}", "height": 480, "width": 640}
        entries.append(pair)
    jf = tmp_path / "cached_set_test.json"
    json.dump({"categories": [], "data": entries}, open(jf, "w"))
    mapper = data.PairMapper(_cfg(), "mp3d_test", uint8=True)
    lazy = data.LazyPairs(data.load_pairs_json(str(jf)), mapper, workers=3, prefetch=4)
    assert len(lazy) == 7 and len(lazy[2:5]) == 3 and lazy[1]["0"]["image_id"] == "h_1_0"
    got = list(lazy.iter_batches(3))
    assert [len(b) for b in got] == [3, 3, 1]
    flat = [p for b in got for p in b]
    for k, p in enumerate(flat):
        ref = mapper(entries[k])
        assert p["0"]["image_id"] == f"h_{k}_0" and torch.equal(p["0"]["image"], ref["0"]["image"]) and torch.equal(p["1"]["image"], ref["1"]["image"])
    assert [p["0"]["image_id"] for p in lazy[5:]] == ["h_5_0", "h_6_0"]
    # the batch path (one native call per batch, csrc/png_host.hip) against the per-image path it replaces, float32 hand-over and BGR too;
    # a palette file and a 16-bit file inside a batch (the second is left to PIL: status -2 -> the general reader)
    Image.fromarray(rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)).quantize(64).save(entries[4]["1"]["file_name"])
    Image.fromarray(rng.integers(0, 65535, (480, 640), dtype=np.uint16)).save(entries[2]["0"]["file_name"])
    for overrides, u8 in (([], False), (["INPUT.FORMAT", "BGR"], True)):
        m2 = data.PairMapper(_cfg(overrides), "mp3d_test", uint8=u8)
        lz = data.LazyPairs(entries, m2, workers=4, prefetch=4)
        batched = [p for b in lz.iter_batches(4) for p in b]
        os.environ["NOPESAC_PNG_NATIVE"] = "0"
        try:
            plain = [p for b in lz.iter_batches(4) for p in b]
        finally:
            os.environ.pop("NOPESAC_PNG_NATIVE")
        assert len(batched) == len(plain) == 7
        for a, b in zip(batched, plain):
            for v in "01":
                assert a[v]["image"].dtype == b[v]["image"].dtype and torch.equal(a[v]["image"], b[v]["image"]), (overrides, a[v]["image_id"])
    assert "image" not in entries[0]["0"]                                                # the json entries are never written to
