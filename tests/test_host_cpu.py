"""CPU: C-ABI surface, config surface, registries/state-dict contract, host-side packaging logic."""
import os

import pytest
import torch

from tests.util import ROOT


def test_library_exports_every_declared_symbol():
    from nopesac_amd import _lib
    lib = _lib.load()
    declared = _lib.declared_symbols()
    assert len(declared) >= 20 and set(declared) == set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.nopesac_version() >= 100


def test_argument_errors_are_reported_without_a_gpu():
    from nopesac_amd import _lib
    lib = _lib.load()
    rc = lib.nopesac_conv2d_nhwc(None, None, None, None, None, None, 1, 1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 0, 0, 0, 0, 0, None)
    assert rc == -1 and b"null pointer" in lib.nopesac_last_error()
    with pytest.raises(_lib.HipKernelError):
        _lib.check(rc, "nopesac_conv2d_nhwc")


def test_no_cpu_fallback():
    from nopesac_amd import ops
    with pytest.raises(AssertionError, match="no CPU path"):
        ops.softmax_rows(torch.zeros(2, 8))


def test_missing_library_fails_loudly(monkeypatch):
    from nopesac_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libnopesac_hip.so")
    with pytest.raises(RuntimeError, match="not built"):
        _lib.load()


@pytest.mark.parametrize("name", ["inference_mp3d.yaml", "inference_scannet.yaml"])
def test_reference_configs_load_unchanged(name):
    from nopesac_amd.config import get_cfg
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs", name))
    cfg.merge_from_list(["TEST.MATCHING_SCORE_THRESHOLD", "0.3", "MODEL.DEVICE", "cpu"])
    cfg.freeze()
    assert cfg.MODEL.META_ARCHITECTURE == "PlaneTR_NopeSAC" and cfg.MODEL.CAMERA_HEAD.NAME == "PlaneCameraHead"
    assert cfg.MODEL.RESNETS.STRIDE_IN_1X1 is False and cfg.MODEL.SEM_SEG_HEAD.NUM_OBJECT_QUERIES == 50
    assert cfg.TEST.MATCHING_SCORE_THRESHOLD == 0.3 and cfg.INPUT.FORMAT == "RGB"
    assert isinstance(cfg.DATASETS.TEST, tuple)
    with pytest.raises(AttributeError):
        cfg.MODEL.DEVICE = "cuda"
    with pytest.raises(KeyError):
        get_cfg().merge_from_list(["MODEL.NO_SUCH_KEY", 1])


def test_meta_arch_registry_and_state_dict_contract(sd50):
    from nopesac_amd.config import get_cfg
    from nopesac_amd.registry import META_ARCH_REGISTRY, build_model
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs", "inference_mp3d.yaml"))
    cfg.MODEL.DEVICE = "cpu"
    model = build_model(cfg)
    assert type(model) is META_ARCH_REGISTRY.get("PlaneTR_NopeSAC")
    assert list(model.state_dict().keys()) and set(model.state_dict()) == set(sd50)
    # d2-style checkpoint dict with the reference-only criterion buffer
    sd = dict(sd50)
    sd["criterion.empty_weight"] = torch.ones(2)
    model.load_state_dict({"model": sd})
    k = "camera_head_list.0.geo_encoder.layers.0.weight"
    assert torch.equal(model.state_dict()[k], sd50[k])
    with pytest.raises(NotImplementedError):
        model.train()([{}])


def test_decode_masks():
    from nopesac_amd.modeling import decode_masks
    winner = torch.tensor([[3 | 0x80, 3, 7 | 0x80], [7, 7 | 0x80, 3 | 0x80]], dtype=torch.uint8)
    m = decode_masks(winner, torch.tensor([3, 7]), False)
    assert m.tolist() == [[[True, False, False], [False, False, True]], [[False, False, True], [False, True, False]]]
    m = decode_masks(winner, torch.tensor([7]), True)
    assert m.tolist() == [[[False, False, True], [True, True, False]]]


def test_synth_is_deterministic():
    from nopesac_amd.synth import structured_image, synth_pair, synth_tensor
    a = synth_tensor("backbone.res2.0.conv1.weight", (64, 64, 1, 1))
    assert torch.equal(a, synth_tensor("backbone.res2.0.conv1.weight", (64, 64, 1, 1)))
    p = synth_pair(3)
    assert p["0"]["image"].shape == (3, 480, 640) and float(p["0"]["image"].max()) <= 255
    assert torch.equal(structured_image(5), structured_image(5))


def test_fp8_weight_packing_cpu():
    """Host side of the fp8 conv (no GPU): per-output-channel e4m3fn quantisation and the fragment-major byte order
    [N/32][K/64][2][64][16] that nopesac_conv2d_nhwc_fp8 documents in include/nopesac_hip.h."""
    import torch
    from nopesac_amd import ops
    g = torch.Generator().manual_seed(3)
    w = torch.randn(64, 3, 3, 64, generator=g) * (1 + torch.arange(64).view(-1, 1, 1, 1))
    w8f, sc = ops.quantize_weights_fp8(w)
    assert w8f.dtype == torch.float8_e4m3fn and w8f.shape == (64, 576) and sc.shape == (64,)
    w8 = (w.reshape(64, -1) / sc[:, None]).clamp(-448, 448).to(torch.float8_e4m3fn)
    assert float(w8.float().abs().amax(dim=1).min()) == 448.0                     # every row uses the full e4m3 range
    assert float(((w8.float() * sc[:, None]) - w.reshape(64, -1)).abs().max() / w.abs().max()) < 0.04
    raw, frag = w8.view(torch.uint8), w8f.view(torch.uint8).view(2, 9, 2, 64, 16)
    for nt, kf, h, lane, j in [(0, 0, 0, 0, 0), (1, 8, 1, 63, 15), (0, 3, 1, 37, 5), (1, 5, 0, 32, 9)]:
        assert int(frag[nt, kf, h, lane, j]) == int(raw[nt * 32 + (lane & 31), kf * 64 + 32 * (lane >> 5) + 16 * h + j])


# ---- plugin boundary under a FOREIGN config node (meta_arch/siamese_planeTR.py:33-38,133: detectron2 builds `cls(cfg)`) ----
class _ForeignCfg(dict):
    """Stand-in for yacs / detectron2's CfgNode: a different dict subclass with attribute access and NO MODEL.AMD node."""

    def __init__(self, init=None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = _ForeignCfg(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _foreign_cfg():
    from nopesac_amd.config import get_cfg
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs", "inference_mp3d.yaml"))
    cfg.MODEL.DEVICE = "cpu"
    plain = _ForeignCfg(cfg)
    del plain["MODEL"]["AMD"]                       # what detectron2 + get_sparseplane_cfg_defaults would hand over
    return plain


def test_configurable_accepts_foreign_cfg_node(sd50):
    from nopesac_amd.registry import META_ARCH_REGISTRY, is_config_node
    cfg = _foreign_cfg()
    assert is_config_node(cfg) and not is_config_node({"MODEL": 3}) and not is_config_node(7)
    cls = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)
    for model in (cls(cfg), cls(cfg=cfg)):                              # d2's build_model calls cls(cfg)
        assert model.num_queries == 50 and model.compute_dtype == torch.float32 and model.output_rle and model.two_streams
        assert set(model.state_dict()) == set(sd50)
    # explicit keyword construction passes straight through, cfg among the kwargs included
    m = cls(num_queries=50, pixel_mean=cfg.MODEL.PIXEL_MEAN, pixel_std=cfg.MODEL.PIXEL_STD, device="cpu", cfg=cfg)
    assert m.num_queries == 50
    with pytest.raises(TypeError):
        cls(3)


def test_amd_options_overlay():
    from nopesac_amd.config import add_amd_defaults, amd_options
    cfg = _foreign_cfg()
    assert amd_options(cfg).COMPUTE_DTYPE == "float32" and "AMD" not in cfg.MODEL          # the foreign node is not touched
    add_amd_defaults(cfg)                                                                   # d2-side opt-in
    assert type(cfg.MODEL.AMD) is _ForeignCfg
    cfg.MODEL.AMD.COMPUTE_DTYPE = "bfloat16"
    o = amd_options(cfg)
    assert o.COMPUTE_DTYPE == "bfloat16" and o.TWO_STREAMS is True


def test_registers_into_detectron2_when_importable(monkeypatch):
    """A stub `detectron2.modeling` with an fvcore-style registry: the drop-in must appear under the reference's name,
    replace a previously registered class of that name, and build through a d2-style build_model."""
    import sys
    import types
    from nopesac_amd import registry

    d2reg = registry.Registry("META_ARCH")                   # same protocol as fvcore.common.registry.Registry
    d2 = types.ModuleType("detectron2")
    d2m = types.ModuleType("detectron2.modeling")
    d2m.META_ARCH_REGISTRY = d2reg
    d2.modeling = d2m
    monkeypatch.setitem(sys.modules, "detectron2", d2)
    monkeypatch.setitem(sys.modules, "detectron2.modeling", d2m)

    class PlaneTR_NopeSAC:                                   # "the reference's class was registered first"
        pass

    d2reg.register(PlaneTR_NopeSAC)
    assert registry.register_into_detectron2(override=False) is False
    assert registry.register_into_detectron2() is True
    ours = registry.META_ARCH_REGISTRY.get("PlaneTR_NopeSAC")
    assert d2reg.get("PlaneTR_NopeSAC") is ours
    assert registry.register_into_detectron2() is True       # idempotent
    cfg = _foreign_cfg()
    model = d2reg.get(cfg.MODEL.META_ARCHITECTURE)(cfg)      # detectron2.modeling.build_model's two lines
    assert type(model) is ours


def test_register_into_detectron2_is_a_noop_without_it():
    import importlib.util
    from nopesac_amd import registry
    if importlib.util.find_spec("detectron2") is None:
        assert registry.register_into_detectron2() is False


def test_derived_weight_caches_are_dropped_on_reload():
    """fragment-major copies for the fused kernels must not survive load_state_dict / .to()."""
    from nopesac_amd.modeling.params import ParamModule
    pm = ParamModule({"a.weight": (4, 4)})
    for name in ParamModule._DERIVED_CACHES:
        pm.__dict__[name] = {0: "stale"}
    pm._packed = {"x": 1}
    pm.load_state_dict({"a.weight": torch.ones(4, 4)})
    assert pm._packed is None and not any(n in pm.__dict__ for n in ParamModule._DERIVED_CACHES)
    pm.__dict__["_fused_w"] = {0: "stale"}
    pm.to(torch.float32)
    assert "_fused_w" not in pm.__dict__


def test_tuner_key_and_eligibility_fallback():
    """a remembered kernel configuration that this call is not eligible for falls back to the heuristic."""
    from nopesac_amd import ops
    t = ops.ConvTuner()
    t.best["k"] = ops.CFG_HALO16
    assert t.choose("k", None, (ops.CFG_BFRAG3, ops.CFG_HALO16)) == ops.CFG_HALO16
    assert t.choose("k", None, (ops.CFG_BFRAG3,)) == 0
    assert t.choose("unknown", None, ()) == 0


def test_library_is_built_without_packed_fp32_instructions():
    """Round-3 finding (profiles/r3_packed_fp32_hazard.txt): v_pk_*_f32 results are corrupted next to another wave's MFMAs on the same
    SIMD, so the build must keep the `packed-fp32-ops` target feature off for every kernel file."""
    from nopesac_amd import build
    flags = " ".join(build.FLAGS)
    assert "-target-feature -Xclang -packed-fp32-ops" in flags, flags
    assert all("packed-fp32" not in " ".join(v) or "-packed-fp32-ops" in " ".join(v) for v in build.EXTRA_FLAGS.values())


def test_host_fetch_and_cached_constant_on_cpu_tensors():
    """ops.HostFetch passes host tensors through untouched (no device, no copy) and ops.cached_constant builds a key once."""
    import torch
    from nopesac_amd import ops
    a, b = torch.arange(6.0).view(2, 3), torch.tensor([1, 2, 3], dtype=torch.int32)
    f = ops.HostFetch({"a": a, "b": b})
    v = f.wait().views()
    assert v["a"] is not None and torch.equal(v["a"], a) and torch.equal(v["b"], b) and f.host is None
    assert ops.gather_to_host({"a": a})["a"].data_ptr() == a.data_ptr()
    calls, cache = [], {}
    make = lambda: (calls.append(1), torch.zeros(3))[1]
    t1 = ops.cached_constant(cache, "k", make)
    t2 = ops.cached_constant(cache, "k", make)
    assert t1 is t2 and len(calls) == 1


def test_new_entry_points_reject_bad_arguments_without_a_gpu():
    """Argument checks of the round-3 entry points run before any HIP call: wrong arguments come back as an error code + message on a
    machine without a GPU (no compute is attempted)."""
    from nopesac_amd import _lib
    lib = _lib.load()
    err = lambda: lib.nopesac_last_error().decode()
    assert lib.nopesac_transformer_tail_bf16(*([None] * 13), 0, *([None] * 4), 0, 0, None, None, None, 0, None, None, None, 0, 64, None) != 0
    assert "transformer_tail" in err()
    assert lib.nopesac_mask_operands(None, 264, None, None, 1, 50, 64, None) != 0 and "mask_operands" in err()
    assert lib.nopesac_rle_compress_device_capped(None, None, None, 4, 8, 8, None, None, None, -1, None) != 0 and "rle_compress_device_capped" in err()
    assert lib.nopesac_tape_replay_on(None, None, None, 0) != 0 and "tape_replay" in err()
    assert lib.nopesac_metric_rows(None, None, None, None, None, None, None, None, 0, None, 4, None) != 0 and "metric_rows" in err()
    assert lib.nopesac_add_rows_bf16(None, None, None, None, 4, 6, 1, None) != 0 and "add_rows_bf16" in err()
    assert lib.nopesac_softmax_rows_pad(None, None, 4, 300, 304, 1, None) != 0 and "softmax_rows_pad" in err()
    assert lib.nopesac_concat_cols(None, 3, None, 4, None, 2, None) != 0 and "concat_cols" in err()


def test_round4_argument_checks_without_a_gpu():
    """Round-4 argument checks (advisor findings of round 3) come back as error codes before any HIP call: a launch tape with more
    streams than a replay accepts, the 4-wave Sinkhorn entry keeps the old
    contract (nq <= 128)."""
    import ctypes
    from nopesac_amd import _lib
    lib = _lib.load()
    err = lambda: lib.nopesac_last_error().decode()
    out = ctypes.c_void_p()
    fake_graph = ctypes.c_void_p(16)                      # never dereferenced: the range check comes first
    assert lib.nopesac_tape_create_ex(fake_graph, 18, ctypes.byref(out), None) != 0 and "max_streams" in err()
    assert lib.nopesac_tape_create_ex(fake_graph, 0, ctypes.byref(out), None) != 0 and "tape_create" in err()
    assert lib.nopesac_matcher_sinkhorn(*([None] * 7), 1.0, 1.0, 200, 0.2, 1, 50, None, None, None) != 0 and "sinkhorn" in err()


def test_training_twin_argument_checks_without_a_gpu():
    """The loss entry of the training-side refine twin and the twin's mode bit of the soft vote check their arguments before any HIP
    call (camera_head.py:737-923 counterpart; include/nopesac_hip.h)."""
    from nopesac_amd import _lib
    lib = _lib.load()
    err = lambda: lib.nopesac_last_error().decode()
    assert lib.nopesac_plane_cam_ref_losses(*([None] * 11), 2, 50, 1.0, None, None) != 0 and "plane_cam_ref_losses" in err()
    assert lib.nopesac_ransac_soft_vote(*([None] * 21), 1, 50, 16 + 4, *([None] * 6), None) != 0 and "ransac_soft_vote" in err()
    assert lib.nopesac_ransac_soft_vote(*([None] * 21), 1, 50, 32, *([None] * 6), None) != 0 and "ransac_soft_vote" in err()
    assert lib.nopesac_camera_pose_loss(None, None, None, 7, None, 7, 4, 0.0, 1.0, None, None) != 0 and "camera_pose_loss" in err()


def test_png_host_decoder_matches_pillow(tmp_path):
    """csrc/png_host.hip (the mp3d split's frames; called with the interpreter lock released) against PIL - the reference's decoder
    (planercnn_transforms.py:210-227 -> utils.read_image) - on every colour type it takes, both channel orders, Pillow's default and
    `optimize` encodings (other filter / zlib choices), a multi-IDAT file; variants it leaves to PIL (16-bit, interlaced-free check via
    the header, sub-byte palette) and damaged files must return None, and data.read_image must give PIL's pixels either way."""
    import numpy as np
    from PIL import Image
    from nopesac_amd import data
    rng = np.random.default_rng(3)
    yy, xx = np.mgrid[0:97, 0:131].astype(np.float32)
    a = np.clip(np.stack([128 + 90 * np.sin(xx / 9 + yy / 14), 128 + 70 * np.cos(yy / 7) * np.sin(xx / 20), 120 + 100 * ((xx // 16 + yy // 12) % 2)], -1)
                + rng.normal(0, 4.0, (97, 131, 3)), 0, 255).astype(np.uint8)
    modes = {"RGB": Image.fromarray(a), "RGBA": Image.fromarray(np.concatenate([a, a[..., :1]], -1), "RGBA"), "L": Image.fromarray(a[..., 0]),
             "LA": Image.fromarray(np.ascontiguousarray(a[..., :2]), "LA"), "P": Image.fromarray(a).quantize(200)}
    for m, im in modes.items():
        for opt in (False, True):
            p = tmp_path / f"{m}_{opt}.png"
            im.save(p, optimize=opt)
            ref = np.asarray(Image.open(p).convert("RGB"))
            blob = p.read_bytes()
            for fmt in ("RGB", "BGR"):
                got = data.read_png_native(blob, fmt)
                assert got is not None and np.array_equal(got, ref if fmt == "RGB" else ref[..., ::-1]), (m, opt, fmt)
            assert np.array_equal(data.read_image(str(p), "BGR"), ref[..., ::-1])
    big = np.clip(rng.normal(128, 60, (300, 400, 3)), 0, 255).astype(np.uint8)          # incompressible: Pillow splits it into many IDAT chunks
    p = tmp_path / "noise.png"
    Image.fromarray(big).save(p)
    assert p.read_bytes().count(b"IDAT") > 1 and np.array_equal(data.read_png_native(p.read_bytes(), "RGB"), big)
    # left to PIL: 16-bit samples, 1-bit palette; damaged: a flipped byte inside the compressed data (CRC), a truncated file
    p16 = tmp_path / "g16.png"
    Image.fromarray((a[..., 0].astype(np.uint16) * 257)).save(p16)
    assert data.read_png_native(p16.read_bytes(), "RGB") is None
    p1 = tmp_path / "bw.png"
    Image.fromarray(a[..., 0] > 128).save(p1)
    assert data.read_png_native(p1.read_bytes(), "RGB") is None
    assert np.array_equal(data.read_image(str(p1), "RGB"), np.asarray(Image.open(p1).convert("RGB")))
    blob = bytearray((tmp_path / "RGB_False.png").read_bytes())
    blob[len(blob) // 2] ^= 0x55
    assert data.read_png_native(bytes(blob), "RGB") is None
    assert data.read_png_native((tmp_path / "RGB_False.png").read_bytes()[:-40], "RGB") is None
    assert data.read_png_native(b"not a png at all, but long enough to hold a header....", "RGB") is None
    # every row filter on every row (Pillow's encoder mostly picks Up and Paeth): files written here with ONE filter type each, three and
    # four bytes per pixel (the SSE2 Paeth path), one and two (its scalar form), odd widths - against PIL's decode of the same file
    import struct
    import zlib

    def png_with_filter(img, ftype, ctype):
        H, W, C = img.shape
        rows = img.reshape(H, W * C).astype(np.int32)
        out, prev = bytearray(), np.zeros(W * C, np.int32)
        for y in range(H):
            cur = rows[y]
            left = np.concatenate([np.zeros(C, np.int32), cur[:-C]])
            ul = np.concatenate([np.zeros(C, np.int32), prev[:-C]])
            if ftype == 0:
                f = cur
            elif ftype == 1:
                f = cur - left
            elif ftype == 2:
                f = cur - prev
            elif ftype == 3:
                f = cur - ((left + prev) >> 1)
            else:
                pp = left + prev - ul
                pa, pb, pc = abs(pp - left), abs(pp - prev), abs(pp - ul)
                f = cur - np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
            out.append(ftype)
            out += (f & 255).astype(np.uint8).tobytes()
            prev = cur
        chunk = lambda t, b: struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b) & 0xffffffff)
        return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, ctype, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(bytes(out), 6))
                + chunk(b"IEND", b""))
    import io
    for C, ctype in ((3, 2), (4, 6), (1, 0), (2, 4)):
        for Wd in (131, 4, 1):
            pix = rng.integers(0, 256, (23, Wd, C), dtype=np.uint8)
            pix[5:9] = pix[4:5]                                   # some exact repeats: ties between the Paeth candidates
            for ftype in range(5):
                blob_f = png_with_filter(pix, ftype, ctype)
                want_f = np.asarray(Image.open(io.BytesIO(blob_f)).convert("RGB"))
                got_f = data.read_png_native(blob_f, "RGB")
                assert got_f is not None and np.array_equal(got_f, want_f), (C, Wd, ftype)
    # the batch entry point (one call, its own threads): every file above at the batch's geometry decodes into its slot, channel-major or
    # interleaved, either channel order; other sizes (-5), variants left to PIL (-2), damaged (-3) and missing (-6) files report a status
    # and leave their slot untouched
    (tmp_path / "broken.png").write_bytes(bytes(blob))
    good = [str(tmp_path / f"{m}_{opt}.png") for m in modes for opt in (False, True)]
    paths = good[:3] + [str(p), str(p16), str(tmp_path / "broken.png"), str(tmp_path / "missing.png")] + good[3:]
    want = [0, 0, 0, -5, -2, -3, -6] + [0] * (len(good) - 3)
    for chw in (True, False):
        for fmt in ("RGB", "BGR"):
            for thr in (1, 5):
                pre = torch.full((len(paths), 3, 97, 131) if chw else (len(paths), 97, 131, 3), 7, dtype=torch.uint8)
                out, st = data.read_png_files(paths, fmt, 97, 131, chw=chw, threads=thr, out=pre)
                assert st == want and out.data_ptr() == pre.data_ptr()
                for i, (f, s_) in enumerate(zip(paths, st)):
                    if s_ != 0:
                        assert int(out[i].min()) == 7 and int(out[i].max()) == 7
                        continue
                    ref = np.asarray(Image.open(f).convert("RGB"))
                    ref = ref if fmt == "RGB" else ref[..., ::-1]
                    got = out[i].numpy()
                    assert np.array_equal(got.transpose(1, 2, 0) if chw else got, ref), (f, chw, fmt, thr)
    out16, st16 = data.read_png_files([str(p16)], "RGB", 97, 131)
    assert st16 == [-2]
    # the decoder's parked threads do not survive a fork: a child (a forked DataLoader worker) must get its own pool, not wait for threads
    # it does not have
    import os
    import signal
    import time
    want_px = data.read_png_files(good, "RGB", 97, 131, threads=4)[0].numpy().copy()
    child_out = torch.empty(len(good), 3, 97, 131, dtype=torch.uint8)           # (allocated here: the child does no torch work of its own -
    pid = os.fork()                                                             #  torch's OpenMP pool does not survive a fork either)
    if pid == 0:
        code = 4
        try:
            _, st_c = data.read_png_files(good, "RGB", 97, 131, threads=4, out=child_out)
            code = 0 if (not any(st_c) and np.array_equal(child_out.numpy(), want_px)) else 3
        finally:
            os._exit(code)
    t0, status = time.time(), None
    while time.time() - t0 < 60:
        done, status = os.waitpid(pid, os.WNOHANG)
        if done:
            break
        time.sleep(0.05)
    else:
        os.kill(pid, signal.SIGKILL)
        os.waitpid(pid, 0)
        raise AssertionError("the forked child did not finish: it waits for pool threads it does not have")
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0
    assert data.read_png_files([], "RGB", 97, 131)[1] == []


def test_cpu_budget_respects_affinity_and_quota(monkeypatch, tmp_path):
    """runner.cpu_budget: thread pools are sized from what the container may keep busy - the affinity mask capped by the cgroup CPU quota
    (cpu.max "1600000 100000" = 16 CPUs on a host that shows 256 hardware threads), never from os.cpu_count()."""
    import builtins
    import os
    from nopesac_amd import runner
    n = runner.cpu_budget()
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    assert 1 <= n <= aff
    real_open = builtins.open
    for text, want in (("1600000 100000\n", min(aff, 16)), ("max 100000\n", aff), ("150000 100000\n", min(aff, 2)), ("garbage", aff)):
        def fake_open(path, *a, **k):
            if str(path) == "/sys/fs/cgroup/cpu.max":
                f = tmp_path / "cpu.max"
                f.write_text(text)
                return real_open(f, *a, **k)
            if str(path).startswith("/sys/fs/cgroup/cpu/"):
                raise OSError("no v1 hierarchy")
            return real_open(path, *a, **k)
        monkeypatch.setattr(builtins, "open", fake_open)
        assert runner.cpu_budget() == want, text
        monkeypatch.setattr(builtins, "open", real_open)


def test_images_lying_back_to_back_become_one_transfer():
    """PlaneTR_NopeSAC._as_one_batch: the images data.LazyPairs yields are views of one batch buffer in the model's order (all view-0
    images, then all view-1 images) - they are recognised and copied to the device as ONE tensor; anything else (separate tensors, another
    order, a gap, mixed dtypes) keeps the per-image copies."""
    import torch
    from nopesac_amd.modeling.meta_arch import PlaneTR_NopeSAC as M
    buf = torch.arange(6 * 3 * 4 * 5, dtype=torch.uint8).view(6, 3, 4, 5)
    imgs = [buf[i] for i in range(6)]
    whole = M._as_one_batch(imgs)
    assert whole is not None and whole.shape == buf.shape and whole.data_ptr() == buf.data_ptr() and torch.equal(whole, buf)
    sub = M._as_one_batch(imgs[2:5])
    assert sub is not None and torch.equal(sub, buf[2:5]) and sub.data_ptr() == buf[2].data_ptr()
    assert M._as_one_batch([b.clone() for b in imgs]) is None
    assert M._as_one_batch([imgs[1], imgs[0], imgs[2]]) is None
    assert M._as_one_batch([imgs[0], imgs[2], imgs[4]]) is None
    assert M._as_one_batch([imgs[0], imgs[1].float()]) is None
    assert M._as_one_batch([buf[0, :, :2], buf[0, :, 2:]]) is None and M._as_one_batch(imgs[:1]) is None


def test_own_inflate_matches_zlib():
    """csrc/inflate_host.h (the PNG decoder's inflate: whole stream in memory, 11-bit primary table with two-literal entries, branch-free
    refill) against zlib - the decoder the reference's PIL uses - on every block type (stored / fixed / dynamic), compression level and
    strategy, small windows, flush points, far and short-period matches; malformed streams (truncated, bad header, flipped bits, an output
    buffer one byte short) must be refused or decode to what zlib decodes, never crash."""
    import ctypes
    import random
    import zlib
    import numpy as np
    from nopesac_amd import _lib
    L = _lib.load()

    def inf(comp, n, cap=None):
        cap = n if cap is None else cap
        out = ctypes.create_string_buffer(max(1, cap))
        got = L.nopesac_inflate_zlib_host(comp, len(comp), out, cap)
        return got, out.raw[:max(0, got)]
    rng = np.random.default_rng(0)
    cases = {
        "empty": b"", "one": b"a", "zeros": bytes(70000), "rand": rng.integers(0, 256, 60000, dtype=np.uint8).tobytes(),
        "text": b"the quick brown fox jumps over the lazy dog " * 2000, "lowent": rng.integers(0, 4, 90000, dtype=np.uint8).tobytes(),
        "rgbflat": bytes([10, 20, 30]) * 30000, "period5": bytes([1, 2, 3, 4, 5]) * 9000 + bytes([7, 7, 9, 9, 9, 9, 9]) * 3000,
        "mixed": b"".join(rng.integers(0, 256, int(rng.integers(1, 400)), dtype=np.uint8).tobytes() + bytes(int(rng.integers(0, 2000))) for _ in range(120)),
        "far": (lambda a: a + bytes(32768 - len(a) - 300) + a * 3)(rng.integers(0, 256, 300, dtype=np.uint8).tobytes()),
        "skew": bytes(np.minimum(255, rng.geometric(0.02, 120000)).astype(np.uint8)),          # long codes: second-level tables
    }
    for name, data in cases.items():
        for level in (0, 1, 6, 9):
            for strat in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
                for wbits in (15, 9):
                    c = zlib.compressobj(level, zlib.DEFLATED, wbits, 8, strat)
                    comp = c.compress(data) + c.flush()
                    got, out = inf(comp, len(data))
                    assert got == len(data) and out == data, (name, level, strat, wbits, got)
    c = zlib.compressobj(6)
    parts = [rng.integers(0, 50, 5000, dtype=np.uint8).tobytes() for _ in range(12)]
    comp = b"".join(c.compress(p) + c.flush(zlib.Z_SYNC_FLUSH if i % 2 else zlib.Z_FULL_FLUSH) for i, p in enumerate(parts)) + c.flush()
    assert inf(comp, 60000) == (60000, b"".join(parts))
    data = cases["text"]
    comp = zlib.compress(data, 6)
    assert inf(comp, len(data), len(data) - 1)[0] == -1 and inf(comp, len(data), len(data) + 100) == (len(data), data)
    assert all(inf(comp[:k], len(data))[0] == -1 for k in (0, 1, 2, 5, 10, len(comp) // 2, len(comp) - 5))
    assert inf(b"\x79" + comp[1:], len(data))[0] == -1 and inf(comp[:1] + b"\x00" + comp[2:], len(data))[0] == -1
    random.seed(1)
    refused = 0
    for name in ("text", "mixed", "skew", "rand"):
        data = cases[name]
        comp = bytearray(zlib.compress(data, 6))
        for _ in range(400):
            b = bytearray(comp)
            for _ in range(random.randint(1, 3)):
                b[random.randrange(2, len(b))] ^= 1 << random.randrange(8)
            got, out = inf(bytes(b), len(data))
            refused += got < 0
            if got > 0:                                       # accepted: then zlib's raw inflate yields the same bytes (raw: the Adler-32
                ref = zlib.decompressobj(-15).decompress(bytes(b)[2:], got)          # trailer is not checked here - a PNG chunk's CRC
                assert out == ref, name                                              # stands in for it)
    assert refused > 100


def test_native_jpeg_batch_prepare_equals_the_python_form(tmp_path):
    """csrc/jpeg_host.hip (one call per batch: file read, marker walk, checks, stuffing removal, derived Huffman tables, launch arrays) against
    jpeg.py's parse() + prepare_batch() - the per-file Python it replaces on the loader's reader threads - array for array, on every committed
    fixture the GPU decoder takes (all sampling modes, grey, restart markers, a 16-bit quantisation table if present) in mixed batches, one and
    several threads; files it must leave alone (progressive, not a JPEG, missing, truncated) are reported per file and fail the batch's fast
    path without touching the rest."""
    import glob
    import os
    import numpy as np
    import torch
    from PIL import Image
    from nopesac_amd import jpeg
    from tests.util import ROOT
    files = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "jpeg", "*.jpg")))
    good, bad = [], []
    for f in files:
        try:
            jpeg.parse(open(f, "rb").read())
            good.append(f)
        except jpeg.JpegUnsupported:
            bad.append(f)
    assert len(good) >= 8 and len(bad) >= 1
    rng = np.random.default_rng(4)
    yy, xx = np.mgrid[0:200, 0:264].astype(np.float32)
    big = tmp_path / "big.jpg"                                # long enough for the self-synchronising decoder's lanes
    Image.fromarray(np.clip(np.stack([128 + 90 * np.sin(xx / 9 + yy / 14), 128 + 70 * np.cos(yy / 7), 120 + 100 * ((xx // 16 + yy // 12) % 2)], -1)
                            + rng.normal(0, 6.0, (200, 264, 3)), 0, 255).astype(np.uint8)).save(big, format="JPEG", quality=95, subsampling=2)
    good = good + [str(big)]
    for order in (good, good[::-1], [good[-1]] * 3 + good[:2]):
        for parallel in (True, False):
            want = jpeg.prepare_batch([jpeg.parse(open(f, "rb").read()) for f in order], parallel)
            for thr in (1, 4):
                got, st = jpeg.prepare_files(order, threads=thr, parallel=parallel)
                assert st is None and got is not None and got.n == want.n
                for name in ("img32", "img64", "tables", "seg32", "seg64", "words"):
                    a, b = getattr(got, name), getattr(want, name)
                    assert a.shape == b.shape and torch.equal(a.view(torch.uint8) if a.dtype != b.dtype else a, b.view(torch.uint8) if a.dtype != b.dtype else b), (name, parallel, thr)
                assert (got.lane_img is None) == (want.lane_img is None) and (got.lane_img is None or torch.equal(got.lane_img, want.lane_img))
                assert (got.n_seg, got.n_lanes, got.n_blocks, got.max_px, got.coef_off, got.plane_off, got.out_off) == (
                    want.n_seg, want.n_lanes, want.n_blocks, want.max_px, want.coef_off, want.plane_off, want.out_off)
                assert [(g.height, g.width) for g in got.infos] == [(w.height, w.width) for w in want.infos]
    assert want.n_lanes == 0 and jpeg.prepare_batch([jpeg.parse(open(str(big), "rb").read())], True).n_lanes > 0
    # files the fast path leaves alone
    notjpg = tmp_path / "x.jpg"
    notjpg.write_bytes(b"\x89PNG not a jpeg at all")
    trunc = tmp_path / "t.jpg"
    trunc.write_bytes(open(good[0], "rb").read()[:-200])
    mixed = [good[0], bad[0], str(notjpg), str(tmp_path / "missing.jpg"), str(trunc), good[1]]
    got, st = jpeg.prepare_files(mixed, threads=3)
    assert got is None and st[0] == 0 and st[5] == 0 and st[1] == -2 and st[2] == -1 and st[3] == -6 and st[4] in (-3, -2)
    assert jpeg.prepare_files([], threads=2) == (None, [])
    # damaged files (bytes overwritten, inserted, cut off): the native walk never takes a file the Python walk refuses (it may be stricter:
    # the batch then goes down the Python path), and where both take it the arrays are equal
    import random
    random.seed(3)
    stricter = 0
    for it in range(150):
        b = bytearray(open(random.choice(good[:-1]), "rb").read())
        mode = random.random()
        if mode < 0.5:
            for _ in range(random.randint(1, 4)):
                b[random.randrange(len(b))] = random.randrange(256)
        elif mode < 0.8:
            b = b[:random.randrange(2, len(b))]
        else:
            i = random.randrange(len(b))
            b[i:i] = bytes(random.randrange(256) for _ in range(random.randint(1, 8)))
        pth = tmp_path / ("fuzz%d.jpg" % (it % 4))
        pth.write_bytes(bytes(b))
        try:
            info = jpeg.parse(bytes(b))
        except jpeg.JpegUnsupported:
            info = None
        hb, st = jpeg.prepare_files([str(pth)], threads=1)
        assert not (hb is not None and info is None), it
        if hb is not None:
            want = jpeg.prepare_batch([info])
            assert all(torch.equal(getattr(hb, k), getattr(want, k)) for k in ("img32", "img64", "tables", "seg32", "seg64", "words")), it
        stricter += hb is None and info is not None
    assert stricter < 30
    # crafted DHT (round-5 advisor finding): BITS = [255, 0, ...] with 255 values passes the "sum of BITS == number of values" test but is
    # no prefix code; the derived-table builder used to index its 512-entry look-ahead table at ~65000 (heap overflow in the batch's
    # `tables` tensor).  The native walk must refuse the file (status -3) and the Python walk must refuse it too.
    b = open(good[0], "rb").read()
    i = b.rfind(b"\xff\xc4")
    L = (b[i + 2] << 8) | b[i + 3]
    assert i > 0 and L > 19
    seg = bytes([b[i + 4]]) + bytes([255] + [0] * 15) + bytes(range(255))
    crafted = b[:i] + b"\xff\xc4" + bytes([(len(seg) + 2) >> 8, (len(seg) + 2) & 255]) + seg + b[i + 2 + L:]
    cp = tmp_path / "crafted_dht.jpg"
    cp.write_bytes(crafted)
    for _ in range(3):
        hb, st = jpeg.prepare_files([str(cp)] * 4, threads=2)
        assert hb is None and st == [-3] * 4, st
    with pytest.raises(jpeg.JpegUnsupported):
        jpeg.prepare_batch([jpeg.parse(crafted)])
    # a table that over-subscribes a LONGER length only (two 1-bit codes are fine, three 2-bit codes on top are not)
    seg = bytes([b[i + 4]]) + bytes([2, 3] + [0] * 14) + bytes(range(5))
    crafted = b[:i] + b"\xff\xc4" + bytes([0, len(seg) + 2]) + seg + b[i + 2 + L:]
    cp.write_bytes(crafted)
    hb, st = jpeg.prepare_files([str(cp)], threads=1)
    assert hb is None and st == [-3], st
