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
