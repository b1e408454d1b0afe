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
