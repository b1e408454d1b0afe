"""detectron2-style registries + `configurable`, so the drop-in plugs in under the reference's names:

    META_ARCH_REGISTRY["PlaneTR_NopeSAC"]        (meta_arch/siamese_planeTR.py:33-34)
    BACKBONE_REGISTRY["build_resnet_backbone"]   (configs/Base.yaml:4)
    SEM_SEG_HEADS_REGISTRY["PlaneTRHead"]        (planeTR_net/planeTR_head.py:17-23)
    MATCHING_HEAD_REGISTRY["MatchingHead"]       (matching_net/matching_head.py:15-24)
    CAMERA_HEAD_REGISTRY["PlaneCameraHead"]      (camera_net/camera_head.py:21-35)

`register_into_detectron2()` (called once when `nopesac_amd.modeling` is imported; a no-op without detectron2) puts the
meta-arch into detectron2's own META_ARCH_REGISTRY too, so that detectron2's
`build_model(cfg)` with `MODEL.META_ARCHITECTURE: "PlaneTR_NopeSAC"` resolves to this implementation.  `configurable`
accepts any yacs / detectron2-style config node (duck-typed), not only this package's CfgNode.
"""
from __future__ import annotations

import functools
import inspect


class Registry:
    def __init__(self, name: str):
        self._name = name
        self._obj_map = {}

    def _do_register(self, name, obj):
        assert name not in self._obj_map, f"An object named '{name}' was already registered in '{self._name}' registry!"
        self._obj_map[name] = obj

    def register(self, obj=None, name=None):
        if obj is None:
            def deco(fn):
                self._do_register(name or fn.__name__, fn)
                return fn
            return deco
        self._do_register(name or obj.__name__, obj)
        return obj

    def get(self, name):
        if name not in self._obj_map:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry!")
        return self._obj_map[name]

    def __contains__(self, name):
        return name in self._obj_map


META_ARCH_REGISTRY = Registry("META_ARCH")
BACKBONE_REGISTRY = Registry("BACKBONE")
SEM_SEG_HEADS_REGISTRY = Registry("SEM_SEG_HEADS")
MATCHING_HEAD_REGISTRY = Registry("MATCHING_HEAD")
CAMERA_HEAD_REGISTRY = Registry("CAMERA_HEAD")


def is_config_node(a) -> bool:
    """A yacs / fvcore / detectron2 / nopesac_amd config node, duck-typed: a mapping (or attribute bag) whose `MODEL` child
    names the META_ARCHITECTURE.  detectron2's own test is isinstance(yacs CfgNode | omegaconf DictConfig) - both qualify."""
    try:
        model = a["MODEL"] if isinstance(a, dict) else getattr(a, "MODEL", None)
        if model is None:
            return False
        return ("META_ARCHITECTURE" in model) if isinstance(model, dict) else hasattr(model, "META_ARCHITECTURE")
    except Exception:
        return False


def configurable(init_func):
    """detectron2.config.configurable for `__init__`: `cls(cfg, *args)` / `cls(cfg=cfg)` -> `cls(**cls.from_config(cfg, *args))`;
    a call with explicit keyword arguments (e.g. `cls(num_queries=..., cfg=cfg, ...)`) passes straight through
    (meta_arch/siamese_planeTR.py:38,133: the reference is built as `cls(cfg)` by detectron2's build_model)."""

    @functools.wraps(init_func)
    def wrapped(self, *args, **kwargs):
        via_cfg = (bool(args) and is_config_node(args[0])) or (not args and set(kwargs) == {"cfg"} and is_config_node(kwargs["cfg"]))
        if via_cfg:
            from_config = type(self).from_config
            if not inspect.ismethod(from_config):
                raise TypeError("from_config must be a classmethod")
            init_func(self, **from_config(*args, **kwargs))
        else:
            init_func(self, *args, **kwargs)

    return wrapped


def register_into_detectron2(override: bool = True) -> bool:
    """Put this package's classes into detectron2's registries under the reference's names.  Returns False when
    detectron2 is not importable.  With `override` an existing entry of the same name (the reference's own class, if
    NopeSAC_Net was imported first) is replaced - that IS the drop-in; without it an existing entry is left alone."""
    try:
        from detectron2.modeling import META_ARCH_REGISTRY as D2_META
    except Exception:
        return False
    from . import modeling  # noqa: F401  (fills the local registries)
    cls = META_ARCH_REGISTRY.get("PlaneTR_NopeSAC")
    existing = getattr(D2_META, "_obj_map", None)
    if existing is not None and "PlaneTR_NopeSAC" in existing:
        if not override or existing["PlaneTR_NopeSAC"] is cls:
            return existing["PlaneTR_NopeSAC"] is cls
        existing["PlaneTR_NopeSAC"] = cls
        return True
    D2_META.register(cls)
    return True


def build_model(cfg):
    """detectron2.modeling.build_model: registry lookup by MODEL.META_ARCHITECTURE, then .to(MODEL.DEVICE)."""
    import torch
    from . import modeling  # noqa: F401  (registers everything)
    model = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)
    model.to(torch.device(cfg.MODEL.DEVICE))
    return model
