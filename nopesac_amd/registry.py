"""detectron2-style registries + `configurable`, so the drop-in plugs in under the reference's names:

    META_ARCH_REGISTRY["PlaneTR_NopeSAC"]        (meta_arch/siamese_planeTR.py:33-34)
    BACKBONE_REGISTRY["build_resnet_backbone"]   (configs/Base.yaml:4)
    SEM_SEG_HEADS_REGISTRY["PlaneTRHead"]        (planeTR_net/planeTR_head.py:17-23)
    MATCHING_HEAD_REGISTRY["MatchingHead"]       (matching_net/matching_head.py:15-24)
    CAMERA_HEAD_REGISTRY["PlaneCameraHead"]      (camera_net/camera_head.py:21-35)

If detectron2 is importable the meta-arch is ALSO registered into detectron2's own META_ARCH_REGISTRY
(see INTEGRATION.md), so `MODEL.META_ARCHITECTURE: "PlaneTR_NopeSAC"` resolves to this implementation.
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


def configurable(init_func):
    """`cls(cfg, *args)` -> `cls(**cls.from_config(cfg, *args))`; explicit kwargs pass straight through."""
    from .config import CfgNode

    @functools.wraps(init_func)
    def wrapped(self, *args, **kwargs):
        if (args and isinstance(args[0], CfgNode)) or isinstance(kwargs.get("cfg"), CfgNode) and not args:
            from_config = type(self).from_config
            assert inspect.ismethod(from_config), "from_config must be a classmethod"
            init_func(self, **from_config(*args, **kwargs))
        else:
            init_func(self, *args, **kwargs)

    return wrapped


def build_model(cfg):
    """detectron2.modeling.build_model: registry lookup by MODEL.META_ARCHITECTURE, then .to(MODEL.DEVICE)."""
    import torch
    from . import modeling  # noqa: F401  (registers everything)
    model = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)
    model.to(torch.device(cfg.MODEL.DEVICE))
    return model
