"""Throw-away import shim: lets the *reference* (pure Python, /root/reference) run on CPU in
THIS container so that golden vectors can be captured from it (oracle/gen_golden.py).

TEST INFRASTRUCTURE ONLY.  Nothing here is imported by the product (nopesac_amd/), and
nothing here can run on the GPU box (there is no /root/reference there).

The reference depends on detectron2==0.4 (README.md:28), fvcore, numpy-quaternion, cv2,
pycocotools and torchvision, none of which exist in this image.  The stubs below restate
just enough of their *semantics* (written from knowledge of
detectron2 v0.4; its source is not in this container) for
NopeSAC_Net.modeling.meta_arch.siamese_planeTR.PlaneTR_NopeSAC to build and run in eval
mode.  The ResNet-50 backbone, which lives inside detectron2, is restated in
oracle/d2_resnet.py and injected as `build_backbone`.
"""
from __future__ import annotations

import functools
import inspect
import os
import pickle
import sys
import tempfile
import types

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

REFERENCE_ROOT = os.environ.get("NOPESAC_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "NopeSAC_Net", "modeling"))


# --------------------------------------------------------------------------------------
# detectron2.config
# --------------------------------------------------------------------------------------
class CfgNode(dict):
    """Attribute dict with the small part of yacs behaviour the reference touches."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:  # pragma: no cover
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def configurable(init_func=None, *, from_config=None):
    """d2 semantics: ctor called with a CfgNode first positional/`cfg` kw => from_config."""
    assert init_func is not None and from_config is None

    @functools.wraps(init_func)
    def wrapped(self, *args, **kwargs):
        cfg = None
        if args and isinstance(args[0], CfgNode):
            cfg = args[0]
        elif isinstance(kwargs.get("cfg"), CfgNode) and not args:
            cfg = kwargs["cfg"]
        if cfg is not None:
            fc = type(self).from_config
            if args and isinstance(args[0], CfgNode):
                explicit = fc(*args, **kwargs)
            else:
                explicit = fc(**kwargs)
            init_func(self, **explicit)
        else:
            init_func(self, *args, **kwargs)

    return wrapped


# --------------------------------------------------------------------------------------
# detectron2.utils.registry
# --------------------------------------------------------------------------------------
class Registry:
    def __init__(self, name):
        self._name = name
        self._map = {}

    def register(self, obj=None):
        if obj is None:
            def deco(o):
                self._map[o.__name__] = o
                return o
            return deco
        self._map[obj.__name__] = obj
        return obj

    def get(self, name):
        return self._map[name]


# --------------------------------------------------------------------------------------
# detectron2.layers
# --------------------------------------------------------------------------------------
class ShapeSpec:
    def __init__(self, channels=None, height=None, width=None, stride=None):
        self.channels, self.height, self.width, self.stride = channels, height, width, stride


class FrozenBatchNorm2d(nn.Module):
    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.num_features = num_features
        self.eps = eps
        self.register_buffer("weight", torch.ones(num_features))
        self.register_buffer("bias", torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features) - eps)

    def forward(self, x):
        scale = self.weight * (self.running_var + self.eps).rsqrt()
        bias = self.bias - self.running_mean * scale
        return x * scale.reshape(1, -1, 1, 1) + bias.reshape(1, -1, 1, 1)

    @classmethod
    def convert_frozen_batchnorm(cls, module):
        return module


def get_norm(norm, out_channels):
    if norm is None or norm == "":
        return None
    if norm == "GN":
        return nn.GroupNorm(32, out_channels)
    if norm == "FrozenBN":
        return FrozenBatchNorm2d(out_channels)
    if norm == "BN":
        return nn.BatchNorm2d(out_channels)
    raise KeyError(norm)


class Conv2d(nn.Conv2d):
    """nn.Conv2d + optional .norm + optional .activation (applied in that order)."""

    def __init__(self, *args, **kwargs):
        norm = kwargs.pop("norm", None)
        activation = kwargs.pop("activation", None)
        super().__init__(*args, **kwargs)
        self.norm = norm
        self.activation = activation

    def forward(self, x):
        x = F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)
        if self.norm is not None:
            x = self.norm(x)
        if self.activation is not None:
            x = self.activation(x)
        return x


# --------------------------------------------------------------------------------------
# misc d2 pieces
# --------------------------------------------------------------------------------------
class ImageList:
    def __init__(self, tensor, image_sizes):
        self.tensor = tensor
        self.image_sizes = image_sizes

    @staticmethod
    def from_tensors(tensors, size_divisibility=0, pad_value=0.0):
        assert size_divisibility in (0, 1)
        sizes = [tuple(t.shape[-2:]) for t in tensors]
        assert len(set(sizes)) == 1, "shim supports equal-size images only"
        return ImageList(torch.stack(tensors, 0), sizes)


class Backbone(nn.Module):
    size_divisibility = 0


class _MetadataCatalog:
    def get(self, name):
        return types.SimpleNamespace()


def _mask_encode(m):
    """pycocotools.mask.encode on a Fortran-ordered HxW uint8/bool array -> uncompressed counts
    (the compressed string format is irrelevant to the hot path; only toBbox consumes it here)."""
    m = np.asarray(m)
    h, w = m.shape
    flat = m.reshape(-1, order="F").astype(np.uint8)
    counts, prev, run = [], 0, 0
    for v in flat:
        if v != prev:
            counts.append(run)
            run, prev = 0, v
        run += 1
    counts.append(run)
    return {"size": [h, w], "counts": counts, "_mask": m.astype(bool)}


def _mask_to_bbox(rle):
    m = rle["_mask"]
    ys, xs = np.nonzero(m)
    if len(xs) == 0:
        return np.zeros(4)
    x0, x1, y0, y1 = xs.min(), xs.max() + 1, ys.min(), ys.max() + 1
    return np.array([x0, y0, x1 - x0, y1 - y0], dtype=np.float64)


def _mask_iou(dt, gt, iscrowd):
    """pycocotools.mask.iou (cocoapi rleIou) restated by MERGING RUNS (the product decodes to dense masks: two independent
    routes to the same number).  dt / gt: RLE dicts with uncompressed `counts` lists (what _mask_encode produces)."""
    out = np.zeros((len(dt), len(gt)), np.float64)
    for i, d in enumerate(dt):
        for j, g in enumerate(gt):
            ca, cb = list(d["counts"]), list(g["counts"])
            ia = ib = 0
            ra, rb = ca[0], cb[0]
            va = vb = 0
            inter = union = 0
            while True:
                step = min(ra, rb)
                if va and vb:
                    inter += step
                if va or vb:
                    union += step
                ra -= step; rb -= step
                if ra == 0:
                    ia += 1
                    if ia == len(ca):
                        break
                    ra, va = ca[ia], va ^ 1
                if rb == 0:
                    ib += 1
                    if ib == len(cb):
                        break
                    rb, vb = cb[ib], vb ^ 1
            if iscrowd[j]:
                union = sum(ca[1::2])
            out[i, j] = inter / union if union > 0 else 0.0
    return out


def load_reference_function(rel_path: str, name: str, namespace: dict):
    """Compile ONE top-level function of a reference source file in `namespace` (this container only; nothing is written to the
    repo): used for functions whose module cannot be imported because of unrelated heavy imports (COCO tooling, visualisers)."""
    import ast
    path = os.path.join(REFERENCE_ROOT, rel_path)
    tree = ast.parse(open(path).read(), filename=path)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name]
    assert len(fn) == 1, (name, path)
    mod = ast.Module(body=fn, type_ignores=[])
    exec(compile(mod, path, "exec"), namespace)
    return namespace[name]


_INSTALLED = False


def install(backbone_builder):
    """Insert the stub modules into sys.modules and make /root/reference importable."""
    global _INSTALLED
    if _INSTALLED:
        return
    if not reference_available():
        raise RuntimeError("reference tree not found at %s" % REFERENCE_ROOT)

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    META_ARCH_REGISTRY = Registry("META_ARCH")

    def _c2_fill(module):  # init only; weights are overwritten by the synthetic checkpoint
        return None

    d2 = mod("detectron2")
    d2.config = mod("detectron2.config", CfgNode=CfgNode, configurable=configurable)
    d2.data = mod("detectron2.data", MetadataCatalog=_MetadataCatalog())
    d2.modeling = mod(
        "detectron2.modeling",
        META_ARCH_REGISTRY=META_ARCH_REGISTRY,
        build_backbone=backbone_builder,
        build_sem_seg_head=lambda cfg, shape: None,
    )
    mod("detectron2.modeling.backbone", Backbone=Backbone)
    mod("detectron2.modeling.postprocessing", sem_seg_postprocess=lambda *a, **k: None)
    mod("detectron2.structures", ImageList=ImageList)
    mod("detectron2.layers", Conv2d=Conv2d, ShapeSpec=ShapeSpec, get_norm=get_norm,
        FrozenBatchNorm2d=FrozenBatchNorm2d)
    d2.utils = mod("detectron2.utils")
    mod("detectron2.utils.registry", Registry=Registry)
    d2.utils.comm = mod("detectron2.utils.comm", is_main_process=lambda: True, get_rank=lambda: 0,
                        get_world_size=lambda: 1, synchronize=lambda: None)
    mod("detectron2.utils.logger", setup_logger=lambda *a, **k: None)
    fv = mod("fvcore")
    fv.nn = mod("fvcore.nn")
    fv.nn.weight_init = mod("fvcore.nn.weight_init", c2_xavier_fill=_c2_fill, c2_msra_fill=_c2_fill)
    mod("quaternion")
    mod("cv2")
    mod("torchvision", _is_tracing=lambda: False)
    pc = mod("pycocotools")
    pc.mask = mod("pycocotools.mask", encode=_mask_encode, toBbox=_mask_to_bbox)

    # matching_head.py hard-codes .cuda() (matching_head.py:274-298); identity on CPU.
    torch.Tensor.cuda = lambda self, *a, **k: self
    if not hasattr(np, "float"):
        np.float = float  # siamese_planeTR.py:727,776 uses the removed alias

    sys.dont_write_bytecode = True
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    _INSTALLED = True


def make_reference_cfg(overrides=None, num_queries=50):
    """cfg = d2 defaults touched by the hot path + config/config.py defaults +
    configs/inference_mp3d.yaml values (+ overrides as dotted keys)."""
    from NopeSAC_Net.config.config import get_sparseplane_cfg_defaults  # noqa: reference import

    C = CfgNode
    cfg = C(
        SOLVER=C(), TEST=C(), DATALOADER=C(), DATASETS=C(),
        MODEL=C(
            DEVICE="cpu",
            PIXEL_MEAN=[123.675, 116.280, 103.530],
            PIXEL_STD=[58.395, 57.120, 57.375],
            SEM_SEG_HEAD=C(NAME="PlaneTRHead", IN_FEATURES=["res2", "res3", "res4", "res5"],
                           NORM="GN", CONVS_DIM=128, COMMON_STRIDE=4, IGNORE_VALUE=255),
            BACKBONE=C(NAME="build_resnet_backbone", FREEZE_AT=0),
        ),
    )
    get_sparseplane_cfg_defaults(cfg)
    cfg.MODEL.META_ARCHITECTURE = "PlaneTR_NopeSAC"
    cfg.MODEL.MASK_ON = True
    cfg.MODEL.CAMERA_ON = True
    cfg.MODEL.EMBEDDING_ON = True
    cfg.MODEL.SEM_SEG_HEAD.NUM_CLASSES = 1
    cfg.MODEL.SEM_SEG_HEAD.PARAM_ON = True
    cfg.MODEL.SEM_SEG_HEAD.CENTER_ON = True
    cfg.MODEL.SEM_SEG_HEAD.NUM_OBJECT_QUERIES = num_queries
    cfg.MODEL.CAMERA_HEAD.REFINE_ON = True
    cfg.MODEL.CAMERA_HEAD.CAM_REC_ON = True
    cfg.MODEL.CAMERA_HEAD.INFERENCE_OUT_CAM_TYPE = "soft"
    cfg.MODEL.CAMERA_HEAD.NAME = "PlaneCameraHead"
    cfg.MODEL.CAMERA_HEAD.WARP_PLANE_IN_CAM_REF_ON = True
    cfg.TEST.MATCHING_SCORE_THRESHOLD = 0.2
    # camCls pickles do not unpickle here and are unused at inference
    dummy = os.path.join(tempfile.gettempdir(), "nopesac_dummy_kmeans.pkl")
    with open(dummy, "wb") as f:
        pickle.dump({"unused": True}, f)
    cfg.MODEL.CAMERA_HEAD.KMEANS_TRANS_PATH = dummy
    cfg.MODEL.CAMERA_HEAD.KMEANS_ROTS_PATH = dummy
    for k, v in (overrides or {}).items():
        node = cfg
        parts = k.split(".")
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = v
    return cfg


def build_reference_model(state_dict, overrides=None, num_queries=50):
    """Import the reference, build PlaneTR_NopeSAC on CPU, load `state_dict`, eval()."""
    from oracle.d2_resnet import build_resnet50_backbone

    install(build_resnet50_backbone)
    import NopeSAC_Net.modeling  # noqa: F401  (registers the meta-arch)
    from NopeSAC_Net.modeling.meta_arch.siamese_planeTR import PlaneTR_NopeSAC

    cfg = make_reference_cfg(overrides, num_queries)
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        model = PlaneTR_NopeSAC(cfg)
    # PlaneTRHead._load_from_state_dict (planeTR_head.py:26-48) renames keys unless the
    # checkpoint carries module-version metadata, as any checkpoint saved from the model does.
    from collections import OrderedDict
    sd = OrderedDict(state_dict)
    sd._metadata = model.state_dict()._metadata
    missing, unexpected = model.load_state_dict(sd, strict=False)
    missing = [k for k in missing if not k.startswith("criterion.")]
    assert not missing and not unexpected, (missing[:5], unexpected[:5])
    return model.eval()
