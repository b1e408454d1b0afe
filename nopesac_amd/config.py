"""Config surface of the drop-in: a yacs/detectron2-compatible `CfgNode` plus the defaults the hot path
needs, so that the reference's configs/inference_*.yaml (with `_BASE_: Base.yaml`) and its
`KEY VALUE` command-line overrides load unchanged without detectron2/yacs installed.

Mirrors: detectron2.config.CfgNode semantics and
NopeSAC_Net/config/config.py:5-114 (`get_sparseplane_cfg_defaults`; key names and default values are
the contract, restated here as data).
"""
from __future__ import annotations

import ast
import copy
import os
from typing import Any, Iterable

import yaml

BASE_KEY = "_BASE_"


class CfgNode(dict):
    """Nested attribute dict: merge_from_file (honours _BASE_), merge_from_list, freeze, clone.
    Merging rejects keys that do not already exist (yacs behaviour), so defaults must be installed first
    (test_NopeSAC.py:183-186 calls get_sparseplane_cfg_defaults before merge_from_file)."""

    _FROZEN = "__frozen__"

    def __init__(self, init=None):
        super().__init__()
        object.__setattr__(self, CfgNode._FROZEN, False)
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    # attribute access
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        if object.__getattribute__(self, CfgNode._FROZEN):
            raise AttributeError(f"Attempted to set {k} on an immutable CfgNode")
        self[k] = v

    def is_frozen(self):
        return object.__getattribute__(self, CfgNode._FROZEN)

    def _set_frozen(self, flag):
        object.__setattr__(self, CfgNode._FROZEN, flag)
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_frozen(flag)

    def freeze(self):
        self._set_frozen(True)

    def defrost(self):
        self._set_frozen(False)

    def clone(self):
        c = copy.deepcopy(self)
        return c

    def __deepcopy__(self, memo):
        c = CfgNode()
        for k, v in self.items():
            dict.__setitem__(c, k, copy.deepcopy(v, memo))
        object.__setattr__(c, CfgNode._FROZEN, self.is_frozen())
        return c

    # merging
    @staticmethod
    def load_yaml_with_base(path: str) -> dict:
        with open(path) as f:
            cfg = yaml.safe_load(f) or {}
        if BASE_KEY in cfg:
            base = cfg.pop(BASE_KEY)
            if not os.path.isabs(base):
                base = os.path.join(os.path.dirname(path), base)
            merged = CfgNode.load_yaml_with_base(base)
            _merge_dicts(cfg, merged)
            return merged
        return cfg

    def merge_from_file(self, path: str):
        self._merge(CfgNode.load_yaml_with_base(path), [])

    def merge_from_other_cfg(self, other: "CfgNode"):
        self._merge(other, [])

    def _merge(self, other: dict, trail):
        if self.is_frozen():
            raise AttributeError("cannot merge into a frozen CfgNode")
        for k, v in other.items():
            full = ".".join(trail + [k])
            if k not in self:
                raise KeyError(f"Non-existent config key: {full}")
            if isinstance(v, dict):
                if not isinstance(self[k], CfgNode):
                    raise KeyError(f"config key {full} is not a node")
                self[k]._merge(v, trail + [k])
            else:
                self[k] = _coerce(_decode(v), self[k], full)

    def merge_from_list(self, opts: Iterable[Any]):
        opts = list(opts)
        assert len(opts) % 2 == 0, "override list must be KEY VALUE pairs"
        for key, val in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                if p not in node:
                    raise KeyError(f"Non-existent config key: {key}")
                node = node[p]
            if parts[-1] not in node:
                raise KeyError(f"Non-existent config key: {key}")
            node[parts[-1]] = _coerce(_decode(val), node[parts[-1]], key)


def _decode(v):
    """yacs `_decode_cfg_value`: strings that parse as Python literals become those literals
    (e.g. the yaml scalar '("mp3d_test",)' -> tuple); everything else is kept."""
    if not isinstance(v, str):
        return v
    try:
        return ast.literal_eval(v)
    except (ValueError, SyntaxError):
        return v


def _merge_dicts(src: dict, dst: dict):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge_dicts(v, dst[k])
        else:
            dst[k] = v


def _coerce(new, old, key):
    """yacs type rule: same type, or int->float, tuple<->list."""
    if old is None or new is None or type(new) == type(old):
        return new
    if isinstance(old, float) and isinstance(new, int):
        return float(new)
    if isinstance(old, tuple) and isinstance(new, list):
        return tuple(new)
    if isinstance(old, list) and isinstance(new, tuple):
        return list(new)
    if isinstance(old, str) and isinstance(new, str):
        return new
    raise ValueError(f"Type mismatch for config key {key}: {type(old).__name__} vs {type(new).__name__}")


CN = CfgNode


def _d2_defaults() -> CfgNode:
    """The detectron2==0.4 default keys that configs/Base.yaml / inference_*.yaml touch or the hot path
    reads (values = d2 defaults, restated from memory of detectron2/config/defaults.py)."""
    c = CN()
    c.VERSION = 2
    c.MODEL = CN()
    c.MODEL.DEVICE = "cuda"
    c.MODEL.META_ARCHITECTURE = "GeneralizedRCNN"
    c.MODEL.WEIGHTS = ""
    c.MODEL.MASK_ON = False
    c.MODEL.KEYPOINT_ON = False
    c.MODEL.LOAD_PROPOSALS = False
    c.MODEL.PIXEL_MEAN = [103.530, 116.280, 123.675]
    c.MODEL.PIXEL_STD = [1.0, 1.0, 1.0]
    c.MODEL.BACKBONE = CN({"NAME": "build_resnet_backbone", "FREEZE_AT": 2})
    c.MODEL.RESNETS = CN({
        "DEPTH": 50, "OUT_FEATURES": ["res4"], "NUM_GROUPS": 1, "NORM": "FrozenBN", "WIDTH_PER_GROUP": 64,
        "STRIDE_IN_1X1": True, "RES5_DILATION": 1, "RES2_OUT_CHANNELS": 256, "STEM_OUT_CHANNELS": 64,
        "DEFORM_ON_PER_STAGE": [False, False, False, False], "DEFORM_MODULATED": False, "DEFORM_NUM_GROUPS": 1})
    c.MODEL.SEM_SEG_HEAD = CN({
        "NAME": "SemSegFPNHead", "IN_FEATURES": ["p2", "p3", "p4", "p5"], "IGNORE_VALUE": 255, "NUM_CLASSES": 54,
        "CONVS_DIM": 128, "COMMON_STRIDE": 4, "NORM": "GN", "LOSS_WEIGHT": 1.0})
    c.INPUT = CN({"MIN_SIZE_TRAIN": (800,), "MIN_SIZE_TRAIN_SAMPLING": "choice", "MAX_SIZE_TRAIN": 1333,
                  "MIN_SIZE_TEST": 800, "MAX_SIZE_TEST": 1333, "RANDOM_FLIP": "horizontal", "FORMAT": "BGR",
                  "MASK_FORMAT": "polygon"})
    c.DATASETS = CN({"TRAIN": (), "TEST": (), "PROPOSAL_FILES_TRAIN": (), "PROPOSAL_FILES_TEST": ()})
    c.DATALOADER = CN({"NUM_WORKERS": 4, "ASPECT_RATIO_GROUPING": True, "SAMPLER_TRAIN": "TrainingSampler",
                       "REPEAT_THRESHOLD": 0.0, "FILTER_EMPTY_ANNOTATIONS": True})
    c.SOLVER = CN({
        "LR_SCHEDULER_NAME": "WarmupMultiStepLR", "MAX_ITER": 40000, "BASE_LR": 0.001, "MOMENTUM": 0.9, "NESTEROV": False,
        "WEIGHT_DECAY": 0.0001, "WEIGHT_DECAY_NORM": 0.0, "GAMMA": 0.1, "STEPS": (30000,), "WARMUP_FACTOR": 1.0 / 1000,
        "WARMUP_ITERS": 1000, "WARMUP_METHOD": "linear", "CHECKPOINT_PERIOD": 5000, "IMS_PER_BATCH": 16,
        "REFERENCE_WORLD_SIZE": 0, "BIAS_LR_FACTOR": 1.0, "WEIGHT_DECAY_BIAS": 0.0001,
        "CLIP_GRADIENTS": CN({"ENABLED": False, "CLIP_TYPE": "value", "CLIP_VALUE": 1.0, "NORM_TYPE": 2.0}),
        "AMP": CN({"ENABLED": False})})
    c.TEST = CN({"EXPECTED_RESULTS": [], "EVAL_PERIOD": 0, "DETECTIONS_PER_IMAGE": 100})
    c.OUTPUT_DIR = "./output"
    c.SEED = -1
    c.CUDNN_BENCHMARK = False
    c.VIS_PERIOD = 0
    return c


def add_nopesac_defaults(cfg: CfgNode) -> CfgNode:
    """Key-for-key counterpart of get_sparseplane_cfg_defaults (config/config.py:5-114)."""
    S, M, T = cfg.SOLVER, cfg.MODEL, cfg.TEST
    S.WEIGHT_DECAY_EMBED = 0.0
    S.OPTIMIZER = "ADAMW"
    S.BACKBONE_MULTIPLIER = 1.0
    S.SEM_SEG_HEAD_MULTIPLIER = 1.0
    S.PLANE_MATCHER_HEAD_MULTIPLIER = 1.0
    M.FREEZE = []
    M.DEPTH_ON = False
    M.EMBEDDING_ON = False
    M.CAMERA_ON = False
    M.MASK_ON = True
    M.HUNGARIAN_MATCHER_ON = True
    M.LOSS_DETECTION_ON = True
    M.LOSS_CAMERA_ON = False
    M.LOSS_EMB_ON = False
    H = M.SEM_SEG_HEAD
    for k, v in dict(DEEP_SUPERVISION=True, NO_OBJECT_WEIGHT=0.1, DICE_WEIGHT=1.0, MASK_WEIGHT=20.0, PARAM_WEIGHT_L1=0.5,
                     PARAM_WEIGHT_COS=10.0, PARAM_HM_WEIGHT_L1=0.5, PARAM_WEIGHT_Q=1.0, PARAM_WEIGHT_CENTER_INS=0.5,
                     PARAM_WEIGHT_ANGLE=0.0028, PARAM_WEIGHT_OFFSET=0.01, NUM_CLASSES=1, CENTER_ON=False, PARAM_ON=False,
                     PARAM_IN_MATCHER=True, NHEADS=8, ENC_LAYERS=6, DEC_LAYERS=6, NUM_OBJECT_QUERIES=50, MASK_DIM=256,
                     HIDDEN_DIM=256).items():
        H[k] = v
    M.CAMERA_BRANCH = "CACHED"
    M.CAMERA_HEAD = CN(dict(
        NAME="", LOSS_WEIGHT=1.0, KMEANS_TRANS_PATH="./camCls/kmeans_trans_32.pkl", KMEANS_ROTS_PATH="./camCls/kmeans_rots_32.pkl",
        TRANS_CLASS_NUM=32, ROTS_CLASS_NUM=32, FEATURE_SIZE=64, BACKBONE_FEATURE="res3", REFINE_ON=False, CAM_REC_ON=False,
        RAND_ON=False, PIXEL_CAM_FIX_ON=False, INFERENCE_OUT_CAM_TYPE="soft", INITIAL_CAM_WEIGHT=1.0, PLANE_CAM_WEIGHT=1.0,
        PLANE_CAM_WEIGHT_PREDPLANE=0.1, CLASSIFICATION_ON=False, INFERENCE_SP_TOPCAM_ON=False, INFERENCE_SP_TOPCAM_PATH="",
        WARP_PLANE_IN_CAM_REF_ON=True))
    M.MATCHING_HEAD = CN(dict(NAME="", INITIAL_CAM_ON=True, OFFSET_MULTIPLIER=4.0, NORMAL_MULTIPLIER=8.0))
    for k, v in dict(EVAL_GT_BOX=False, OVERLAP_THRESHOLD=0.6, PLANE_SCORE_THRESHOLD=0.6, MASK_PROB_THRESHOLD=0.5,
                     EVAL_FULL_SCENE=False, MATCHING_SCORE_THRESHOLD=0.2, POSE_REFINEMENT_WITH_GT_MATCHERS=False,
                     POSE_REFINEMENT_WITH_GT_NOISE_MATCHERS=False, POSE_REFINEMENT_WITH_GT_NOISE_MATCHERS_OFFSET_SCALE=0.1,
                     POSE_REFINEMENT_WITH_GT_NOISE_MATCHERS_NORMAL_SCALE=10.0).items():
        T[k] = v
    cfg.DATALOADER.ASPECT_RATIO_GROUPING = False
    cfg.DATALOADER.AUGMENTATION = False
    cfg.DEBUG_ON = False
    cfg.DEBUG_CAMERA_ON = False
    cfg.SEED = 42
    cfg.FIX_SEED = True
    cfg.DATASETS.ROOT_DIR = ""
    return cfg


def amd_options(cfg) -> CfgNode:
    """cfg.MODEL.AMD with every missing key filled from the defaults, as a detached node: the model reads its own options
    through this, so that a FOREIGN config (detectron2's CfgNode + the reference's get_sparseplane_cfg_defaults, which has
    no MODEL.AMD, possibly frozen) builds the model in its default configuration instead of failing."""
    out = add_amd_defaults(CN({"MODEL": {}})).MODEL.AMD
    model = cfg["MODEL"] if isinstance(cfg, dict) else cfg.MODEL
    given = model.get("AMD") if isinstance(model, dict) else getattr(model, "AMD", None)
    if given is not None:
        for k, v in (given.items() if isinstance(given, dict) else vars(given).items()):
            out[k] = v
    return out


def add_amd_defaults(cfg) -> CfgNode:
    """Keys that only this implementation has (the reference ignores them).  Works on any yacs-style node: call it on
    detectron2's cfg before merge_from_file / merge_from_list if MODEL.AMD.* overrides are wanted there."""
    cfg.MODEL.AMD = type(cfg.MODEL)(dict(
        COMPUTE_DTYPE="float32",      # "float32" (parity path) or "bfloat16" (dense convs on bf16 MFMA)
        OUTPUT_MASKS=True,            # decode pred_plane_masks [n,H,W] from the winner map for every image
        OUTPUT_RLE=True,              # COCO RLE "segmentation" + "bbox" in every `instances` entry (siamese_planeTR.py:703-720)
        USE_HIP_GRAPH=False,          # capture the static-shape forward of a batch once and replay it (no Python per launch)
        GRAPH_REPLAY="launches",      # how a captured forward is replayed: "launches" = the launch tape (csrc/tape.hip: the graph's
                                      # kernel nodes re-issued as plain launches on the caller's stream - batches in flight overlap
                                      # like eager ones); "graph" = hipGraphLaunch of the whole graph.  "launches" falls back to
                                      # "graph" (with a warning) if the runtime cannot read the captured graph back
        TWO_STREAMS=True,             # pixel pose net on a side HIP stream, overlapped with the plane head
        CHECK_FINITE=True,            # count Inf / NaN in the returned poses / plane parameters on the device; `model(...)` raises
                                      # FloatingPointError when the results are fetched (the reference drops into pdb instead)
        CPU_AFFINITY=True,            # runner with more than one rank on a host: pin rank r to the r-th of WORLD_SIZE equal shares of the
                                      # cores this process may use (contiguous ids - on the usual two-socket node the GPUs 0-3 / 4-7 and
                                      # the low / high core ids share a socket) and size the decode-thread pool to that share; without it
                                      # eight ranks each start min(32, cores) decoder threads on the same cores.  Unmeasured on hardware
        AUTOTUNE=True,                # runner (nopesac_amd/run.py), bfloat16 mode: before the first batch, time the library's equivalent
                                      # kernel configurations for every conv / GEMM shape of a (pairs-per-batch, 480, 640) forward and
                                      # keep the fastest (PlaneTR_NopeSAC.autotune: a few seconds; +10-15 % throughput, same results up
                                      # to the summation order).  Library users call model.autotune(pairs) themselves
        ROUTING_FILE="",              # JSON of kernel-routing decisions (ops.ConvTuner): loaded if it exists (listed shapes are not
                                      # re-measured), written back by rank 0 when the tuning pass measured new shapes
        BACKBONE_FP8=False,           # BASELINE config 5: the 3x3 convs of the res3-res5 bottlenecks (44 % of the backbone FLOPs, its
                                      # MFMA-bound layers) on the fp8 (e4m3fn) K = 64 MFMA: per-output-channel weight scales, static
                                      # per-layer activation scales (PlaneTR_NopeSAC.calibrate_fp8); needs COMPUTE_DTYPE bfloat16
        POSE_FP32_PARTS="aim",        # bfloat16 mode: stages of the camera head that keep f32 operands, space / comma separated, of
                                      # "decoder branches fc aim refine" (scripts/bf16_attribution.py tables what each one contributes
                                      # to the bf16 pose error against the fp32 path).  Default "aim" (round 5): the two re-embedding
                                      # MLPs take a 4- / 3-vector - rounding a unit quaternion to bf16 alone is up to 0.45 deg - and
                                      # cost nothing measurable in f32 (8 tiny launches per step): camera_initRec R 1.70 / 4.13 ->
                                      # 1.41 / 3.61 deg mean / max on the benchmark workload, same pairs/s (profiles/r5_c_aim_fp32_ab.txt)
    ))
    return cfg


def get_cfg() -> CfgNode:
    """d2 defaults + NopeSAC defaults + AMD keys: the node `merge_from_file(configs/inference_*.yaml)` expects."""
    return add_amd_defaults(add_nopesac_defaults(_d2_defaults()))
