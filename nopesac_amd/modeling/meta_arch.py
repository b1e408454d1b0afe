"""`PlaneTR_NopeSAC` — the drop-in META_ARCHITECTURE (meta_arch/siamese_planeTR.py:33-34, inference half
:338-473).  Same registry name, same `model(batched_inputs: list[dict]) -> list[dict]` contract, same
state-dict key names; the body runs on libnopesac_hip.so.

Differences by design (MI355X-first):
  * any number of pairs per call (the reference asserts batch == 1, :340): the 2B images of B pairs go
    through backbone / plane head / post-selection as one batch, the B pairs through the camera head;
  * no host synchronisation inside the forward: ragged plane / match counts stay on the device, results
    are fetched once at the end when the per-pair dicts are built;
  * training is out of scope (`forward` in training mode raises).
"""
from __future__ import annotations

import os
from typing import Dict, List

import numpy as np
import torch
from torch import nn

from .. import ops, rle
from ..config import amd_options
from ..registry import META_ARCH_REGISTRY, configurable
from .backbone import build_backbone
from .params import DERIVED_EPOCH
from .camera_head import build_camera_head
from .matching_head import build_matching_head
from .plane_head import build_planeTR_head, post_select

_DTYPES = {"float32": torch.float32, "bfloat16": torch.bfloat16}


@META_ARCH_REGISTRY.register()
class PlaneTR_NopeSAC(nn.Module):
    @configurable
    def __init__(self, *, num_queries: int, pixel_mean, pixel_std, device, cfg):
        super().__init__()
        self.cfg = cfg
        assert cfg.MODEL.MASK_ON and cfg.MODEL.EMBEDDING_ON and cfg.MODEL.CAMERA_ON, \
            "implemented: the inference configuration (MASK_ON, EMBEDDING_ON, CAMERA_ON; configs/inference_*.yaml)"
        self.backbone = build_backbone(cfg)
        self.sem_seg_head = build_planeTR_head(cfg, self.backbone.output_shape())
        self.matching_head = build_matching_head(cfg)
        self.camera_head_list = nn.ModuleList([build_camera_head(cfg, self.backbone.output_shape())])
        self.num_queries = num_queries
        self.register_buffer("pixel_mean", torch.tensor(pixel_mean, dtype=torch.float32), False)
        self.register_buffer("pixel_std", torch.tensor(pixel_std, dtype=torch.float32), False)
        amd = amd_options(cfg)            # MODEL.AMD.* with defaults filled in (a detectron2 cfg has no such node)
        self.compute_dtype = _DTYPES[amd.COMPUTE_DTYPE]
        self.output_masks = bool(amd.OUTPUT_MASKS)
        self.output_rle = bool(amd.OUTPUT_RLE)
        for mod in (self.sem_seg_head, self.matching_head, self.camera_head_list[0]):
            mod.gemm_dtype = self.compute_dtype     # bf16 => head GEMMs run f32-activation x bf16-weight MFMA
        self.infer_iter = 0
        self.two_streams = bool(amd.TWO_STREAMS)
        self.check_finite = bool(amd.CHECK_FINITE)
        self._side_stream = None
        # MODEL.AMD.USE_HIP_GRAPH: the whole static-shape forward of a batch as ONE hipGraph replay (~280 launches, 8 ms of Python /
        # ctypes launch time per 32-pair batch -> < 1 ms).  GRAPH_SLOTS independent captures (own input buffer, own outputs) are
        # used round-robin, so a caller may keep GRAPH_SLOTS - 1 earlier results un-packaged while it submits the next batch.
        self.use_hip_graph = bool(amd.USE_HIP_GRAPH)
        self.graph_replay = str(amd.GRAPH_REPLAY)
        assert self.graph_replay in ("launches", "graph"), self.graph_replay
        self.graph_slots = 2
        self.graph_fetch = os.environ.get("NOPESAC_GRAPH_FETCH", "1") != "0"      # the result fetch + RLE encode inside the captured graph
        self._graphs = {}
        # camCls k-means pickles (siamese_planeTR.py:119-128) are not needed for inference math

    # captured hipGraphs hold the addresses of the packed weights of the moment of capture: any change of the parameters
    # (checkpoint load, .to(), .half() ...) drops them - the next call of a shape re-captures
    def _apply(self, fn, *a, **k):
        self.__dict__["_graphs"] = {}
        return super()._apply(fn, *a, **k)

    @classmethod
    def from_config(cls, cfg):
        return {"num_queries": cfg.MODEL.SEM_SEG_HEAD.NUM_OBJECT_QUERIES, "pixel_mean": cfg.MODEL.PIXEL_MEAN,
                "pixel_std": cfg.MODEL.PIXEL_STD, "device": torch.device(cfg.MODEL.DEVICE), "cfg": cfg}

    @property
    def device(self):
        return self.pixel_mean.device

    def load_state_dict(self, state_dict, strict: bool = True):
        """Accepts a bare reference state dict or a detectron2 checkpoint {"model": sd}; the reference-only
        `criterion.*` entries are ignored."""
        if "model" in state_dict and isinstance(state_dict["model"], dict):
            state_dict = state_dict["model"]
        sd = {k: v for k, v in state_dict.items() if not k.startswith("criterion.")}
        self.__dict__["_graphs"] = {}                       # captured hipGraphs point at the old packed weights
        return super().load_state_dict(sd, strict=strict)

    # ------------------------------------------------------------------------------------------
    def forward(self, batched_inputs: List[dict]):
        if self.training:
            raise NotImplementedError("nopesac_amd implements the inference path only (call .eval())")
        self.infer_iter += 1
        with torch.no_grad():
            dev_out = self.forward_device(batched_inputs)
            return self.package(batched_inputs, dev_out)

    def preprocess_image(self, batched_inputs: List[dict]) -> torch.Tensor:
        """Stack the 2B images (all view-"0" images first), normalise, NCHW->NHWC (siamese_planeTR.py:534-542)."""
        imgs = [x["0"]["image"] for x in batched_inputs] + [x["1"]["image"] for x in batched_inputs]
        sizes = {tuple(i.shape) for i in imgs}
        assert len(sizes) == 1, "all images of a batch must share one size (size_divisibility 0, no padding)"
        x = self._to_device_batch(imgs)
        return ops.preprocess(x, self.pixel_mean, self.pixel_std, self.backbone.STEM_CIN_PAD, self.compute_dtype)

    def stack_images(self, batched_inputs: List[dict]) -> torch.Tensor:
        """The 2B raw images as one f32 NCHW tensor on the device (all view-"0" images first): the bf16 path hands this to the
        fused stem, which normalises while it loads (`forward_tensors(None, ..., raw_images=...)`)."""
        imgs = [x["0"]["image"] for x in batched_inputs] + [x["1"]["image"] for x in batched_inputs]
        sizes = {tuple(i.shape) for i in imgs}
        assert len(sizes) == 1, "all images of a batch must share one size (size_divisibility 0, no padding)"
        return self._to_device_batch(imgs)

    def _to_device_batch(self, imgs) -> torch.Tensor:
        """[2B,3,H,W] f32 on the device: every image is copied straight into its slice of ONE buffer (async from pinned host memory;
        no per-image device tensor, no torch.stack pass over the batch)."""
        return self._copy_images(imgs, torch.empty((len(imgs),) + tuple(imgs[0].shape), device=self.device, dtype=torch.float32))

    def _copy_images(self, imgs, out: torch.Tensor, staging: torch.Tensor = None) -> torch.Tensor:
        """imgs (host or device, float32 as the reference mapper makes them or uint8 as PairMapper(uint8=True) does) -> `out`
        (f32 [2B,3,H,W], device).  uint8 images cross PCIe as bytes (a quarter of the traffic) and are widened on the device."""
        assert len({i.dtype for i in imgs}) == 1, "all images of a batch must share one dtype (float32 or uint8)"
        if self.device.type == "cuda":
            cur = torch.cuda.current_stream()
            for im in imgs:                                   # device images decoded on another stream (data.LazyPairs): their memory
                if im.is_cuda:                                # must not be recycled there while this stream still reads it
                    im.record_stream(cur)
        whole = self._as_one_batch(imgs)                      # data.LazyPairs hands out views of ONE batch buffer (pinned host memory for
        if imgs[0].dtype == torch.uint8 and self.device.type == "cuda":         # PNG splits, HBM for GPU-decoded JPEG splits), in this order:
            if whole is not None and whole.is_cuda and staging is None:         # one copy instead of 2B (each ~15 us of launch-thread time) -
                return ops.u8_to_f32(whole, out)                                # or none at all when the bytes are in HBM already
            u8 = staging if staging is not None else torch.empty(out.shape, device=self.device, dtype=torch.uint8)
            if whole is not None:
                u8.copy_(whole, non_blocking=True)
            else:
                for k, im in enumerate(imgs):
                    u8[k].copy_(im, non_blocking=True)
            return ops.u8_to_f32(u8, out)
        if whole is not None and whole.dtype == out.dtype:
            out.copy_(whole, non_blocking=True)
            return out
        for k, im in enumerate(imgs):
            out[k].copy_(im, non_blocking=True)               # (a uint8 image on the CPU path is widened by copy_)
        return out

    @staticmethod
    def _as_one_batch(imgs):
        """[n, ...] view over `imgs` when they are tensors of one device lying back to back, in order, in one storage (what data.LazyPairs
        yields); None otherwise."""
        a = imgs[0]
        if not a.is_contiguous() or len(imgs) < 2:
            return None
        nb, base, st = a.numel() * a.element_size(), a.data_ptr(), a.untyped_storage().data_ptr()
        for k, im in enumerate(imgs):
            if (im.device != a.device or im.dtype != a.dtype or im.shape != a.shape or not im.is_contiguous() or im.data_ptr() != base + k * nb
                    or im.untyped_storage().data_ptr() != st):
                return None
        return torch.as_strided(a, (len(imgs),) + tuple(a.shape), (a.numel(),) + tuple(a.stride()))

    def forward_device(self, batched_inputs: List[dict], diagnostics: bool = False, forced: dict = None) -> dict:
        """All device work for B pairs; returns device tensors only (no synchronisation)."""
        B = len(batched_inputs)
        H, W = batched_inputs[0]["0"]["image"].shape[-2:]
        if (self.use_hip_graph and not diagnostics and self.device.type == "cuda" and self.compute_dtype == torch.bfloat16
                and self.backbone.fused_stem and getattr(self, "stage_events", None) is None and not ops.TUNER.measuring):
            d = self._forward_graph(batched_inputs, B, H, W, forced)
            if d.get("static_outputs"):
                d = dict(d)                                    # per call: the fetch below belongs to THIS batch, not to the slot
        elif self.compute_dtype == torch.bfloat16 and self.backbone.fused_stem:
            d = self.forward_tensors(None, B, H, W, diagnostics, forced=forced, raw_images=self.stack_images(batched_inputs))
        else:
            d = self.forward_tensors(self.preprocess_image(batched_inputs), B, H, W, diagnostics, forced=forced)
        if self.device.type == "cuda":
            d["fetch"] = self._enqueue_fetch(d)
        return d

    def _fetch_device_work(self, d: dict, static: bool = False, hosts=None, rle_args=None) -> dict:
        """The device side of the fetch: the small result tensors -> pinned host memory (ops.HostFetch: one kernel), the COCO RLE
        strings (rle.PendingRLE).  static = True: called INSIDE the graph capture - the fetch is replayed with the graph / tape (the
        ~30 eager launches behind every replay were 0.3 ms of host-bound tail per one-pair call) into the pinned buffers `hosts`
        (allocated before the capture; sizes from the slot's eager warm-up pass)."""
        sel, cam = d["sel"], d["cam"]
        need = {"n_kept": sel["n_kept"], "kept_idx": sel["kept_idx"], "planes": sel["planes"], "centers": sel["centers"],
                "scores": sel["scores"], "areas": sel["areas"], "flags": sel["flags"], "m": cam["m"],
                "onepp_t": cam["refine"]["maps"]["trans_all"], "onepp_r": cam["refine"]["maps"]["rots_all"]}
        for k, (t, r) in cam["cameras"].items():
            need["cam_t:" + k], need["cam_r:" + k] = t, r
        for k in ("pred_assignment_beforeRef0", "pred_assignment_afterRef0", "pred_assignment"):
            need["ass:" + k] = cam[k]
        if "nonfinite" in cam:
            need["nonfinite"] = cam["nonfinite"]
        hosts = hosts or (None, None)
        f = {"small": ops.HostFetch(need, private_views=static, host=hosts[0])}
        if self.output_rle:
            f["rle"] = rle.PendingRLE(*(rle_args or (sel["winner"], sel["kept_idx"], sel["n_kept"], sel["flags"])), host=hosts[1])
        return f

    def _enqueue_fetch(self, d: dict) -> dict:
        """Everything `package()` needs on the host, enqueued NOW on the batch's stream behind its forward: the small result tensors
        (one kernel into pinned memory), the COCO RLE strings (rle.PendingRLE) and - graph mode - private copies of
        the device tensors the result dicts hand out; then an event.  package() waits for that event and does host work only.
        With several batches in flight this matters: device work issued at fetch time queues behind the other batches' launches
        (6.8 ms of the 9.7 ms package() took per 32-pair step were that wait)."""
        sel = d["sel"]
        sf = d.get("static_fetch")
        if sf is not None and (("rle" in sf) or not self.output_rle):
            f = {k: v for k, v in sf.items() if k != "rle" or self.output_rle}      # recorded in the graph: this replay has run it
            if d.get("static_outputs"):
                f["sel"] = dict(sel, feats=sel["feats"].clone(), winner=sel["winner"].clone())
        else:
            rs = None
            if d.get("static_outputs"):       # hipGraph mode: these device tensors are overwritten by the slot's next replay
                rs = dict(sel, feats=sel["feats"].clone(), winner=sel["winner"].clone())
            # (graph mode: the overflow fallback of PendingRLE.finish re-reads its inputs - hand it the private winner map; kept_idx /
            #  n_kept / flags are small and cloned here for the same reason)
            f = self._fetch_device_work(d, rle_args=None if rs is None or not self.output_rle else
                                        (rs["winner"], sel["kept_idx"].clone(), sel["n_kept"].clone(), sel["flags"].clone()))
            if rs is not None:
                f["sel"] = rs
            if d.get("_warm") is not None:     # the slot's eager warm-up pass: the capture that follows allocates pinned buffers of these sizes
                d["_warm"]["fetch_sizes"] = (f["small"].host_bytes(), f["rle"].fetch.host_bytes() if "rle" in f else None)
        f["ready"] = torch.cuda.Event()
        f["ready"].record()
        if d.get("static_outputs"):       # ... which must not start before the copies above have run (whatever stream it is issued on)
            d["_owner"]["clone_done"] = f["ready"]
        return f

    def forward_tensors(self, x_nhwc: torch.Tensor, B: int, H: int, W: int, diagnostics: bool = False,
                        forced: dict = None, raw_images: torch.Tensor = None) -> dict:
        """x_nhwc: the preprocessed batch (NHWC, channel-padded) - or None with `raw_images` (f32 NCHW [2B,3,H,W], bf16 mode)."""
        mark = self._mark
        mark("start")
        is_cuda = (x_nhwc if x_nhwc is not None else raw_images).is_cuda
        dev = (x_nhwc if x_nhwc is not None else raw_images).device
        feats = self.backbone(x_nhwc, raw=None if raw_images is None else (raw_images, self.pixel_mean, self.pixel_std))
        mark("backbone")
        head = self.camera_head_list[0]
        pose = None
        if self.two_streams and is_cuda:
            # the pixel pose net depends only on the backbone maps: run it on a side HIP stream, concurrently
            # with the (launch-latency-bound) transformer of the plane head
            main = torch.cuda.current_stream()
            if self._side_stream is None:
                self._side_stream = {}
            if main.cuda_stream not in self._side_stream:      # one side stream per caller stream (batches may be pipelined)
                self._side_stream[main.cuda_stream] = torch.cuda.Stream(device=dev)
            side = self._side_stream[main.cuda_stream]
            side.wait_stream(main)
            with torch.cuda.stream(side):
                pose = head.initial_pose(feats, B)
                if not torch.cuda.is_current_stream_capturing():
                    for t in pose:
                        t.record_stream(main)
        self.sem_seg_head.mark = mark if getattr(self, "stage_events", None) is not None else None
        head_out, query_feat = self.sem_seg_head(feats, want_logits=diagnostics)
        mark("plane_head")
        sel = post_select(head_out, query_feat, H, W, self.cfg)
        mark("post_select")
        forced_A = None
        if forced is not None:
            sel, forced_A = self._force_k(sel, head_out, query_feat, forced, B)
        if pose is not None:
            torch.cuda.current_stream().wait_stream(side)
        cam = head(feats, sel, self.matching_head, B, diagnostics, forced_assignment=forced_A, pose=pose, mark=mark)
        mark("refine")
        if self.check_finite and is_cuda:
            # Inf / NaN anywhere in what the caller gets back (poses of every stage, kept plane parameters): one int32 on the device
            cam["nonfinite"] = ops.count_nonfinite([t for pair in cam["cameras"].values() for t in pair] + [sel["planes"]])
        return {"B": B, "H": H, "W": W, "sel": sel, "cam": cam, "head_out": head_out if diagnostics else None,
                "feats": feats if diagnostics else None, "query_feat": query_feat if diagnostics else None}

    def _forward_graph(self, batched_inputs: List[dict], B: int, H: int, W: int, forced: dict = None) -> dict:
        """forward_device through a captured hipGraph (MODEL.AMD.USE_HIP_GRAPH): per (B, H, W, K control) and slot, the first call
        runs eagerly (warm-up: weight packing, kernel attributes), the second captures, later calls copy the host images into the
        slot's static input buffer and replay.  The returned tensors are the slot's static outputs: valid until the slot is
        replayed again (graph_slots calls later)."""
        imgs = [x["0"]["image"] for x in batched_inputs] + [x["1"]["image"] for x in batched_inputs]
        assert len({tuple(i.shape) for i in imgs}) == 1, "all images of a batch must share one size (size_divisibility 0, no padding)"
        slot = self.infer_iter % self.graph_slots
        key = (B, H, W, None if forced is None else id(forced), slot)
        epoch = (DERIVED_EPOCH[0], len(ops.TUNER.best))
        if self._graphs and self.__dict__.get("_graph_epoch") != epoch:
            # packed weights re-built (ParamModule.invalidate / load), fp8 scales re-calibrated or new kernel routing decisions since
            # the capture: the recorded launches point at stale tensors / kernels - drop every slot's graph and capture again
            torch.cuda.synchronize()
            self._graphs = {}
        self.__dict__["_graph_epoch"] = epoch
        st = self._graphs.get(key)
        if st is None:
            st = self._graphs[key] = {"in": torch.empty((2 * B, 3, H, W), device=self.device, dtype=torch.float32), "graph": None, "out": None,
                                      "calls": 0}
        buf = st["in"]
        if imgs[0].dtype == torch.uint8 and "in_u8" not in st:
            st["in_u8"] = torch.empty(buf.shape, device=self.device, dtype=torch.uint8)
        if st.get("clone_done") is not None:
            # the slot's previous replay (its stem still reads `buf`) and the copy-out of its results must be complete before the new
            # images overwrite the static input buffer - also when the caller rotates its slots across streams
            torch.cuda.current_stream().wait_event(st.pop("clone_done"))
        if st.get("fetch_pending"):
            # the slot's recorded fetch writes the SAME pinned host buffers on every replay: replaying it before package() has
            # snapshotted the previous batch's results would silently hand that caller this batch's numbers
            raise RuntimeError("PlaneTR_NopeSAC: graph slot %d is replayed before the results of its previous batch were packaged - more "
                               "batches in flight than model.graph_slots (%d); raise model.graph_slots to the in-flight depth" % (slot, self.graph_slots))
        self._copy_images(imgs, buf, st.get("in_u8"))
        st["calls"] += 1
        if st["graph"] is not None:
            if st["out"].get("static_fetch") is not None:
                st["fetch_pending"] = True
            if st.get("tape") is not None:
                st["tape"].replay(sides=self._tape_sides())   # the recorded launches, on the caller's current stream
            else:
                st["graph"].replay()
            return st["out"]
        if st["calls"] == 1:                                   # warm-up pass, eager
            out = self.forward_tensors(None, B, H, W, forced=forced, raw_images=buf)
            out["_warm"] = st
            return out
        cur = torch.cuda.current_stream()
        want_tape = self.graph_replay == "launches"
        try:
            g = torch.cuda.CUDAGraph(keep_graph=True) if want_tape else torch.cuda.CUDAGraph()
        except TypeError:                                      # a torch without keep_graph: whole-graph replay only
            g, want_tape = torch.cuda.CUDAGraph(), False
        cap = torch.cuda.Stream(device=self.device)
        cap.wait_stream(cur)
        sizes = st.get("fetch_sizes") if self.graph_fetch else None
        hosts = None if sizes is None else tuple(None if n is None else torch.empty(n, dtype=torch.uint8, pin_memory=True) for n in sizes)
        with torch.cuda.graph(g, stream=cap):
            out = self.forward_tensors(None, B, H, W, forced=forced, raw_images=buf)
            if hosts is not None and (hosts[1] is not None or not self.output_rle):
                out["static_fetch"] = self._fetch_device_work(out, static=True, hosts=hosts)
        cur.wait_stream(cap)
        out["static_outputs"] = True                           # package() must not hand out views of graph-owned memory
        out["_owner"] = st
        st["graph"], st["out"], st["tape"] = g, out, None
        if want_tape:
            from ..tape import LaunchTape, TapeUnsupported
            try:
                st["tape"] = LaunchTape(g, max_streams=int(self.__dict__.get("tape_streams", 4)))
                self.__dict__["tape_counts"] = dict(st["tape"].counts)
            except TapeUnsupported as e:
                import warnings
                warnings.warn("nopesac_amd: launch tape unavailable, replaying the whole hipGraph instead: %s" % (e,))
                self.__dict__["tape_error"] = str(e)
        if st["tape"] is not None:
            st["tape"].replay(sides=self._tape_sides())       # capture does not execute: run this batch
        else:
            if hasattr(g, "instantiate") and want_tape:
                g.instantiate()                                # (keep_graph=True defers the instantiation)
            g.replay()
        return out

    def _tape_sides(self):
        """The side stream this model would use for a batch on the current stream (bound by streams.StreamSet.bind or created by an
        earlier eager call): the tape's side chain runs there, on a hardware queue the caller chose."""
        side = (self._side_stream or {}).get(torch.cuda.current_stream().cuda_stream)
        return [side] if side is not None else None

    def calibrate_fp8(self, batched_inputs: List[dict]) -> dict:
        """Static activation scales of the fp8 backbone mode (MODEL.AMD.BACKBONE_FP8) from representative pairs; returns them."""
        with torch.no_grad():
            scales = self.backbone.calibrate_fp8(self.preprocess_image(batched_inputs))
        DERIVED_EPOCH[0] += 1                                  # captured graphs hold the old scales / fp8 weight copies
        return scales

    def autotune(self, pairs: int, height: int = 480, width: int = 640) -> int:
        """One dedicated single-stream forward on zeros that lets ops.TUNER pick, per conv/GEMM shape of this
        (batch, resolution), the fastest of the library's equivalent kernel configurations.  Returns the number of
        shapes whose choice differs from the built-in heuristic."""
        dev = self.device
        x = torch.zeros(2 * pairs, height, width, self.backbone.STEM_CIN_PAD, device=dev, dtype=self.compute_dtype)
        two, self.two_streams = self.two_streams, False
        ops.TUNER.measuring = True
        try:
            with torch.no_grad():
                self.forward_tensors(x, pairs, height, width)
            torch.cuda.synchronize(dev)
        finally:
            ops.TUNER.measuring = False
            self.two_streams = two
        return sum(1 for v in ops.TUNER.best.values() if v)

    def _mark(self, name: str):
        """Stage boundary marker: records a HIP event on the current stream when `self.stage_events` is a list
        (bench.py --stages); a no-op otherwise."""
        ev = getattr(self, "stage_events", None)
        if ev is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            ev.append((name, e))

    def _force_k(self, sel: dict, head_out: dict, query_feat: torch.Tensor, forced: dict, B: int):
        """BENCHMARK-ONLY K control: with random weights the threshold-based selection keeps
        ~1 plane per view, so the stages after post-selection would see K = 1.  The selection kernels still
        run (their outputs are overwritten): every view gets its K highest-scoring queries, view-2 appearance
        = permuted view-1 appearance + 1% noise, plane parameters and the K matches come from `forced`."""
        K = forced["K"]
        # one launch (csrc/ransac.hip: force_k_select_kernel; the first version was 18 torch launches inside the timed region)
        feats, n_kept = ops.force_k_select(head_out["pred_logits"].contiguous(), query_feat.contiguous(), forced["perm"], forced["noise"], B, K)
        out = dict(sel)
        out["feats"] = feats
        out["planes"] = forced["planes"]
        out["n_kept"] = n_kept
        return out, forced["assignment"]

    # ------------------------------------------------------------------------------------------
    def package(self, batched_inputs: List[dict], d: dict) -> List[dict]:
        """Build the reference's per-pair result dicts (siamese_planeTR.py:384-450); the only host sync."""
        B, sel, cam = d["B"], d["sel"], d["cam"]
        H, W = d["H"], d["W"]
        f = d.get("fetch")
        if f is None and sel["planes"].is_cuda:                # a `d` straight from forward_tensors: fetch now
            f = self._enqueue_fetch(d)
        if f is not None:
            f["ready"].synchronize()                           # the only host wait
            hostd = f["small"].views()
            sel = f.get("sel", sel)
            if d.get("static_fetch") is not None and d.get("_owner") is not None and "rle" not in f:
                d["_owner"]["fetch_pending"] = False           # (with RLE strings: cleared below, once they are copied out too)
        else:
            hostd = {k: v for k, v in (("n_kept", sel["n_kept"]), ("kept_idx", sel["kept_idx"]), ("planes", sel["planes"]),
                                       ("centers", sel["centers"]), ("scores", sel["scores"]), ("areas", sel["areas"]),
                                       ("flags", sel["flags"]), ("m", cam["m"]), ("onepp_t", cam["refine"]["maps"]["trans_all"]),
                                       ("onepp_r", cam["refine"]["maps"]["rots_all"]))}
            for k, (t, r) in cam["cameras"].items():
                hostd["cam_t:" + k], hostd["cam_r:" + k] = t, r
            for k in ("pred_assignment_beforeRef0", "pred_assignment_afterRef0", "pred_assignment"):
                hostd["ass:" + k] = cam[k]
            if "nonfinite" in cam:
                hostd["nonfinite"] = cam["nonfinite"]
        n_kept, kept_idx = hostd["n_kept"].tolist(), hostd["kept_idx"]
        planes, centers, scores, areas = hostd["planes"], hostd["centers"], hostd["scores"], hostd["areas"]
        flags = hostd["flags"].tolist()
        cams = {k: (hostd["cam_t:" + k].numpy(), hostd["cam_r:" + k].numpy()) for k in cam["cameras"]}
        m = hostd["m"].tolist()
        if "nonfinite" in cam and int(hostd["nonfinite"][0]) != 0:
            raise FloatingPointError("nopesac_amd: %d non-finite values in the predicted poses / plane parameters of this batch "
                                     "(the reference traps this case with pdb.set_trace(), camera_head.py:1072-1074)" % int(hostd["nonfinite"][0]))
        ass = {k: hostd["ass:" + k] for k in ("pred_assignment_beforeRef0", "pred_assignment_afterRef0", "pred_assignment")}
        onepp_t, onepp_r = hostd["onepp_t"].numpy(), hostd["onepp_r"].numpy()
        rles = None
        if self.output_rle:
            rles = (f["rle"].finish(n_kept) if f is not None and "rle" in f else
                    rle.encode_views(sel["winner"], sel["kept_idx"], sel["n_kept"], sel["flags"], n_kept_host=n_kept))
        if d.get("_owner") is not None:
            d["_owner"]["fetch_pending"] = False               # every pinned buffer of the slot's recorded fetch is copied out: replay allowed
        # the per-view tensors below are VIEWS of this call's private host copies (one D2H copy per field, no per-view clone);
        # scalars come from .tolist() once (indexing a tensor per instance cost 4 ms per 32-pair step)
        kept_l, scores_l = kept_idx.tolist(), scores.tolist()
        masks_all, mask_off = None, [0]
        if self.output_masks:
            for nk in n_kept:
                mask_off.append(mask_off[-1] + nk)
            if sel["winner"].is_cuda:              # all views in one launch (was three torch launches per view)
                masks_all = ops.decode_masks(sel["winner"], sel["kept_idx"], sel["n_kept"], sel["flags"], mask_off[-1])
        results = []
        for i in range(B):
            res = {}
            for v, j in (("0", i), ("1", B + i)):
                n = n_kept[j]
                inp = batched_inputs[i][v]
                image_id, file_name = inp.get("image_id"), inp.get("file_name")
                view = {"image_id": image_id, "file_name": file_name,
                        "pred_plane": planes[j, :n], "pred_plane_feats": sel["feats"][j:j + 1, :n],
                        "pred_plane_oriIdxs": kept_l[j][:n], "pred_plane_ins_center": centers[j, :n],
                        "pred_plane_scores": scores[j, :n], "pred_plane_areas": areas[j, :n],
                        "winner_map": sel["winner"][j], "fallback_mask": bool(flags[j] & 2)}
                if self.output_masks:
                    view["pred_plane_masks"] = (masks_all[mask_off[j]:mask_off[j] + n] if masks_all is not None else
                                                decode_masks(sel["winner"][j], kept_idx[j, :n].to(sel["winner"].device), bool(flags[j] & 2)))
                inst = []
                sc_j = scores_l[j]
                rl_j = rles[j] if rles is not None else None
                for k in range(n):       # siamese_planeTR.py:705-720 (bbox_mode 1 = XYWH_ABS)
                    ins = {"image_id": image_id, "file_name": file_name, "category_id": 0, "score": sc_j[k]}
                    if rl_j is not None:
                        ins["segmentation"], ins["bbox"] = rl_j[k]["segmentation"], rl_j[k]["bbox"]
                    ins["bbox_mode"] = 1
                    inst.append(ins)
                view["instances"] = inst
                res[v] = view
            res["pred_aff"] = None
            res["depth"] = {"0": None, "1": None}
            for k, (t, r) in cams.items():
                res[k] = {"tran": t[i], "rot": r[i]}
            if m[i] >= 2:        # camera_head.py:635-639: only present when the refine stage produced all hypotheses
                res["camera_onePP"] = {"tran": onepp_t[i, :m[i] + 1], "rot": onepp_r[i, :m[i] + 1]}
            n1, n2 = n_kept[i], n_kept[B + i]
            for k, A in ass.items():
                res[k] = A[i, :n1, :n2]
            res["matched_num"] = m[i]
            results.append(res)
        return results


def decode_masks(winner: torch.Tensor, kept_idx: torch.Tensor, fallback: bool) -> torch.Tensor:
    """winner uint8 [H,W] (low 7 bits = arg-max query, bit 7 = above mask threshold) -> bool [n,H,W]
    (siamese_planeTR.py:685 / :743 in the fallback case)."""
    ids = (winner & 0x7F).to(torch.int64)
    eq = ids.unsqueeze(0) == kept_idx.view(-1, 1, 1).to(torch.int64)
    return eq if fallback else eq & ((winner & 0x80) != 0).unsqueeze(0)
