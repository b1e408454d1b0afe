"""Inference CLI: the counterpart of the reference's `test_NopeSAC.py` + detectron2 `inference_on_dataset`.

    python -m nopesac_amd.run --config-file configs/inference_mp3d.yaml --eval-only [--num-gpus N] KEY VALUE ...
    python -m torch.distributed.run --nproc-per-node N -m nopesac_amd.run --config-file ... --num-gpus N ...

Reproduces: cfg = defaults -> NopeSAC defaults -> merge_from_file -> merge_from_list -> freeze
(test_NopeSAC.py:182-192); build_model + checkpoint load by state-dict key names (:198-201; `{"model": sd}` or a
bare state dict); per-rank contiguous sharding of the pair list (InferenceSampler, :48-54); the batch loop
`outputs = model(inputs); evaluator.process(inputs, outputs)` with a s/pair log (:157-179); evaluator.evaluate()
with the single RCCL gather.  Dataset I/O is out of scope: pairs come from `--pairs-file`
(a torch-saved list of reference-format input dicts) or are synthetic (`--synthetic-pairs N`).
"""
from __future__ import annotations

import argparse
import json
import logging
import os
import sys
import time

import torch

from . import runner
from .config import get_cfg
from .evaluation import PoseEvaluator, create_small_table, dump_predictions, evaluate_for_matchings
from .registry import build_model
from .synth import synth_pair, synth_state_dict

logger = logging.getLogger("nopesac_amd")


def default_argument_parser():
    ap = argparse.ArgumentParser(description="NopeSAC inference on MI355X (nopesac_amd)")
    ap.add_argument("--config-file", default="", metavar="FILE")
    ap.add_argument("--eval-only", action="store_true")
    ap.add_argument("--num-gpus", type=int, default=1)
    ap.add_argument("--pairs-per-batch", type=int, default=32, help="pairs per model call (the reference: 1; throughput saturates around 32)")
    ap.add_argument("--inflight", type=int, default=4, help="batches in flight, each on a HIP stream with a hardware queue of its own (1 = strictly serial, like the reference loop; 4 = the number of hardware queues: 1590 / 2290 / 3250 pairs/s at 1 / 2 / 4)")
    ap.add_argument("--pairs-file", default="", help="torch-saved list of input dicts (reference mapper format)")
    ap.add_argument("--dataset", default="", help="registered split name (mp3d_test, scannet_test, ...): read <datasets-dir>/<split json> "
                    "through the PairMapper (nopesac_amd/data.py); default = cfg.DATASETS.TEST[0] when its json exists")
    ap.add_argument("--datasets-dir", default="./datasets")
    ap.add_argument("--limit", type=int, default=0, help="use only the first N pairs of the dataset")
    ap.add_argument("--decode-workers", type=int, default=0, help="decoder threads of a dataset split (0 = max(DATALOADER.NUM_WORKERS, min(32, cores)): "
                    "one PIL decode is ~5 ms, the GPU consumes a pair in under 0.5 ms)")
    ap.add_argument("--uint8-images", action="store_true", help="hand the decoded 8-bit images to the model as uint8 tensors (widened on the device: "
                    "identical results, a quarter of the host-to-device bytes) instead of the reference mapper's float32 tensors")
    ap.add_argument("--synthetic-pairs", type=int, default=0)
    ap.add_argument("--structured", action="store_true", help="structured synthetic images instead of noise")
    ap.add_argument("--synthetic-weights", action="store_true", help="name-seeded checkpoint instead of cfg.MODEL.WEIGHTS")
    ap.add_argument("--output", default="", help="write the result summary JSON here")
    ap.add_argument("--eval-matchings", action="store_true", help="plane-matching precision / recall / F-score (mp3d_evaluation.py:746-849): "
                    "needs pairs with `gt_corrs` and RLE `annotations` (the dataset json's own fields)")
    ap.add_argument("--dump-dir", default="", help="write NopeSAC_instances_predictions.pth + continuous.pkl here (eval_full_scene)")
    ap.add_argument("--stub-model", action="store_true", help="TEST ONLY (results are marked invalid): the CLI's sharding, batch loop, evaluator "
                    "gather and dumps around a stub that fabricates result dicts instead of running the model - what the gloo CPU tests drive at "
                    "world sizes no box offers")
    ap.add_argument("opts", nargs=argparse.REMAINDER, default=[], help="KEY VALUE config overrides")
    return ap


def setup(args):
    cfg = get_cfg()
    if args.config_file:
        cfg.merge_from_file(args.config_file)
    cfg.merge_from_list(args.opts)
    cfg.freeze()
    return cfg


def load_checkpoint(model, cfg, synthetic: bool):
    if synthetic:
        model.load_state_dict(synth_state_dict(cfg.MODEL.SEM_SEG_HEAD.NUM_OBJECT_QUERIES))
        return "synthetic(name-seeded)"
    path = cfg.MODEL.WEIGHTS
    if not path or not os.path.exists(path):
        raise FileNotFoundError(f"MODEL.WEIGHTS={path!r} not found (pass --synthetic-weights for the name-seeded checkpoint)")
    ckpt = torch.load(path, map_location="cpu")
    model.load_state_dict(ckpt)
    return path


def load_pairs(args, cfg=None):
    if args.pairs_file:
        return torch.load(args.pairs_file)
    name = args.dataset or (cfg.DATASETS.TEST[0] if cfg is not None and len(cfg.DATASETS.TEST) and not args.synthetic_pairs else "")
    if name:
        from . import data
        if args.dataset or os.path.exists(data.dataset_json(name, args.datasets_dir)):
            lazy = data.build_inference_pairs(cfg, name, args.datasets_dir, args.limit, uint8=args.uint8_images, lazy=True,
                                              prefetch=max(64, 3 * args.pairs_per_batch))
            lazy.workers = args.decode_workers or max(lazy.workers, min(32, runner.cpu_budget()))
            return lazy
    n = args.synthetic_pairs or 8
    pairs = []
    for i in range(n):
        if args.stub_model:      # no pixels needed: tiny images, a ground-truth pose so that the evaluator's error tables are exercised
            p = synth_pair(i, 8, 8)
            p["rel_pose"] = {"position": [float(i), 0.5, -0.25], "rotation": [1.0, 0.0, 0.0, 0.0]}
        else:
            p = synth_pair(i, structured=args.structured)
        pairs.append(p)
    return pairs


def inference_on_dataset(model, pairs, evaluator, pairs_per_batch: int, keep_outputs: list = None, inflight: int = 1):
    """Batch loop of detectron2's inference_on_dataset (eval mode, no_grad, timing log).
    inflight > 1 (GPU only): that many batches are in flight on their own HIP streams - batch i's host-to-device copies and forward
    are enqueued before the results of batch i - inflight + 1 are fetched, packaged and handed to the evaluator (in order); the
    head stages of one batch then run under the convolutions of the next (2-3x the strictly serial rate)."""
    evaluator.reset()
    model.eval()
    t0, n_done, t_compute = time.perf_counter(), 0, 0.0
    on_gpu = torch.cuda.is_available() and next(model.parameters()).is_cuda
    depth = max(1, inflight) if on_gpu else 1
    streams = []
    if depth > 1:
        # one hardware queue per batch stream, the pose-net side stream on its batch's queue (streams.py: two batch streams on one
        # queue - what a fresh process gets for streams 3 and 4 - serialise those batches: 2560 instead of 3240 pairs/s)
        from .streams import stream_set
        streams = stream_set(depth, next(model.parameters()).device, 0).bind(model).mains
    if depth > 1 and getattr(model, "use_hip_graph", False):
        model.graph_slots = max(model.graph_slots, depth)      # a graph slot's outputs must outlive the batches submitted after it
    pending = []
    pace, period, last_submit = bool(getattr(model, "use_hip_graph", False)), 0.0, 0.0

    def consume(batch, outputs):
        nonlocal n_done
        evaluator.process(batch, outputs)
        if keep_outputs is not None:         # what evaluate_for_matchings reads: ids, RLE instances, assignment matrices
            for out in outputs:
                keep_outputs.append({**{v: {"image_id": out[v]["image_id"], "instances": out[v]["instances"]} for v in "01"},
                                     **{k: out[k].detach().cpu().clone() for k in out if "assignment" in k}})   # (a view of the batch's pinned buffer)
        n_done += len(batch)

    def finish(item):
        batch, dev_out, ev, st = item
        ev.synchronize()
        with torch.cuda.stream(st):
            return batch, model.package(batch, dev_out)

    # a lazily decoded split (data.LazyPairs) hands out its batches itself, decoding ahead with its thread pool
    batches = pairs.iter_batches(pairs_per_batch) if hasattr(pairs, "iter_batches") else (pairs[i:i + pairs_per_batch]
                                                                                         for i in range(0, len(pairs), pairs_per_batch))
    with torch.no_grad():
        for bi, batch in enumerate(batches):
            t1 = time.perf_counter()
            if depth > 1:
                st = streams[bi % depth]
                if pace and period > 0.0:
                    # graph / tape mode submits a batch in ~1 ms: keep submissions 0.45 of a batch period apart, as eager launching
                    # does by itself, or batches that finish together restart together and run in lockstep (runner.InflightLoop.step)
                    rest = 0.45 * period - (time.perf_counter() - last_submit)
                    if rest > 0.0:
                        time.sleep(rest)
                now = time.perf_counter()
                if bi > 0:
                    period = (now - last_submit) if period == 0.0 else 0.8 * period + 0.2 * (now - last_submit)
                last_submit = now
                with torch.cuda.stream(st):
                    model.infer_iter += 1
                    dev_out = model.forward_device(batch)
                    ev = torch.cuda.Event()
                    ev.record()
                pending.append((batch, dev_out, ev, st))
                if bi == 0:
                    # the model builds its shared, lazily packed tensors (packed / fragment-major weights, fp8 copies, position
                    # encodings) with kernels on THIS batch's stream: the other streams must not read them before they are
                    # complete, so the first batch runs to the end before a second stream is used
                    torch.cuda.synchronize()
                done = [finish(pending.pop(0))] if len(pending) >= depth else []
            else:
                outputs = model(batch)
                if on_gpu:
                    torch.cuda.synchronize()
                done = [(batch, outputs)]
            t_compute += time.perf_counter() - t1
            for b, o in done:
                consume(b, o)
            if bi % 10 == 0:
                logger.info("Inference done %d/%d pairs. %.4f s / pair", n_done, len(pairs), t_compute / max(n_done, 1))
        while pending:
            t1 = time.perf_counter()
            b, o = finish(pending.pop(0))
            t_compute += time.perf_counter() - t1
            consume(b, o)
    total = time.perf_counter() - t0
    return {"pairs": n_done, "total_s": total, "compute_s": t_compute, "s_per_pair": t_compute / max(n_done, 1), "batches_in_flight": depth}


class _StubModel(torch.nn.Module):
    """--stub-model: result dicts of the reference's shape, fabricated from the inputs (pose = ground truth + an offset that depends on
    the pair's image ids, so that every rank's rows are distinguishable after the gather).  No kernel runs."""

    def __init__(self):
        super().__init__()
        self.anchor = torch.nn.Parameter(torch.zeros(1), requires_grad=False)
        self.infer_iter = 0

    def forward(self, batch):
        import numpy as np
        import zlib
        out = []
        for inp in batch:
            gt = inp.get("rel_pose") or {"position": [0.0, 0.0, 0.0], "rotation": [1.0, 0.0, 0.0, 0.0]}
            h = zlib.crc32((str(inp["0"].get("image_id")) + "|" + str(inp["1"].get("image_id"))).encode()) % 1000
            t = np.asarray(gt["position"], np.float32) + np.float32(1e-3 * h)
            q = np.asarray(gt["rotation"], np.float32)
            cam = {"tran": t, "rot": q}
            res = {v: {"image_id": inp[v].get("image_id"), "file_name": inp[v].get("file_name"), "instances": [],
                       "pred_plane": torch.zeros(1 + h % 3, 3)} for v in "01"}
            res.update({k: dict(cam) for k in ("camera", "camera_init", "camera_initRec", "camera_avgRef0", "camera_softRef0")})
            res.update(pred_aff=None, depth={"0": None, "1": None}, matched_num=float(h % 7),
                       pred_assignment=torch.zeros(1, 1 + h % 3, 1 + h % 3))
            out.append(res)
        return out


def tune_kernels(model, cfg, pairs_per_batch: int, rank: int = 0) -> int:
    """MODEL.AMD.AUTOTUNE / ROUTING_FILE: load-time kernel selection for the shapes of a `pairs_per_batch` forward (bfloat16 mode on a
    GPU only; the fp32 parity path keeps the built-in heuristic).  Returns the number of shapes measured now."""
    from . import ops
    from .config import amd_options
    amd = amd_options(cfg)
    if not (next(model.parameters()).is_cuda and model.compute_dtype == torch.bfloat16):
        return 0
    path = str(amd.ROUTING_FILE or "")
    if path and os.path.exists(path):
        logger.info("kernel routing: %d entries from %s", ops.TUNER.load(path), path)
    if not amd.AUTOTUNE:
        return 0
    t0 = time.perf_counter()
    n_before = len(ops.TUNER.log)
    model.autotune(pairs_per_batch)
    measured = len(ops.TUNER.log) - n_before
    logger.info("kernel autotuning: %d conv / GEMM shapes measured in %.1f s (%d decisions in force)", measured, time.perf_counter() - t0,
                len(ops.TUNER.best))
    if path and rank == 0 and measured:
        merged = dict(ops.TUNER.loaded)
        merged.update({ops.TUNER.key_str(k): int(v) for k, v in ops.TUNER.best.items()})
        with open(path, "w") as f:
            json.dump({"format": "nopesac_amd.ConvTuner/1", "meta": {"written_by": "nopesac_amd.run"},
                       "kernels": {str(k): v for k, v in ops.CONV_CFG_KERNEL.items()}, "routing": dict(sorted(merged.items()))}, f, indent=1)
    return measured


def main(argv=None):
    """Entry point (test_NopeSAC.py:207-216): `--num-gpus N` outside a torchrun environment starts N ranks itself."""
    args = default_argument_parser().parse_args(argv)
    if args.num_gpus > 1 and "WORLD_SIZE" not in os.environ:
        return runner.launch(_main_rank, args.num_gpus, (args,))
    return _main_rank(args)


def _main_rank(args):
    logging.basicConfig(level=logging.INFO, format="[%(asctime)s %(name)s]: %(message)s")
    rank, world, local = runner.init_distributed()
    cfg = setup(args)
    if cfg.MODEL.DEVICE == "cuda" and torch.cuda.is_available():
        torch.cuda.set_device(local)
    # one rank's share of the host: cores (MODEL.AMD.CPU_AFFINITY) and, from them, the decode-thread budget - eight ranks that each
    # start min(32, cores) decoder threads would oversubscribe the host eightfold
    from .config import amd_options
    ranks_here = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    my_cores = runner.pin_rank_to_cores(local, ranks_here) if (world > 1 and amd_options(cfg).CPU_AFFINITY) else None
    n_cores = min(len(my_cores), max(1, runner.cpu_budget() // max(1, ranks_here))) if my_cores is not None else max(1, runner.cpu_budget() // max(1, ranks_here))
    if not args.decode_workers:
        args.decode_workers = max(1, min(32, n_cores - (1 if n_cores > 2 else 0)))      # (one core stays with the thread that launches kernels)
    # torch's own CPU thread pool (dtype conversions of host batches, the evaluator's small ops) defaults to every hardware thread it sees
    torch.set_num_threads(max(1, min(torch.get_num_threads(), n_cores)))
    if args.stub_model:
        model, src = _StubModel(), "stub (no model: fabricated results)"
    else:
        model = build_model(cfg)
        src = load_checkpoint(model, cfg, args.synthetic_weights)
        tune_kernels(model, cfg, args.pairs_per_batch, rank)
    import gc
    gc.collect()
    gc.freeze()          # model, packed weights and caches leave the cyclic GC's generations: a gen-2 pass over them cost 30-50 ms every few batches
    pairs = load_pairs(args, cfg)
    lo, hi = runner.shard_range(len(pairs), rank, world)
    logger.info("rank %d/%d: weights=%s pairs [%d,%d) of %d", rank, world, src, lo, hi, len(pairs))
    evaluator = PoseEvaluator(keep_predictions=bool(args.dump_dir))
    kept = [] if args.eval_matchings else None
    try:
        timing = inference_on_dataset(model, pairs[lo:hi], evaluator, args.pairs_per_batch, kept, inflight=args.inflight)
    finally:
        gc.unfreeze()
    results = evaluator.evaluate()
    if args.eval_matchings:
        if world > 1:                            # comm.gather semantics: rank-ordered concatenation
            parts = [None] * world
            torch.distributed.all_gather_object(parts, kept)
            kept = [p for part in parts for p in part]
        annotated = pairs.entries if hasattr(pairs, "entries") else pairs           # (a LazyPairs keeps the json entries: no pixels needed here)
        dataset_dict = {p["0"]["image_id"] + "__" + p["1"]["image_id"]: p for p in annotated if "gt_corrs" in p}
        if rank == 0 and dataset_dict:
            results["matching"] = evaluate_for_matchings([k for k in kept if k["0"]["image_id"] + "__" + k["1"]["image_id"] in dataset_dict],
                                                         dataset_dict)
            for k, v in results["matching"].items():
                logger.info("Plane metrics (%s):\n%s", k, create_small_table({kk: float(vv) for kk, vv in v.items()}))
        elif rank == 0:
            logger.warning("--eval-matchings: no pair carries gt_corrs / annotations; nothing to evaluate")
    timing.update(decode_threads=args.decode_workers, cores_of_this_rank=n_cores, pinned=my_cores is not None)
    results["timing(rank0)"] = timing
    if args.stub_model:
        results["INVALID_stub_model"] = True
    if args.dump_dir:            # eval_full_scene dumps (mp3d_evaluation.py:330-341)
        files = dump_predictions(evaluator._predictions, args.dump_dir)
        if rank == 0:
            logger.info("wrote %s", files)
    if rank == 0:
        for k, v in results.items():
            if isinstance(v, dict) and v and all(isinstance(x, (int, float)) for x in v.values()):
                logger.info("%s metrics (final output mode -> %s):\n%s", k, cfg.MODEL.CAMERA_HEAD.INFERENCE_OUT_CAM_TYPE,
                            create_small_table({kk: float(vv) for kk, vv in v.items()}))
        if args.output:
            os.makedirs(os.path.dirname(os.path.abspath(args.output)), exist_ok=True)
            with open(args.output, "w") as f:
                json.dump(results, f, indent=1)
    if world > 1:
        torch.distributed.destroy_process_group()
    return results


if __name__ == "__main__":
    main()
