"""Multi-GPU inference plumbing: one process per GPU, pairs sharded contiguously across ranks, and ONE
collective — an all_gather of fixed-width fp32 per-pair metric rows over RCCL/xGMI — replacing the
reference's `comm.synchronize(); comm.gather(pickled predictions)` on a Gloo group
(evaluation/mp3d_evaluation.py:316-319; sharding = detectron2 InferenceSampler, test_NopeSAC.py:48-54).
The model forward itself has no collective.
"""
from __future__ import annotations

import math
import os
from typing import Callable, List, Optional

import numpy as np
import torch
import torch.distributed as dist

METRIC_WIDTH = 16   # [t(3), q(4), n1, n2, m, T_err, R_err, pair_idx, nonfinite count of the pair's batch, pad(2)]


def init_distributed(backend: Optional[str] = None):
    """Read RANK/WORLD_SIZE/LOCAL_RANK/MASTER_* (torchrun) and join the process group.  backend "nccl" is
    RCCL on ROCm; "gloo" is used by the CPU tests."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            try:      # bind the communicator to this rank's GPU (barrier() would otherwise guess the device from the rank)
                dist.init_process_group(backend=backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
            except TypeError:                                      # older torch: no device_id argument
                dist.init_process_group(backend=backend, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def _free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _launch_worker(local_rank: int, world: int, port: int, main_func, args):
    os.environ.update(RANK=str(local_rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # the host driver only supports dmabuf IPC (RCCL needs it)
    main_func(*args)


def launch(main_func, num_gpus_per_machine: int, args=()):
    """detectron2.engine.launch for ONE machine (test_NopeSAC.py:209-216): with num_gpus > 1 and no torchrun environment,
    spawn one process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set for `init_distributed`) and run
    `main_func(*args)` in each; otherwise (1 GPU, or already under torchrun) call it in this process.
    Returns main_func's value in the single-process case, None after a multi-process run (rank 0 writes the results)."""
    if num_gpus_per_machine <= 1 or "WORLD_SIZE" in os.environ:
        return main_func(*args)
    import torch.multiprocessing as mp
    mp.start_processes(_launch_worker, args=(num_gpus_per_machine, _free_port(), main_func, tuple(args)),
                       nprocs=num_gpus_per_machine, join=True, start_method="spawn")
    return None


def cpu_budget() -> int:
    """CPUs this process can actually keep busy: the cores it may run on (sched_getaffinity) capped by its cgroup's CPU quota
    (cgroup v2 `cpu.max`, v1 `cpu.cfs_quota_us / cpu.cfs_period_us`).  A container that SEES 256 hardware threads under a 16-CPU quota
    (the MI355X boxes of this build) runs 32 busy threads for half of every 100 ms period and parks ALL of them for the other half -
    measured as 40 ms stalls of every decoder thread at once; thread pools are sized from this number, not from os.cpu_count()."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = float(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = float(f.read())
            if q > 0 and per > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(math.ceil(quota))))
    return max(1, n)


def pin_rank_to_cores(local_rank: int, ranks_on_host: int) -> List[int]:
    """Restrict this process (and the threads it starts later) to the local_rank-th of `ranks_on_host` equal, contiguous shares of
    the cores it is allowed to use; returns the cores it now owns (unchanged set when the share would be empty or the platform has no
    sched_setaffinity).  MODEL.AMD.CPU_AFFINITY."""
    if not hasattr(os, "sched_setaffinity") or ranks_on_host <= 1:
        return sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    cores = sorted(os.sched_getaffinity(0))
    per = len(cores) // ranks_on_host
    if per < 1:
        return cores
    mine = cores[local_rank * per:(local_rank + 1) * per]
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return cores
    return mine


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of rank `rank` (InferenceSampler semantics: ceil(N/W) per rank)."""
    per = int(math.ceil(n_items / world))
    return min(rank * per, n_items), min((rank + 1) * per, n_items)


def rotation_error_deg(q_pred: np.ndarray, q_ref: np.ndarray) -> np.ndarray:
    """2*acos(|<q̂,q>|) in degrees (evaluation/mp3d_evaluation.py:463-465)."""
    d = np.abs(np.sum(q_pred * q_ref, axis=-1))
    return 2 * np.arccos(np.clip(d, -1.0, 1.0)) * 180.0 / np.pi


def translation_error(t_pred: np.ndarray, t_ref: np.ndarray) -> np.ndarray:
    """||t̂ - t||₂ (evaluation/mp3d_evaluation.py:389-391)."""
    return np.linalg.norm(t_pred - t_ref, axis=-1)


def metric_rows(trans: torch.Tensor, rot: torch.Tensor, n1: torch.Tensor, n2: torch.Tensor, m: torch.Tensor,
                pair_idx0: int, t_err: Optional[torch.Tensor] = None, r_err: Optional[torch.Tensor] = None,
                nonfinite: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Pack per-pair results into [B, 16] fp32 rows on the tensors' device (no host sync)."""
    B = trans.shape[0]
    if (trans.is_cuda and trans.dtype == torch.float32 and rot.dtype == torch.float32 and trans.is_contiguous() and rot.is_contiguous()
            and all(v.dtype == torch.int32 and v.is_contiguous() for v in (n1, n2, m))
            and all(v is None or (v.dtype == torch.float32 and v.is_contiguous() and v.is_cuda) for v in (t_err, r_err))
            and (nonfinite is None or nonfinite.dtype == torch.int32)):
        from . import ops
        return ops.metric_rows(trans, rot, n1, n2, m, pair_idx0, t_err, r_err, nonfinite)     # one launch instead of a dozen torch ones
    rows = torch.zeros(B, METRIC_WIDTH, device=trans.device, dtype=torch.float32)
    rows[:, 0:3], rows[:, 3:7] = trans, rot
    rows[:, 7], rows[:, 8], rows[:, 9] = n1.float(), n2.float(), m.float()
    if t_err is not None:
        rows[:, 10] = t_err
    if r_err is not None:
        rows[:, 11] = r_err
    rows[:, 12] = torch.arange(pair_idx0, pair_idx0 + B, device=trans.device, dtype=torch.float32)
    if nonfinite is not None:                 # device int32[1] from ops.count_nonfinite: travels with the rows (no host sync here)
        rows[:, 13] = nonfinite.to(torch.float32)
    return rows


def gather_metrics(rows: torch.Tensor) -> torch.Tensor:
    """all_gather of equally-sized [B,16] blocks -> [world*B, 16] on every rank (single ring step; the
    payload is KBs, i.e. latency bound)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return rows
    world = dist.get_world_size()
    out = torch.empty(world * rows.shape[0], rows.shape[1], device=rows.device, dtype=rows.dtype)
    if dist.get_backend() == "gloo":      # CPU test backend: list form of the same collective
        dist.all_gather(list(out.chunk(world, 0)), rows.contiguous())
    else:
        dist.all_gather_into_tensor(out, rows.contiguous())
    return out


def all_ranks_agree(ok: bool, device=None) -> bool:
    """True only when EVERY rank says ok (an all-reduce MIN of one flag; a single process: `ok`).  Every rank must call it at the same
    point: it is what lets a step that may fail locally (capturing a forward into a graph / launch tape) sit in front of collectives -
    the ranks first agree that all of them succeeded, and only then enter the barrier / gather that follows."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return bool(ok)
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def capture_on_all_ranks(capture_local: Callable[[], None], verify_collective: Callable[[], bool], abandon: Callable[[], None], device=None,
                         log=None) -> bool:
    """The launch-tape / graph leg under world > 1 (round 6; it used to be single-process only: a rank whose capture failed returned to
    eager launching while the others walked into the check's barrier and hung there).
      1. capture_local(): this rank's captures - NO collectives inside; an exception means "this rank cannot";
      2. all_ranks_agree: if any rank could not, every rank calls abandon() (drop its captures) and stays eager;
      3. verify_collective(): the replay check - it may use the loop's barriers and gathers, every rank runs it - returns this rank's verdict;
      4. all_ranks_agree again: one rank whose replay does not reproduce its eager rows sends every rank back to eager launching.
    Returns True when every rank replays."""
    err = None
    try:
        forced = os.environ.get("NOPESAC_FAIL_CAPTURE_RANK")           # tests / node bring-up: make this rank's capture fail
        if forced is not None and int(forced) == (dist.get_rank() if dist.is_available() and dist.is_initialized() else 0):
            raise RuntimeError("capture failure forced by NOPESAC_FAIL_CAPTURE_RANK")
        capture_local()
    except Exception as e:  # noqa: BLE001 - whatever the capture raised: the rank stays eager and says so
        err = e
    if not all_ranks_agree(err is None, device):
        if log is not None:
            log("capture abandoned on every rank (%s)" % ("this rank: %r" % (err,) if err is not None else "another rank could not capture"))
        abandon()
        return False
    try:
        ok = bool(verify_collective())
    except Exception as e:  # noqa: BLE001
        err, ok = e, False
    if not all_ranks_agree(ok, device):
        if log is not None:
            log("replay check failed (%s): every rank stays eager" % ("this rank: %r" % (err,) if err is not None else "on this or another rank"))
        abandon()
        return False
    return True


def summarize(rows: torch.Tensor) -> dict:
    r = rows.detach().cpu().numpy()
    return {"pairs": int(r.shape[0]), "mean_T_err": float(r[:, 10].mean()), "mean_R_err": float(r[:, 11].mean()),
            "mean_planes": float((r[:, 7] + r[:, 8]).mean() / 2), "mean_matches": float(r[:, 9].mean())}


class InflightLoop:
    """`n_slots` batches in flight: step i runs on HIP stream i % n_slots (its own stream, its own pinned host buffer for the
    gathered rows), so the latency-bound head stages of one batch overlap with the HBM / MFMA-bound backbone of the next.  Per
    step and rank there is exactly ONE collective (`gather_metrics`, an all_gather of [B,16] rows), issued in step order on every
    rank - collectives of different slots go through the same communicator in the same order everywhere.  A slot's results are
    complete when its event has fired (checked before the slot is reused and by `drain`).  On a CPU (the gloo tests) the same
    control flow runs without streams / events."""

    def __init__(self, n_slots: int, rows_per_rank: int, device=None, world: int = 1, side_shift: Optional[int] = 0, pace_s: float = 0.0,
                 gather_every: int = 1):
        """side_shift: see streams.StreamSet (which hardware queue a batch's pose-net side stream shares); None = plain
        torch.cuda.Stream()s, whatever queues the runtime hands out.
        gather_every = G > 1 (round 5): the rows of G consecutive steps are parked in a device buffer and travel in ONE all_gather of
        [G * B, 16] rows, issued by the step that completes the group (the reference gathers once, at the end).  With one
        collective per step every rank waits for the slowest rank every ~8 ms and an RCCL kernel competes with the persistent conv
        workgroups every step; with G = 8 the ranks are coupled every ~70 ms.  `flush()` (called by `barrier`) gathers a partial group -
        every rank has then run the same number of steps, so the collective sequence is identical everywhere."""
        self.n_slots = max(1, int(n_slots))
        self.G = max(1, int(gather_every))
        self.world, self.rows_per_rank = int(world), int(rows_per_rank)
        self.cuda = device is not None and torch.device(device).type == "cuda"
        self.stream_set = None
        if self.cuda and side_shift is not None:
            from .streams import stream_set
            self.stream_set = stream_set(self.n_slots, device, side_shift)
            self.streams = self.stream_set.mains
        else:
            self.streams = [torch.cuda.Stream(device=device) for _ in range(self.n_slots)] if self.cuda else [None] * self.n_slots
        self.host_bufs = [torch.empty(world * rows_per_rank, METRIC_WIDTH, dtype=torch.float32) for _ in range(self.n_slots)]
        if self.cuda:
            self.host_bufs = [b.pin_memory() for b in self.host_bufs]
        self.done = [None] * self.n_slots
        if self.G > 1:                                         # two groups: one filling, one whose gather may still be in flight
            dev = device if self.cuda else None
            self.acc = [torch.zeros(self.G, rows_per_rank, METRIC_WIDTH, dtype=torch.float32, device=dev) for _ in range(2)]
            self.acc_ev = [[None] * self.G for _ in range(2)]
            self.group_host = [torch.empty(world * self.G * rows_per_rank, METRIC_WIDTH, dtype=torch.float32) for _ in range(2)]
            if self.cuda:
                self.group_host = [b.pin_memory() for b in self.group_host]
            self.group_done = [None, None]
            self.group_idx, self.group_fill, self.latest = 0, 0, None
            self.collectives = 0
        self.host_seconds = 0.0
        self.last = None
        self.pace_s = float(pace_s)                            # minimum time between two submissions (see step)
        self._last_submit = 0.0

    def step(self, i: int, device_step):
        """device_step(slot) -> (anything, rows [B,16] on the device); returns (anything, this slot's host buffer)."""
        import contextlib
        import time
        slot = i % self.n_slots
        if self.done[slot] is not None:
            self.done[slot].synchronize()                      # the slot's previous results have reached the host
        if self.pace_s > 0.0:
            # keep submissions apart.  Eager submission takes ~4 ms per batch, which staggers the batches in flight by itself; a tape
            # replay submits a batch in < 1 ms, slots that finish together are then re-submitted together and run in lockstep
            # (all in the MFMA / HBM-bound backbone, then all in the latency-bound heads): measured 3488 pairs/s unpaced, 3608 with
            # 4 ms between submissions (eager: 3635)
            rest = self.pace_s - (time.perf_counter() - self._last_submit)
            if rest > 3e-4:
                time.sleep(rest - 2e-4)
            while time.perf_counter() - self._last_submit < self.pace_s:
                pass
        t0 = time.perf_counter()
        self._last_submit = t0
        ctx = torch.cuda.stream(self.streams[slot]) if self.cuda else contextlib.nullcontext()
        host = None
        with torch.no_grad(), ctx:
            d, rows = device_step(slot)
            if self.G == 1:
                allrows = gather_metrics(rows)                 # the only collective (RCCL all_gather, KBs)
                self.host_bufs[slot].copy_(allrows, non_blocking=True)   # results leave the device once per step
                host = self.host_bufs[slot]
            else:
                g, pos = self.group_idx & 1, self.group_fill
                if pos == 0 and self.group_done[g] is not None:
                    self.group_done[g].synchronize()           # this buffer's previous gather (two groups ago) has reached the host
                self.acc[g][pos].copy_(rows, non_blocking=True)
                if self.cuda:
                    self.acc_ev[g][pos] = torch.cuda.Event()
                    self.acc_ev[g][pos].record()
                self.group_fill += 1
                if self.group_fill == self.G:
                    self._gather_group()
            if self.cuda:
                ev = torch.cuda.Event()
                ev.record()
                self.done[slot] = ev
        self.host_seconds += time.perf_counter() - t0
        self.last = (d, slot)
        return d, host

    def _gather_group(self):
        """ONE all_gather of the group's rows on the CURRENT stream (the stream of the step that completes the group), behind the
        row writes of the group's other steps (their streams' events)."""
        g, n = self.group_idx & 1, self.group_fill
        if self.cuda:
            cur = torch.cuda.current_stream()
            for ev in self.acc_ev[g][:n]:
                cur.wait_event(ev)
        allrows = gather_metrics(self.acc[g][:n].reshape(n * self.rows_per_rank, METRIC_WIDTH))     # [world * n * B, 16], rank-major
        self.collectives += 1
        self.group_host[g][:allrows.shape[0]].copy_(allrows, non_blocking=True)
        if self.cuda:
            self.group_done[g] = torch.cuda.Event()
            self.group_done[g].record()
        self.latest = (g, n)
        self.group_idx += 1
        self.group_fill = 0

    def flush(self):
        """gather_every > 1: gather the steps since the last collective (a partial group).  COLLECTIVE: every rank calls it after the
        same number of steps (barrier() does)."""
        if self.G > 1 and self.group_fill > 0:
            import contextlib
            slot = self.last[1] if self.last is not None else 0
            ctx = torch.cuda.stream(self.streams[slot]) if self.cuda else contextlib.nullcontext()
            with torch.no_grad(), ctx:
                self._gather_group()

    def last_group_rows(self) -> Optional[torch.Tensor]:
        """gather_every > 1: the most recent gathered group as [world, n steps, B, 16] (host; valid once its event has fired - after
        drain() / barrier())."""
        if self.G == 1 or self.latest is None:
            return None
        g, n = self.latest
        return self.group_host[g][:self.world * n * self.rows_per_rank].view(self.world, n, self.rows_per_rank, METRIC_WIDTH)

    def last_step_rows(self) -> Optional[torch.Tensor]:
        """The gathered rows [world * B, 16] of the last step that has been gathered (rank-major), as step() returns them with
        gather_every = 1."""
        if self.G == 1:
            return self.host_bufs[self.last[1]] if self.last is not None else None
        grp = self.last_group_rows()
        return None if grp is None else grp[:, -1].reshape(self.world * self.rows_per_rank, METRIC_WIDTH)

    def drain(self):
        for ev in self.done:
            if ev is not None:
                ev.synchronize()
        if self.G > 1:
            for ev in self.group_done:
                if ev is not None:
                    ev.synchronize()

    def barrier(self):
        self.flush()
        self.drain()
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.barrier()
        if self.cuda:
            torch.cuda.synchronize()

