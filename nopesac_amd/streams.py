"""HIP streams with KNOWN hardware queues.

A process's HIP streams are multiplexed onto a few hardware queues (4 by default, `GPU_MAX_HW_QUEUES`): the runtime hands a new
stream the least-referenced queue, the null stream holds one from the start, ties are broken its own way - and launches of two
streams that share a queue execute strictly in submission order (each waits for the one in front of it).  With four batches in
flight on (batch stream, pose-net side stream) pairs that is decisive: when two BATCH streams land on one queue - which is what
happens to streams 3 and 4 of a fresh process - two of the four batches serialise and the drop-in boundary loses 20 % (measured:
2560 vs 3150 pairs/s, profiles/r3_m_hw_queues.txt); which side stream shares whose queue moves the resident-input loop by 4 %.
The runtime offers no way to ask for a queue, but sharing can be OBSERVED: a tiny kernel enqueued behind a long spin kernel of
another stream finishes early iff the two streams are on different queues.  `StreamSet` creates candidate streams, sorts them into
queue classes with that probe and picks the batch / side streams by class:

    batch stream i      : its own queue class, for every i (so never behind another batch's launches)
    side stream of i    : a queue no batch stream uses while there is one (fewer slots than queues: the pose net then overlaps with
                          its own batch), else a stream in the class of batch stream (i + side_shift) % n.  side_shift = 0 (behind
                          its own batch: the four queues stay independent) is the default; measured at the drop-in boundary with 4
                          slots: shift 0 / 1 / 2 / 3 = 3240 / 2340 / 2960 / 2990 pairs/s, queues as the runtime hands them out 2530
                          (profiles/r3_o_stream_policy.txt); resident-input loop 3640 / - / 3700 / 3450, runtime's choice 3590

Unused candidates stay alive (destroying one would change the reference counts the runtime balances by)."""
from __future__ import annotations

from typing import List, Optional

import torch

_SPIN = 2_000_000          # shader cycles of the blocking probe (~0.8 ms)


def shares_queue(a: "torch.cuda.Stream", b: "torch.cuda.Stream", scratch: torch.Tensor = None) -> bool:
    """True iff work enqueued on `b` waits for earlier work on `a` (= one hardware queue).  Both streams must be idle.
    `scratch`: int64[4] on the device (allocated HERE it would come from the caching allocator's pool of the current stream - and a
    stream's first allocation is a hipMalloc, which waits for the whole device: every pair would look shared)."""
    from . import _lib
    L = _lib.load()
    if scratch is None:
        scratch = torch.zeros(4, device=a.device, dtype=torch.int64)
        torch.cuda.synchronize(a.device)
    s0, s1, t1 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    s0.record(a)
    _lib.check(L.nopesac_clock_probe(scratch.data_ptr(), _SPIN, a.cuda_stream), "nopesac_clock_probe")
    s1.record(a)
    _lib.check(L.nopesac_clock_probe(scratch.data_ptr() + 16, 2000, b.cuda_stream), "nopesac_clock_probe")
    t1.record(b)
    s1.synchronize()
    t1.synchronize()
    return s0.elapsed_time(t1) > 0.5 * s0.elapsed_time(s1)


def expected_hw_queues() -> int:
    """Hardware queues the runtime multiplexes a process's streams onto (GPU_MAX_HW_QUEUES, default 4)."""
    import os
    try:
        return max(1, int(os.environ.get("GPU_MAX_HW_QUEUES", "4")))
    except ValueError:
        return 4


class StreamSet:
    def __init__(self, n_slots: int, device=None, side_shift: int = 0, candidates: int = 16, with_sides: bool = True):
        import os
        self.device = torch.device(device if device is not None else "cuda")
        self.n = int(n_slots)
        self.probe_passes = 0
        with torch.cuda.device(self.device):
            scratch = torch.zeros(4, device=self.device, dtype=torch.int64)
            torch.cuda.synchronize()
            self.pool: List[torch.cuda.Stream] = [torch.cuda.Stream(device=self.device) for _ in range(max(candidates, 2 * self.n))]
            from . import _lib
            for s in self.pool:            # first use: the runtime creates the stream's queue object now (milliseconds) - not inside a probe
                _lib.check(_lib.load().nopesac_clock_probe(scratch.data_ptr(), 1000, s.cuda_stream), "nopesac_clock_probe")
            torch.cuda.synchronize()

            def classify():
                classes: List[List[torch.cuda.Stream]] = []           # streams grouped by hardware queue
                for s in self.pool:
                    for c in classes:
                        if shares_queue(c[0], s, scratch):
                            c.append(s)
                            break
                    else:
                        classes.append([s])
                torch.cuda.synchronize()
                return classes

            # The probe is a TIMING observation: on a GPU that other processes keep busy (eight ranks are one process per GPU, but a
            # shared development box, a profiler or a co-tenant is not) the tiny kernel can be delayed like a shared queue would delay
            # it, and distinct queues merge into one class.  A process with N streams sees min(N, GPU_MAX_HW_QUEUES) queues: fewer
            # classes than that means the observation was disturbed - look once more, then give up and use the streams as the runtime
            # placed them (round 5; NOPESAC_STREAM_PROBE=0 skips the probe altogether).
            want = min(len(self.pool), expected_hw_queues())
            self.classes = [list(self.pool)]
            if os.environ.get("NOPESAC_STREAM_PROBE", "1") != "0":
                for _ in range(2):
                    self.classes = classify()
                    self.probe_passes += 1
                    if len(self.classes) >= min(self.n, want):
                        break
        self.queue_classes = len(self.classes)
        self.inconclusive = self.n > 1 and len(self.classes) < min(self.n, want)
        if self.inconclusive:
            # fewer queue classes than batch streams to place although the runtime has that many queues: either the process really has
            # fewer hardware queues than GPU_MAX_HW_QUEUES says, or the probe was disturbed.  Forcing several batch streams into one
            # observed class would serialise them for certain if the observation is right and gain nothing if it is wrong - hand out the
            # streams as the runtime placed them instead (its own balancing: least-referenced queue first).
            self.mains = self.pool[:self.n]
            self.sides = list(self.pool[self.n:2 * self.n]) if with_sides else [None] * self.n
            return
        # batch streams: one per class while classes last (then round robin: more slots than queues must share)
        self.mains: List[torch.cuda.Stream] = []
        order = sorted(self.classes, key=len, reverse=True)
        used = {id(c): 0 for c in order}
        for i in range(self.n):
            c = order[i % len(order)]
            if used[id(c)] >= len(c):                                   # uneven classes (a probe disturbed under load can leave a singleton)
                rest = [d for d in order if used[id(d)] < len(d)]
                if not rest:                                            # every candidate handed out: a new stream, wherever the runtime puts it
                    self.mains.append(torch.cuda.Stream(device=self.device))
                    continue
                c = rest[0]
            self.mains.append(c[used[id(c)]])
            used[id(c)] += 1
        self.sides: List[Optional[torch.cuda.Stream]] = [None] * self.n
        if with_sides:
            free = [c for c in order[self.n:]]                          # queues no batch stream sits on (fewer slots than queues)
            for i in range(self.n):
                if free:                                                # an otherwise idle queue: the pose net overlaps with its own batch
                    c = free.pop(0)
                else:
                    c = order[(i + side_shift) % self.n % len(order)]
                k = used[id(c)]
                if k < len(c):
                    self.sides[i] = c[k]
                    used[id(c)] += 1
                else:                                                   # class exhausted: a new stream, wherever the runtime puts it
                    self.sides[i] = torch.cuda.Stream(device=self.device)

    def bind(self, model) -> "StreamSet":
        """Make `model` (PlaneTR_NopeSAC) run the pose net of a batch submitted on mains[i] on sides[i]."""
        if model._side_stream is None:
            model._side_stream = {}
        for m, s in zip(self.mains, self.sides):
            if s is not None:
                model._side_stream[m.cuda_stream] = s
        return self

    def describe(self) -> dict:
        idx = {id(s): k for k, c in enumerate(self.classes) for s in c}
        return {"queue_classes": self.queue_classes, "probe_inconclusive": self.inconclusive, "probe_passes": self.probe_passes,
                "class_sizes": [len(c) for c in self.classes],
                "batch_stream_class": [idx.get(id(s)) for s in self.mains],
                "side_stream_class": [idx.get(id(s)) if s is not None else None for s in self.sides]}


_CACHE = {}


def stream_set(n_slots: int, device=None, side_shift: int = 0) -> StreamSet:
    """The process-wide StreamSet for (device, n_slots, side_shift): probing takes ~50 ms and every set keeps 16 streams alive, so
    loops that come and go (one per `inference_on_dataset` call) share one."""
    dev = torch.device(device if device is not None else "cuda")
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    key = (dev.index, int(n_slots), int(side_shift))
    if key not in _CACHE:
        _CACHE[key] = StreamSet(n_slots, dev, side_shift=side_shift)
    return _CACHE[key]
