"""Launch tape: a captured forward replayed as plain kernel launches by a C loop (csrc/tape.hip, include/nopesac_hip.h).

    g = torch.cuda.CUDAGraph(keep_graph=True)       # the graph (and its private memory pool) must outlive the tape
    with torch.cuda.graph(g, stream=s): out = forward(static_inputs)
    tape = LaunchTape(g)                            # reads the graph's nodes back; raises TapeUnsupported if it cannot
    tape.replay()                                   # ~270 hipLaunchKernel calls on torch's current stream, no Python in between

Why not g.replay(): a whole-graph launch cuts the submit time just as well, but with several batches in flight the graphs of
different slots overlap less than eagerly launched streams (measured 2199 vs 2790 pairs/s at the drop-in boundary, round 2)."""
from __future__ import annotations

import ctypes

import torch

from . import _lib


class TapeUnsupported(RuntimeError):
    """The captured graph cannot be turned into a tape on this runtime (node kinds / parameters not readable)."""


class LaunchTape:
    def __init__(self, graph: "torch.cuda.CUDAGraph", max_streams: int = 4):
        if not hasattr(graph, "raw_cuda_graph"):
            raise TapeUnsupported("this torch build does not expose CUDAGraph.raw_cuda_graph()")
        try:
            raw = graph.raw_cuda_graph()
        except Exception as e:                                   # not created with keep_graph=True
            raise TapeUnsupported("raw_cuda_graph(): %r" % (e,))
        lib = _lib.load()
        handle = ctypes.c_void_p()
        counts = (ctypes.c_int32 * 4)()
        rc = lib.nopesac_tape_create_ex(ctypes.c_void_p(int(raw)), int(max_streams), ctypes.byref(handle), counts)
        if rc != 0:
            msg = lib.nopesac_last_error()
            raise TapeUnsupported("nopesac_tape_create failed (rc=%d): %s" % (rc, msg.decode() if msg else ""))
        self._lib, self._h, self._graph = lib, handle, graph     # the graph's nodes own the argument blocks the tape points at
        self.counts = {"kernels": counts[0], "memsets": counts[1], "memcpys": counts[2], "streams": counts[3]}
        self._get_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)

    def replay(self, stream: int = None, sides=None):
        """Enqueue the recorded launches on `stream` (a raw hipStream_t as int; default: torch's current stream).  `sides`: streams
        (torch.cuda.Stream or raw ints) for the tape's side chains instead of its own ones - see streams.StreamSet."""
        if stream is None:
            stream = (self._get_stream(torch._C._cuda_getDevice()) if self._get_stream is not None
                      else torch.cuda.current_stream().cuda_stream)
        if sides:
            raw = [int(getattr(s, "cuda_stream", s)) for s in sides]
            arr = (ctypes.c_void_p * len(raw))(*raw)
            rc = self._lib.nopesac_tape_replay_on(self._h, stream, arr, len(raw))
        else:
            rc = self._lib.nopesac_tape_replay(self._h, stream)
        if rc != 0:
            _lib.check(rc, "nopesac_tape_replay")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self._lib.nopesac_tape_destroy(h)
            except Exception:
                pass
