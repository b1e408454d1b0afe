"""Does the 'tiny kernel behind a long spin kernel' probe tell hardware queues apart?  Several formulations."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nopesac_amd import ops
dev = torch.device("cuda:0")
streams = [torch.cuda.Stream() for _ in range(8)]
outa = torch.zeros(2, device=dev, dtype=torch.int64)
outb = torch.zeros(2, device=dev, dtype=torch.int64)
from nopesac_amd import _lib
L = _lib.load()
def probe(out, cycles, s):
    _lib.check(L.nopesac_clock_probe(out.data_ptr(), int(cycles), s.cuda_stream), "probe")
torch.cuda.synchronize()
for s in streams:            # warm
    probe(outa, 1000, s)
torch.cuda.synchronize()
print("query formulation: a = streams[0] spins 2 ms; b = streams[k]: is a still busy when b is done?")
for k in range(1, 8):
    a, b = streams[0], streams[k]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    probe(outa, 5_000_000, a)
    probe(outb, 1000, b)
    b.synchronize()
    tb = time.perf_counter() - t0
    busy = not a.query()
    a.synchronize()
    ta = time.perf_counter() - t0
    print("  k=%d: b done after %.3f ms, a busy then: %s, a done after %.3f ms" % (k, 1e3 * tb, busy, 1e3 * ta))
print("event formulation")
for k in range(1, 8):
    a, b = streams[0], streams[k]
    torch.cuda.synchronize()
    s0, s1, t1 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    s0.record(a); probe(outa, 5_000_000, a); s1.record(a)
    probe(outb, 1000, b); t1.record(b)
    s1.synchronize(); t1.synchronize()
    print("  k=%d: spin %.3f ms, b's event at %.3f ms" % (k, s0.elapsed_time(s1), s0.elapsed_time(t1)))
