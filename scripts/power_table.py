"""Round 5 (verdict item 4): package power and engine clock per kernel.  For each workload: run it back to back for ~4 s, sample
`rocm-smi --showpower --showclocks` every 0.25 s from a thread AND read the engine clock from inside the GPU (ops.clock_probe: shader
cycles per 100 MHz reference tick, on a side stream), print kernel -> W, MHz (smi), MHz (probe), TFLOP/s or TB/s.
Workloads: the p8 kernel on its 1200-tile and 300-tile 3x3 layers with random and with all-zero operands (same instruction stream, no
data toggling), the 256x128-tile kernel (1.25 LDS reads and 1.5x the LDS-DMA issues per MFMA of p8), the weights-from-L2 kernel (half the LDS
traffic of p8 per MFMA), the HBM-bound expand conv, a fused res2 tail, and the whole four-in-flight benchmark loop (bench.py as a
subprocess)."""
import json
import os
import re
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nopesac_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.load()
samples, stop = [], [False]


def sampler():
    while not stop[0]:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            pw = re.findall(r"(?:Power|Graphics Package Power)[^:\n]*\(W\):\s*([0-9.]+)", out)
            ck = re.findall(r"sclk clock level:.*?\((\d+)Mhz\)", out)
            samples.append((time.time(), float(pw[0]) if pw else None, int(ck[0]) if ck else None, None if pw else out[-300:]))
        except Exception as e:  # noqa: BLE001
            samples.append((time.time(), None, None, repr(e)))
        time.sleep(0.25)


def window(t0, t1):
    w = [s for s in samples if t0 + 0.7 <= s[0] <= t1]
    pw = [s[1] for s in w if s[1] is not None]
    ck = [s[2] for s in w if s[2] is not None]
    return (sum(pw) / len(pw) if pw else None, max(pw) if pw else None, sum(ck) / len(ck) if ck else None, len(w))


def conv_case(kind, B, H, W, Cin, Cout, k, s, zero=False, res=False):
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    x = torch.zeros(B, H, W, Cin, device=dev).bfloat16() if zero else torch.randn(B, H, W, Cin, device=dev).bfloat16()
    w = torch.zeros(Cout, k, k, Cin, device=dev).bfloat16() if zero else (torch.randn(Cout, k, k, Cin, device=dev) / (Cin * k * k) ** 0.5).bfloat16()
    sc, bi = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
    r = torch.randn(B, Ho, Wo, Cout, device=dev).bfloat16() if res else None
    y = torch.empty(B, Ho, Wo, Cout, device=dev, dtype=torch.bfloat16)
    wf = ops._frag_weights(w) if kind == "bfrag" else None
    st = torch.cuda.current_stream().cuda_stream
    keep = (x, w, sc, bi, r, y, wf)
    if kind == "p8":
        fn = lambda: L.nopesac_conv2d_nhwc_p8(x.data_ptr(), w.data_ptr(), sc.data_ptr(), bi.data_ptr(), r.data_ptr() if res else None, y.data_ptr(), B, H, W, Cin, Cout,
                                              k, k, s, pad, Cin, Cout, Cout if res else 0, ops.ACT_RELU, 1, 32, st)
    elif kind == "p8n":
        fn = lambda: L.nopesac_conv2d_nhwc_p8n(x.data_ptr(), w.data_ptr(), sc.data_ptr(), bi.data_ptr(), y.data_ptr(), B, H, W, Cin, Cout, k, k, s, pad, Cin, Cout,
                                               ops.ACT_RELU, 32, st)
    else:
        fn = lambda: L.nopesac_conv2d_nhwc_bfrag(x.data_ptr(), wf.data_ptr(), sc.data_ptr(), bi.data_ptr(), None, y.data_ptr(), B, H, W, Cin, Cout, k, k, s, pad,
                                                 Cin, Cout, 0, ops.ACT_RELU, 1, 3 + 256, st)
    flops = 2.0 * B * Ho * Wo * Cout * k * k * Cin
    nbytes = (x.numel() + w.numel() + y.numel() * (2 if res else 1)) * 2
    return fn, flops, nbytes, keep


def run(name, fn, flops, nbytes, seconds=4.0):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    probe_stream = torch.cuda.Stream(device=dev)
    probes = []
    t0 = time.time()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 0
    while time.time() - t0 < seconds:
        for _ in range(40):
            fn()
        n += 40
        probes.append(ops.clock_probe(300000, probe_stream))
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    t1 = time.time()
    us = 1e3 * e0.elapsed_time(e1) / n
    vals = sorted(100.0 * float(t[0]) / float(t[1]) for t in (q.cpu() for q in probes) if int(t[1]) > 0)
    pw, pmax, ck, ns = window(t0, t1)
    row = {"workload": name, "us_per_launch": round(us, 1), "TFLOP/s": round(flops / us / 1e6, 1), "TB/s_algorithmic": round(nbytes / us / 1e6, 2),
           "power_W_mean": None if pw is None else round(pw, 0), "power_W_max": pmax, "sclk_MHz_smi": None if ck is None else round(ck, 0),
           "sclk_MHz_probe_median": round(vals[len(vals) // 2], 0) if vals else None, "smi_samples": ns}
    print(json.dumps(row), flush=True)
    time.sleep(1.5)                                            # let the package cool / the clock relax between workloads
    return row


th = threading.Thread(target=sampler, daemon=True)
th.start()
time.sleep(2.0)
idle = window(time.time() - 2.0, time.time())
print(json.dumps({"workload": "idle", "power_W_mean": idle[0], "sclk_MHz_smi": idle[2], "raw": samples[-1][3]}), flush=True)
rows = []
CASES = [("p8 3x3 256->256 @60x80 (1200 tiles), random operands", ("p8", 64, 60, 80, 256, 256, 3, 1)),
         ("p8 same layer, ALL-ZERO operands", ("p8", 64, 60, 80, 256, 256, 3, 1, True)),
         ("p8 3x3 256->256 @30x40 (300 tiles)", ("p8", 64, 30, 40, 256, 256, 3, 1)),
         ("p8n (256x128 tiles) 3x3 256->256 @60x80", ("p8n", 64, 60, 80, 256, 256, 3, 1)),
         ("p8n 3x3 128->128 @60x80 (res3 conv2)", ("p8n", 64, 60, 80, 128, 128, 3, 1)),
         ("bfrag<3,64> (weights from L2) 3x3 256->256 @60x80", ("bfrag", 64, 60, 80, 256, 256, 3, 1)),
         ("p8 1x1 256->1024 + residual @30x40 (HBM-bound expand conv)", ("p8", 64, 30, 40, 256, 1024, 1, 1, False, True))]
for name, a in CASES:
    fn, fl, nb, keep = conv_case(*a)
    rows.append(run(name, fn, fl, nb))
    del keep
# the whole benchmark loop, four batches in flight
t0 = time.time()
r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "500", "--warmup", "10", "--no-cpu-baseline", "--no-boundary", "--no-fp32-path",
                    "--no-accuracy", "--no-other-configs", "--no-tape"], capture_output=True, text=True, cwd=ROOT)
t1 = time.time()
line = [l for l in r.stdout.splitlines() if l.startswith("{")]
if line:
    d = json.loads(line[-1])
    # the timed region = the last steps * ms_per_step seconds before the instrumented steps; take the window's second half
    pw, pmax, ck, ns = window(t1 - 0.6 * (t1 - t0), t1 - 3.0)
    print(json.dumps({"workload": "bench.py four batches in flight (500 steps)", "pairs_per_s": d["value"], "ms_per_step": d["ms_per_step"],
                      "power_W_mean": None if pw is None else round(pw, 0), "power_W_max": pmax, "sclk_MHz_smi": None if ck is None else round(ck, 0),
                      "sclk_MHz_probe_median": (d["roofline"].get("engine_clock") or {}).get("sclk_mhz_under_benchmark_load"), "smi_samples": ns}), flush=True)
else:
    print("bench failed", r.stderr[-500:])
stop[0] = True
