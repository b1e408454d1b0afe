"""GPU box: is the MFMA-bound conv power-limited?  Runs one kernel configuration in a loop for a few seconds per data fill and
samples rocm-smi (power, sclk) meanwhile; prints TFLOP/s, average power and clock per (kernel, fill)."""
import os
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nopesac_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.load()
st = torch.cuda.current_stream().cuda_stream
B, H, W, Cin, Cout, k = 64, 60, 80, 256, 256, 3
flops = 2.0 * B * H * W * Cout * Cin * k * k
samples = []
stop = [False]


def sampler():
    while not stop[0]:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            pw = re.findall(r"Power \(W\):\s*([0-9.]+)", out)
            ck = re.findall(r"sclk clock level:.*?\((\d+)Mhz\)", out)
            samples.append((time.time(), float(pw[0]) if pw else None, int(ck[0]) if ck else None))
        except Exception as e:  # noqa: BLE001
            samples.append((time.time(), None, None))
        time.sleep(0.2)


def fill(name, shape, scale=1.0):
    g = torch.Generator(device=dev).manual_seed(1)
    if name == "zero":
        return torch.zeros(shape, device=dev).bfloat16()
    if name == "randn":
        return (torch.randn(shape, device=dev, generator=g) * scale).bfloat16()
    if name == "uniform":
        return ((torch.rand(shape, device=dev, generator=g) * 2 - 1) * scale).bfloat16()
    if name == "relu":
        return (torch.randn(shape, device=dev, generator=g).clamp_min(0) * scale).bfloat16()
    raise ValueError(name)


th = threading.Thread(target=sampler, daemon=True)
th.start()
y = torch.empty(B, H, W, Cout, device=dev, dtype=torch.bfloat16)
sc, bi = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
for kern in ("p8", "bfrag"):
    for f in ("randn", "uniform", "relu", "zero"):
        x = fill(f, (B, H, W, Cin))
        w = fill("randn" if f != "zero" else "zero", (Cout, k, k, Cin), (Cin * k * k) ** -0.5)
        wf = ops._frag_weights(w)

        def call():
            if kern == "p8":
                return L.nopesac_conv2d_nhwc_p8(x.data_ptr(), w.data_ptr(), sc.data_ptr(), bi.data_ptr(), None, y.data_ptr(), B, H, W, Cin, Cout, k, k, 1, 1,
                                                Cin, Cout, 0, 1, 1, 0, st)
            return L.nopesac_conv2d_nhwc_bfrag(x.data_ptr(), wf.data_ptr(), sc.data_ptr(), bi.data_ptr(), None, y.data_ptr(), B, H, W, Cin, Cout, k, k, 1, 1,
                                               Cin, Cout, 0, 1, 1, 3, st)
        for _ in range(20):
            call()
        torch.cuda.synchronize()
        t0 = time.time()
        n = 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        while time.time() - t0 < 3.0:
            for _ in range(50):
                call()
            n += 50
            torch.cuda.synchronize()
        e1.record()
        torch.cuda.synchronize()
        t1 = time.time()
        ms = e0.elapsed_time(e1) / n
        sel = [s for s in samples if t0 + 0.5 <= s[0] <= t1]
        pw = [s[1] for s in sel if s[1] is not None]
        ck = [s[2] for s in sel if s[2] is not None]
        print(f"{kern:6s} {f:8s} {ms * 1e3:7.1f} us  {flops / ms / 1e9:6.0f} TF   power {sum(pw) / max(len(pw), 1):6.0f} W (n={len(pw)})  sclk {sum(ck) / max(len(ck), 1):5.0f} MHz", flush=True)
stop[0] = True
out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower"], capture_output=True, text=True).stdout
print(out[-1500:])
