"""Thread scaling of the batch PNG decoder (data.read_png_files) into a reused and into a fresh output buffer (page faults)."""
import os, sys, time, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from PIL import Image
from nopesac_amd import data
rng = np.random.default_rng(0)
yy, xx = np.mgrid[0:480, 0:640].astype(np.float32)
td = tempfile.mkdtemp()
paths = []
for i in range(8):
    a = np.stack([128 + 90 * np.sin(xx / (20 + i) + yy / 45), 128 + 70 * np.cos(yy / (17 + i)) * np.sin(xx / 70), 120 + 100 * ((xx // 80 + yy // 60) % 2)], -1)
    p = os.path.join(td, "f%d.png" % i)
    Image.fromarray(np.clip(a + rng.normal(0, 3.0, a.shape), 0, 255).astype(np.uint8)).save(p)
    paths.append(p)
N = 256
files = [paths[i % 8] for i in range(N)]
pre = torch.empty(N, 3, 480, 640, dtype=torch.uint8)
pre.fill_(1)
for thr in (1, 4, 8, 16, 32, 64, 128):
    n = min(N, max(8, thr * 4))
    data.read_png_files(files[:n], "BGR", 480, 640, threads=thr, out=pre[:n])
    t0 = time.perf_counter(); data.read_png_files(files[:n], "BGR", 480, 640, threads=thr, out=pre[:n]); t1 = time.perf_counter()
    data.read_png_files(files[:n], "BGR", 480, 640, threads=thr); t2 = time.perf_counter()
    print("threads %3d  n %3d  reused out: %7.0f img/s (%.2f ms/img/thread)   fresh out: %7.0f img/s" % (thr, n, n / (t1 - t0), 1e3 * (t1 - t0) * thr / n, n / (t2 - t1)), flush=True)
