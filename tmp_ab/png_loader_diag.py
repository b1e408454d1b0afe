import os, sys, time, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from PIL import Image
from types import SimpleNamespace as NS
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
torch.zeros(1, device="cuda")
nthr = 32
cfgl = NS(INPUT=NS(FORMAT="BGR"), DATASETS=NS(ROOT_DIR="", TEST=("mp3d_test",)), DATALOADER=NS(NUM_WORKERS=nthr))
entries = [{v: {"file_name": paths[(2 * k + int(v)) % 8], "height": 480, "width": 640, "image_id": "%d_%s" % (k, v)} for v in "01"} for k in range(1024)]
lazy = data.LazyPairs(entries, data.PairMapper(cfgl, "mp3d_test", uint8=True, gpu_jpeg=False), workers=nthr)
# phases of one batch
for rep in range(4):
    t0 = time.perf_counter()
    buf = torch.empty((64, 3, 480, 640), dtype=torch.uint8, pin_memory=True)
    t1 = time.perf_counter()
    files = [n for e in entries[:32] for n in lazy.mapper.file_names(e)]
    data.read_png_files(files, "BGR", 480, 640, threads=nthr, out=buf)
    t2 = time.perf_counter()
    items = [lazy.mapper(e, images=[buf[2 * i], buf[2 * i + 1]]) for i, e in enumerate(entries[:32])]
    t3 = time.perf_counter()
    del buf, items
    print("alloc %.2f ms  decode %.2f ms  map %.2f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2)), flush=True)
# instrumented copy of the loader's structure
import gc
if os.environ.get('NOGC'): gc.disable()

import threading
from collections import deque
from concurrent.futures import ThreadPoolExecutor
lock = threading.Lock()
log = []
def png_batch(ents, k):
    t0 = time.perf_counter()
    files = [n for e in ents for n in lazy.mapper.file_names(e)]
    buf = torch.empty((len(files), 3, 480, 640), dtype=torch.uint8, pin_memory=True)
    t1 = time.perf_counter()
    with lock:
        t2 = time.perf_counter()
        data.read_png_files(files, "BGR", 480, 640, threads=nthr, out=buf)
        t3 = time.perf_counter()
    items = [lazy.mapper(e, images=[buf[2 * i], buf[2 * i + 1]]) for i, e in enumerate(ents)]
    t4 = time.perf_counter()
    log.append((k, t0, t1, t2, t3, t4))
    return items
chunks = [entries[i:i + 32] for i in range(0, 512, 32)]
T0 = time.perf_counter()
with ThreadPoolExecutor(max_workers=2) as pool:
    pending, nxt, got = deque(), 0, []
    while nxt < len(chunks) or pending:
        while nxt < len(chunks) and len(pending) < 3:
            pending.append(pool.submit(png_batch, chunks[nxt], nxt)); nxt += 1
        b = pending.popleft().result()
        got.append(time.perf_counter())
        del b
print("total %.1f ms for %d batches" % (1e3 * (time.perf_counter() - T0), len(chunks)))
for (k, t0, t1, t2, t3, t4), g in zip(sorted(log), got):
    if t3 - t2 > 0.012: print("batch %2d: start %.1f alloc %.2f lockwait %.2f decode %.2f map %.2f -> consumer got it at %.1f" % (k, 1e3 * (t0 - T0), 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3), 1e3 * (g - T0)))
