"""GPU box: random JPEG encodings (Pillow) decoded by the device decoder and compared with Pillow's decode, bit for bit.
Sizes 1..700, qualities 1..100, 4:4:4 / 4:2:2 / 4:2:0 / gray, optimised Huffman tables, restart markers, noise / smooth / flat content.
usage: jpeg_fuzz.py [cases = 300] [seed = 0]"""
import io, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from PIL import Image  # noqa: E402
from nopesac_amd import jpeg  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
files, refs, desc = [], [], []
for c in range(n_cases):
    h, w = int(rng.integers(1, 700)), int(rng.integers(1, 700))
    if rng.random() < 0.15:
        h, w = int(rng.integers(1, 20)), int(rng.integers(1, 20))
    kind = rng.integers(0, 4)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    if kind == 0:
        a = rng.integers(0, 256, (h, w, 3))
    elif kind == 1:
        a = np.stack([128 + 120 * np.sin(xx / 9 + yy / 17), 128 + 100 * np.cos(yy / 5), 255 * ((xx // 7 + yy // 3) % 2)], -1) + rng.normal(0, rng.uniform(0, 30), (h, w, 3))
    elif kind == 2:
        a = np.full((h, w, 3), rng.integers(0, 256, 3))
    else:
        a = np.cumsum(rng.normal(0, 3, (h, w, 3)), 1) + 128
    a = np.clip(a, 0, 255).astype(np.uint8)
    opt = {"quality": int(rng.integers(1, 101))}
    gray = rng.random() < 0.15
    if gray:
        a = a[..., 0]
    else:
        opt["subsampling"] = int(rng.integers(0, 3))
    if rng.random() < 0.3:
        opt["optimize"] = True
    r = rng.random()
    if r < 0.2:
        opt["restart_marker_blocks"] = int(rng.integers(1, 40))
    elif r < 0.3:
        opt["restart_marker_rows"] = int(rng.integers(1, 4))
    b = io.BytesIO()
    try:
        Image.fromarray(a).save(b, format="JPEG", **opt)
    except OSError:                                          # (Pillow's encoder refuses some restart / size combinations)
        continue
    files.append(b.getvalue())
    refs.append(np.asarray(Image.open(io.BytesIO(b.getvalue())).convert("RGB")))
    desc.append((h, w, gray, opt))
t0 = time.time()
bad = 0
n_cases = len(files)
for s in range(0, n_cases, 32):
    st = {}
    outs = jpeg.decode_batch(files[s:s + 32], dev, stats=st)
    outs2 = jpeg.decode_batch(files[s:s + 32], dev, parallel=False)
    for i, (o, o2) in enumerate(zip(outs, outs2)):
        g = o.cpu().numpy()
        if g.shape != refs[s + i].shape or not np.array_equal(g, refs[s + i]) or not torch.equal(o, o2):
            bad += 1
            print("MISMATCH", desc[s + i], "bytes", len(files[s + i]), "settled", int(st["par_done"][i]) if st else None, flush=True)
print("%d cases, %d mismatching, %.1f s; file sizes %d .. %d bytes" % (n_cases, bad, time.time() - t0, min(map(len, files)), max(map(len, files))))
sys.exit(1 if bad else 0)
