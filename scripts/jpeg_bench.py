"""GPU box: throughput of the device JPEG decoder on ScanNet-sized frames (968 x 1296, 4:2:0) next to Pillow on one host core.
usage: jpeg_bench.py [images per batch = 64]"""
import io, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from PIL import Image  # noqa: E402
from nopesac_amd import jpeg, ops  # noqa: E402

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rng = np.random.default_rng(0)
yy, xx = np.mgrid[0:968, 0:1296].astype(np.float32)


def frame(i, **opt):
    a = np.stack([128 + 90 * np.sin(xx / (40 + i) + yy / 90), 128 + 70 * np.cos(yy / (35 + i)) * np.sin(xx / 140), 120 + 100 * ((xx // 160 + yy // 120) % 2)], -1)
    a = np.clip(a + rng.normal(0, 3.0, a.shape), 0, 255).astype(np.uint8)
    b = io.BytesIO()
    Image.fromarray(a).save(b, format="JPEG", **opt)
    return b.getvalue()


for label, opt in (("no restart markers", dict(quality=90, subsampling=2)), ("restart marker per MCU row", dict(quality=90, subsampling=2, restart_marker_rows=1))):
    base = [frame(i, **opt) for i in range(8)]
    files = [base[i % 8] for i in range(N)]
    t0 = time.perf_counter()
    for f in base:
        np.asarray(Image.open(io.BytesIO(f)).convert("RGB"))
    pil_ms = 1e3 * (time.perf_counter() - t0) / len(base)
    t0 = time.perf_counter()
    infos = [jpeg.parse(f) for f in files]
    parse_ms = 1e3 * (time.perf_counter() - t0) / N
    st = {}
    outs = jpeg.decode_batch(files, dev, infos=infos, stats=st)
    ref = np.asarray(Image.open(io.BytesIO(files[0])).convert("RGB"))
    assert np.array_equal(outs[0].cpu().numpy(), ref)
    if st:
        print("   self-synchronising decoder: %d of %d images settled; lanes that moved per pass (all images): %s of %d lanes" % (
            int(st["par_done"].sum()), N, st["changed"].sum(1).tolist(), sum(-(-i.seg_bytes[0] * 8 // (jpeg.SUB_WORDS * 32)) for i in infos)), flush=True)
    print("%s: %d KB per file; Pillow %.2f ms per image on one core; host parse + stuffing removal %.3f ms per image" % (label, len(base[0]) // 1024, pil_ms, parse_ms), flush=True)
    for n_streams in (1, 2, 4):
        streams = [torch.cuda.Stream() for _ in range(n_streams)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3
        for r in range(reps):
            for s in streams:
                with torch.cuda.stream(s):
                    outs = jpeg.decode_batch(files, dev, infos=infos)
                    small = [ops.resize_bilinear_u8(o, 480, 640) for o in outs[:2]]     # (the mapper's next step, two of them)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print("   %d batch(es) of %d images in flight: %.1f ms per round, %.0f images/s = %.0f pairs/s" % (n_streams, N, 1e3 * el / reps, reps * n_streams * N / el, reps * n_streams * N / el / 2), flush=True)
    e0, e1, e2, e3 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
