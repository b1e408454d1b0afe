"""Writes the JPEG fixtures of tests/test_jpeg_cpu.py / tests/test_jpeg_gpu.py: small files encoded by Pillow (libjpeg-turbo) in the
configurations the decoder supports, and what Pillow - the reference's decoder (detectron2 utils.read_image, planercnn_transforms.py:
210-227) - decodes them to.  Run in the build container:  python -m oracle.gen_jpeg_golden
Fixtures are data: the .jpg files and tests/golden/jpeg_decoded.npz (uint8 arrays)."""
import io
import os

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "jpeg")


def picture(h, w, seed):
    """structured content (gradients, edges, a little noise): like a camera frame, every Huffman code length and run length shows up"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    a = np.stack([128 + 100 * np.sin(xx / (3.0 + seed) + yy / 17.0), 128 + 90 * np.cos(yy / (5.0 + seed)) * np.sin(xx / 29.0),
                  255 * ((xx // 7 + yy // 5) % 2)], -1)
    a[h // 4:h // 2, w // 5:w // 2] = (240, 20, 60)
    a += rng.normal(0, 6 + 2 * seed, a.shape)
    return np.clip(a, 0, 255).astype(np.uint8)


CASES = [  # name, height, width, save() options
    ("s420_q85_61x83", 61, 83, dict(quality=85, subsampling=2)),
    ("s420_q30_48x64", 48, 64, dict(quality=30, subsampling=2)),
    ("s420_q97_17x5", 17, 5, dict(quality=97, subsampling=2)),
    ("s420_q90_2x3", 2, 3, dict(quality=90, subsampling=2)),
    ("s420_q75_rst_100x76", 100, 76, dict(quality=75, subsampling=2, restart_marker_blocks=4)),
    ("s420_q80_opt_121x162", 121, 162, dict(quality=80, subsampling=2, optimize=True)),
    ("s422_q85_37x53", 37, 53, dict(quality=85, subsampling=1)),
    ("s422_q60_rst_64x49", 64, 49, dict(quality=60, subsampling=1, restart_marker_rows=1)),
    ("s422_q95_9x4", 9, 4, dict(quality=95, subsampling=1)),
    ("s444_q85_33x41", 33, 41, dict(quality=85, subsampling=0)),
    ("s444_q100_16x16", 16, 16, dict(quality=100, subsampling=0)),
    ("s444_q50_rst_40x72", 40, 72, dict(quality=50, subsampling=0, restart_marker_blocks=7)),
    ("gray_q80_45x31", 45, 31, dict(quality=80)),
    ("gray_q92_rst_24x40", 24, 40, dict(quality=92, restart_marker_blocks=5)),
    ("s420_q88_scannet_like_242x324", 242, 324, dict(quality=88, subsampling=2)),     # 968 x 1296 / 4: odd MCU rows at the bottom edge
]


def main():
    os.makedirs(OUT, exist_ok=True)
    dec = {}
    for i, (name, h, w, opt) in enumerate(CASES):
        a = picture(h, w, i)
        if name.startswith("gray"):
            a = a[..., 0]
        b = io.BytesIO()
        Image.fromarray(a).save(b, format="JPEG", **opt)
        data = b.getvalue()
        with open(os.path.join(OUT, name + ".jpg"), "wb") as f:
            f.write(data)
        dec[name] = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
    # one file outside the supported subset (progressive): the decoder must refuse it
    b = io.BytesIO()
    Image.fromarray(picture(40, 56, 99)).save(b, format="JPEG", quality=80, progressive=True)
    with open(os.path.join(OUT, "unsupported_progressive_40x56.jpg"), "wb") as f:
        f.write(b.getvalue())
    dec["unsupported_progressive_40x56"] = np.asarray(Image.open(io.BytesIO(b.getvalue())).convert("RGB"))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "jpeg_decoded.npz"), **dec)
    print("wrote %d files, %d bytes of JPEG" % (len(dec), sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))))


if __name__ == "__main__":
    main()
