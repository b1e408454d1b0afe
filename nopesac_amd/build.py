"""Build libnopesac_hip.so (gfx950) in-tree with hipcc.  `python -m nopesac_amd.build [--force]`."""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libnopesac_hip.so")
OBJ = os.path.join(CSRC, "_obj")
SOURCES = ["capi.hip", "conv_igemm.hip", "conv_p8.hip", "conv_p8n.hip", "stem.hip", "conv3x3_c64.hip", "conv3x3_halo.hip", "pwchain.hip", "gnn_layer.hip", "enc_tail.hip", "mask_head.hip", "resize.hip", "rle.hip", "elementwise.hip", "attention.hip", "postselect.hip", "matcher.hip",
           "ransac.hip", "refine_bwd.hip", "mlp_chain.hip", "tape.hip", "posenet_branch.hip", "jpeg.hip", "jpeg_host.hip", "png_host.hip"]
# -packed-fp32-ops: NO v_pk_{fma,mul,add}_f32 anywhere in the library.  Round-3 finding (scripts/lds_victim.py): a wave
# executing packed-f32 VALU instructions gets the results of its lanes 48-63 corrupted when a wave of ANOTHER kernel issues MFMAs on the
# same SIMD - ransac_score_maps_kernel (whose f32 math the SLP vectoriser had packed) returned different scores on identical inputs in
# 40-60 % of its launches next to the bottleneck-tail / 3x3 kernels of other batches in flight, never alone.  The compiler only knows the
# hazards inside one wave; with the feature off it emits the scalar-f32 forms.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
if os.environ.get("NOPESAC_ALLOW_PACKED_FP32"):           # A/B builds only
    FLAGS = FLAGS[:5]
EXTRA_FLAGS = {}                                          # per-file additions
if os.environ.get("NOPESAC_HIPCC_EXTRA"):                 # A/B builds only (e.g. -DSOME_TUNING_MACRO=1): part of the digest, so a rebuild follows
    FLAGS = FLAGS + os.environ["NOPESAC_HIPCC_EXTRA"].split()


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm with gfx950 support)")


def _have_zlib(cc: str) -> bool:
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "z.cpp")
        with open(src, "w") as f:
            f.write("#include <zlib.h>\nint main() { z_stream s{}; return inflateInit(&s) == Z_OK ? inflateEnd(&s) : 1; }\n")
        r = subprocess.run([cc, "-x", "c++", src, "-o", os.path.join(d, "z"), "-lz"], capture_output=True, text=True)
        return r.returncode == 0


def _digest() -> str:
    h = hashlib.sha256((" ".join(FLAGS) + repr(sorted(EXTRA_FLAGS.items()))).encode())
    for name in sorted(os.listdir(CSRC)) + ["../../include/nopesac_hip.h"]:
        p = os.path.join(CSRC, name)
        if os.path.isfile(p) and (name.endswith((".hip", ".h"))):
            h.update(name.encode())
            h.update(open(p, "rb").read())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    stamp = os.path.join(OBJ, "digest.txt")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    cc = _hipcc()
    # zlib only serves png_host.hip's NOPESAC_PNG_ZLIB_INFLATE=1 A/B path (the decoder's own inflate needs nothing).  Decided ONCE, here, by a
    # test compile + link, and handed to the compiler and the linker together: -DNPS_HAVE_ZLIB=1 exactly when -lz is on the link line.
    have_zlib = _have_zlib(cc)
    per_build = {"png_host.hip": ["-DNPS_HAVE_ZLIB=%d" % int(have_zlib)]}       # (not part of the digest: a property of the box, not of the sources)

    def compile_one(src):
        obj = os.path.join(OBJ, src.replace(".hip", ".o"))
        cmd = [cc, *FLAGS, *EXTRA_FLAGS.get(src, []), *per_build.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, r.stderr[-4000:]))
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    zlib = ["-lz"] if have_zlib else []
    r = subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, *zlib], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr[-4000:])
    with open(stamp, "w") as f:
        f.write(dig)
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
