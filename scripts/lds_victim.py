"""Which kernel disturbs ransac_score_maps_kernel when it runs CONCURRENTLY on another stream?  (Round 3: with four batches in flight
two launches of the score-map kernel on identical inputs disagree; its global inputs are intact, only the outputs that pass through
its LDS arrays are off.)  For every aggressor: stream B loops the aggressor, stream A loops the victim; the victim's outputs are
compared with a reference computed on an idle GPU.  usage: lds_victim.py [iters]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nopesac_amd import ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
B, nq = 32, 50
g = torch.Generator().manual_seed(3)
rn = lambda *s: torch.randn(*s, generator=g).to(dev)
geo_local = rn(B, nq, 6)
rot_raw, trans_raw = rn(B, nq, 4), rn(B, nq, 3)
init_rot = torch.nn.functional.normalize(rn(B, 4), dim=-1)
init_trans = rn(B, 3)
m = torch.full((B,), 32, device=dev, dtype=torch.int32)


def victim():
    return ops.ransac_score_maps(geo_local, rot_raw, trans_raw, init_rot, init_trans, m, diagnostics=False)


ref = victim()
torch.cuda.synchronize()
bf = lambda *s: (torch.randn(*s, device=dev) * 0.1).bfloat16()
model = bench.build_model(dev, nq, "bfloat16")
routing = os.path.join(ROOT, "profiles", "routing_r5.json")
if os.path.exists(routing):
    ops.TUNER.load(routing)
raw = torch.randint(0, 256, (64, 3, 480, 640), device=dev).float()
x256 = bf(64, 30, 40, 256)
w3 = bf(256, 3, 3, 256)
sc, bi = torch.ones(256, device=dev), torch.zeros(256, device=dev)
xb3 = bf(64, 60, 80, 128)
w3f = ops.mfma_fragment_major(bf(512, 128))
w1f = ops.mfma_fragment_major(bf(128, 512))
s512, b512, s128, b128 = torch.ones(512, device=dev), torch.zeros(512, device=dev), torch.ones(128, device=dev), torch.zeros(128, device=dev)
res3 = bf(64, 60, 80, 512)
x64 = bf(64, 120, 160, 64).relu()
w64 = bf(64, 3, 3, 64)
s64, b64 = torch.ones(64, device=dev), torch.zeros(64, device=dev)
forced = bench.make_forced(B, 32, nq, dev, 7)


def p8():
    rc = ops._L().nopesac_conv2d_nhwc_p8(x256.data_ptr(), w3.data_ptr(), sc.data_ptr(), bi.data_ptr(), None, torch.empty_like(x256).data_ptr(), 64, 30, 40, 256, 256,
                                         3, 3, 1, 1, 256, 256, 0, ops.ACT_RELU, ops.BF16, 32, ops._stream())
    assert rc == 0


def canary(lds_bytes, wgs=512, spin=40000, rounds=4):
    cnt = torch.zeros(1, device=dev, dtype=torch.int32)
    log = torch.zeros(16, 4, device=dev, dtype=torch.int32)
    rc = ops._L().nopesac_lds_canary(wgs, lds_bytes, spin, rounds, cnt.data_ptr(), log.data_ptr(), 16, ops._stream())
    assert rc == 0, ops._L().nopesac_last_error()
    return cnt, log


aggressors = {
    "nothing": lambda: None,
    "p8 conv 3x3 256": p8,
    "rt4 tail res3": lambda: ops.bottleneck_tail(xb3, w3f, s512, b512, residual=res3, w1=w1f, s1=s128, b1=b128),
    "conv3x3_c64": lambda: ops.conv3x3_c64(x64, w64, s64, b64),
    "backbone (whole)": lambda: model.backbone(None, raw=(raw, model.pixel_mean, model.pixel_std)),
    "whole forward": lambda: model.forward_tensors(None, B, 480, 640, forced=forced, raw_images=raw),
}
aggressors["tight canary (barriers)"] = lambda: canary(9264, wgs=4096, spin=0, rounds=50)
aggressors["canary, 64 KB LDS"] = lambda: canary(65536, wgs=2048, spin=0, rounds=50)
aggressors["canary, spinning"] = lambda: canary(9264, wgs=2048, spin=20000, rounds=2)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
with torch.no_grad():
    model.forward_tensors(None, B, 480, 640, forced=forced, raw_images=raw)          # warm caches
torch.cuda.synchronize()
for name, agg in aggressors.items():
    bad, worst = 0, 0.0
    with torch.no_grad():
        for it in range(iters):
            with torch.cuda.stream(sb):
                agg()
                agg()
            outs = []
            with torch.cuda.stream(sa):
                for _ in range(8):
                    outs.append(victim())
            torch.cuda.synchronize()
            for o in outs:
                d = float((o["normal_score"] - ref["normal_score"]).abs().max() + (o["param_score"] - ref["param_score"]).abs().max())
                if d != 0.0:
                    bad += 1
                    worst = max(worst, d)
                    if bad == 1:
                        idx = ((o["normal_score"] != ref["normal_score"]) | (o["param_score"] != ref["param_score"])).nonzero()[:24].tolist()
                        print("    first off launch: differing (pair, hypothesis h, plane j): %s" % idx)
                        b0, h0, j0 = idx[0]
                        print("    values there: normal %.6g (ref %.6g) param %.6g (ref %.6g)" % (float(o["normal_score"][b0, h0, j0]), float(ref["normal_score"][b0, h0, j0]),
                                                                                               float(o["param_score"][b0, h0, j0]), float(ref["param_score"][b0, h0, j0])))
    print("%-22s victim launches off: %3d of %d   worst |diff| %.3g" % (name, bad, iters * 8, worst), flush=True)

# ---- other victims next to the rt4 tail: how general is it?
xs = torch.randn(32, 51, 50, device=dev)
ln_g, ln_b = torch.ones(256, device=dev), torch.zeros(256, device=dev)
xl = torch.randn(1600, 256, device=dev)
victims = {
    "torch clone (326 KB)": lambda: xs.clone(),
    "torch x * 2 + 1": lambda: xs * 2 + 1,
    "torch exp(-x)": lambda: torch.exp(-xs),
    "torch acos(clamp(x))": lambda: torch.acos(xs.clamp(-1, 1)),
    "ops.softmax_rows": lambda: ops.softmax_rows(xl),
    "ops.layernorm": lambda: ops.layernorm(xl, ln_g, ln_b),
    "ops.normalize_rows": lambda: ops.normalize_rows(rot_raw.view(-1, 4).contiguous()),
}
for vname, vf in victims.items():
    vref = vf()
    torch.cuda.synchronize()
    for name in ("nothing", "rt4 tail res3", "conv3x3_c64"):
        agg = aggressors[name]
        bad = 0
        with torch.no_grad():
            for it in range(iters):
                with torch.cuda.stream(sb):
                    agg()
                    agg()
                with torch.cuda.stream(sa):
                    outs = [vf() for _ in range(8)]
                torch.cuda.synchronize()
                bad += sum(1 for o in outs if not torch.equal(o, vref))
        print("victim %-22s vs %-16s launches off: %3d of %d" % (vname, name, bad, iters * 8), flush=True)

# ---- barrier-tight canary: no spin, hundreds of write / barrier / read-what-OTHER-threads-wrote rounds
for name, agg in aggressors.items():
    tot, first = 0, None
    with torch.no_grad():
        for it in range(max(4, iters // 4)):
            with torch.cuda.stream(sb):
                agg()
                agg()
            with torch.cuda.stream(sa):
                cnt, log = canary(9264, wgs=256, spin=0, rounds=400)
            torch.cuda.synchronize()
            c = int(cnt[0])
            tot += c
            if c and first is None:
                first = [[int(v) & 0xFFFFFFFF for v in row] for row in log[:min(c, 6)].tolist()]
    print("tight canary vs %-20s mismatching dwords: %d %s" % (name, tot, "" if first is None else
          "first: " + "; ".join("wg %d dword %d want %08x got %08x" % tuple(r) for r in first)), flush=True)

# ---- generic LDS canaries of several sizes next to the same aggressors
for lds_bytes in (9264,):
    for name, agg in aggressors.items():
        tot, first = 0, None
        with torch.no_grad():
            for it in range(max(4, iters // 4)):
                with torch.cuda.stream(sb):
                    agg()
                    agg()
                with torch.cuda.stream(sa):
                    cnt, log = canary(lds_bytes)
                torch.cuda.synchronize()
                c = int(cnt[0])
                tot += c
                if c and first is None:
                    first = [[int(v) & 0xFFFFFFFF for v in row] for row in log[:min(c, 6)].tolist()]
        print("canary %6d B vs %-20s mismatching dwords: %d %s" % (lds_bytes, name, tot, "" if first is None else
              "first: " + "; ".join("wg %d dword %d want %08x got %08x" % tuple(r) for r in first)), flush=True)

