"""Which side of the packed-f32 observation carries the effect?  Run with the LIBRARY built with packed-f32 enabled
(NOPESAC_HIPCC_EXTRA="-Xclang -target-feature -Xclang +packed-fp32-ops" python -m nopesac_amd.build --force):
  (i)  the library's victim (ransac_score_maps, 142 v_pk_*_f32) next to the stand-alone MFMA aggressors of repro_packed_fp32_hazard.hip
  (ii) the stand-alone packed victims next to the library's aggressors (res3 tail, conv3x3_c64)
usage: repro_packed_fp32_mix.py <path to the shared library built from repro_packed_fp32_hazard.hip with -DREPRO_SHARED>"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nopesac_amd import ops  # noqa: E402
R = ctypes.CDLL(sys.argv[1])
for n in ("repro_victim_packed", "repro_victim_forms", "repro_victim_trans", "repro_aggressor_mfma", "repro_aggressor_agpr"):
    getattr(R, n).argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    getattr(R, n).restype = None
dev = torch.device("cuda:0")
B, nq, iters = 32, 50, 16
g = torch.Generator().manual_seed(3)
rn = lambda *s: torch.randn(*s, generator=g).to(dev)
geo_local, rot_raw, trans_raw = rn(B, nq, 6), rn(B, nq, 4), rn(B, nq, 3)
init_rot, init_trans = torch.nn.functional.normalize(rn(B, 4), dim=-1), rn(B, 3)
m = torch.full((B,), 32, device=dev, dtype=torch.int32)
lib_victim = lambda: ops.ransac_score_maps(geo_local, rot_raw, trans_raw, init_rot, init_trans, m, diagnostics=False)
bf = lambda *s: (torch.randn(*s, device=dev) * 0.1).bfloat16()
xb3, res3 = bf(64, 60, 80, 128), bf(64, 60, 80, 512)
w3f, w1f = ops.mfma_fragment_major(bf(512, 128)), ops.mfma_fragment_major(bf(128, 512))
s512, b512, s128, b128 = torch.ones(512, device=dev), torch.zeros(512, device=dev), torch.ones(128, device=dev), torch.zeros(128, device=dev)
x64, w64 = bf(64, 120, 160, 64).relu(), bf(64, 3, 3, 64)
s64, b64 = torch.ones(64, device=dev), torch.zeros(64, device=dev)
sink = torch.zeros(16, device=dev)
R.repro_aggressor_c64like.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
R.repro_aggressor_c64like.restype = None
wfrag = (torch.randn(36 * 64 * 8, device=dev) * 0.1).bfloat16()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
st = lambda s: ctypes.c_void_p(s.cuda_stream)
lib_aggr = {"library res3 tail": lambda: ops.bottleneck_tail(xb3, w3f, s512, b512, residual=res3, w1=w1f, s1=s128, b1=b128),
            "library conv3x3_c64": lambda: ops.conv3x3_c64(x64, w64, s64, b64)}
syn_aggr = {"stand-alone MFMA loop (2 blocks per CU)": lambda: R.repro_aggressor_mfma(sink.data_ptr(), 512, 400000, st(sb)),
            "stand-alone AGPR + LDS MFMA loop (3 blocks per CU)": lambda: R.repro_aggressor_agpr(sink.data_ptr(), 768, 100000, st(sb)),
            "stand-alone c64-like loop (VGPR acc, LDS fragments, ring)": lambda: R.repro_aggressor_c64like(wfrag.data_ptr(), sink.data_ptr(), 5120, 4, st(sb))}
ref = lib_victim()
torch.cuda.synchronize()
for name, agg in list(lib_aggr.items()) + list(syn_aggr.items()):
    bad = 0
    for it in range(iters):
        with torch.cuda.stream(sb):
            agg(); agg()
        with torch.cuda.stream(sa):
            outs = [lib_victim() for _ in range(8)]
        torch.cuda.synchronize()
        bad += sum(1 for o in outs if not (torch.equal(o["normal_score"], ref["normal_score"]) and torch.equal(o["param_score"], ref["param_score"])))
    print("library victim (ransac_score_maps) next to %-52s launches off %3d of %d" % (name, bad, iters * 8), flush=True)
for vname, vf in (("stand-alone dense packed victim", R.repro_victim_packed), ("stand-alone operand-form victim", R.repro_victim_forms),
                  ("stand-alone transcendental victim", R.repro_victim_trans)):
    out = torch.zeros(1024 * 256, 2, device=dev)
    vf(out.data_ptr(), 1024, 1000, st(torch.cuda.current_stream())); torch.cuda.synchronize()
    vref = out.clone()
    for name, agg in lib_aggr.items():
        bad = 0
        for it in range(iters):
            with torch.cuda.stream(sb):
                agg(); agg()
            res = []
            with torch.cuda.stream(sa):
                for _ in range(8):
                    o = torch.zeros_like(out)
                    vf(o.data_ptr(), 1024, 1000, st(sa))
                    res.append(o)
            torch.cuda.synchronize()
            bad += sum(1 for o in res if not torch.equal(o, vref))
        print("%-33s next to %-52s launches off %3d of %d" % (vname, name, bad, iters * 8), flush=True)
