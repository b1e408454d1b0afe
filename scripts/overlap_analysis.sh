#!/bin/bash
# How busy is the GPU in the steady state of the default (overlapped) bench?  kernel trace of 24 steps, analysis of the middle ones.
R=$PWD; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_ovl -o bench -- python $R/bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-accuracy "$@" > $R/gpurun_out/prof_ovl.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/prof_ovl/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if (("stem_fused_kernel<1>" in r["Kernel_Name"] or "stem_fused_kernel<2>" in r["Kernel_Name"]) or "preprocess" in r["Kernel_Name"])]
# the last 27 step starts = 24 timed steps + 3 instrumented ones; analyse timed steps 6 .. 20
a, b = starts[-27 + 6], starts[-27 + 20]
sel = rows[a:b]
t0 = int(sel[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in sel)
def wgs(r):
    g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    w = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    return g // max(w, 1)
big = [r for r in sel if wgs(r) >= 512]
def union(rs):
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rs)
    tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
print("steady window %.2f ms for 14 steps = %.3f ms/step" % ((t1 - t0) / 1e6, (t1 - t0) / 1e6 / 14))
print("some kernel running: %.3f   a chip-filling (>= 512 workgroups) kernel running: %.3f" % (union(sel) / (t1 - t0), union(big) / (t1 - t0)))
ev = []
for r in big: ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort(); cur = 0; last = t0; hist = collections.Counter()
for t, d in ev:
    hist[cur] += t - last; last = t; cur += d
tot = sum(hist.values())
print("number of chip-filling kernels resident at once:", {k: round(v / tot, 3) for k, v in sorted(hist.items())})
print("sum of chip-filling kernel durations per step: %.3f ms; of all kernels: %.3f ms" % (
    sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in big) / 1e6 / 14, sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel) / 1e6 / 14))
PY
