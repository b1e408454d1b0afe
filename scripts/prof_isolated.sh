#!/bin/bash
# Per-kernel cost WITHOUT overlap: one batch in flight, pose net on the main stream, kernel-trace stats, totals per step.
# Usage: bash scripts/prof_isolated.sh <tag> [bench args]
TAG=${1:-iso}; shift
R=$PWD; export TMPDIR=/tmp; cd /tmp
STEPS=6
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG} -o bench -- python $R/bench.py --steps $STEPS --warmup 2 --inflight 1 --single-stream --no-cpu-baseline --no-accuracy --no-boundary --no-other-configs --no-tape --no-fp32-path "$@" > $R/gpurun_out/prof_${TAG}.log 2>&1
cd $R
tail -1 gpurun_out/prof_${TAG}.log | cut -c1-200
python - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/prof_${TAG}/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# step boundaries = the first kernel of a step (raw-input stem, or preprocess); the run ends with 3 instrumented steps
starts = [i for i, r in enumerate(rows) if (("stem_fused_kernel<1" in r["Kernel_Name"] or "stem_fused_kernel<2" in r["Kernel_Name"]) or "preprocess" in r["Kernel_Name"])]
n = $STEPS
sel = rows[starts[-(n + 3)]:starts[-3]]           # n full timed steps (skips the 3 instrumented ones)
tot = collections.defaultdict(lambda: [0, 0.0])
for r in sel:
    k = r["Kernel_Name"].split("(")[0].replace("void nps::", "").replace("nps::", "")[:90]
    tot[k][0] += 1
    tot[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
wall = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e3 / n
busy = sum(v[1] for v in tot.values()) / n
print("per step: wall %.1f us, kernel-busy %.1f us, launches %d" % (wall, busy, len(sel) // n))
with open("gpurun_out/iso_${TAG}.tsv", "w") as out:
    for k, (c, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        line = "%-92s n/step=%6.1f us/step=%9.1f avg_us=%8.2f" % (k, c / n, us / n, us / c)
        out.write(line + "\n")
for l in open("gpurun_out/iso_${TAG}.tsv").read().split("\n")[:45]: print(l)
PY
