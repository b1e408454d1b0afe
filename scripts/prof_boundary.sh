#!/bin/bash
# Kernel totals per step of the drop-in boundary loop (scripts/queue_map.py: host images in, result dicts out, 4 batches in flight).
TAG=${1:-bd}; shift
R=$PWD; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/prof_${TAG} -o bd -- python $R/scripts/queue_map.py 0 0 "$@" > $R/gpurun_out/prof_${TAG}.log 2>&1
cd $R
tail -2 gpurun_out/prof_${TAG}.log
python - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/prof_${TAG}/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if ("stem_fused_kernel<1" in r["Kernel_Name"] or "stem_fused_kernel<2" in r["Kernel_Name"])]
n = 40
sel = rows[starts[-(n + 1)]:starts[-1]]
tot = collections.defaultdict(lambda: [0, 0.0])
for r in sel:
    k = r["Kernel_Name"].split("(")[0].replace("void nps::", "").replace("nps::", "")[:90]
    tot[k][0] += 1
    tot[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
wall = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e3 / n
print("per step: wall %.1f us, kernel-sum %.1f us, launches %d" % (wall, sum(v[1] for v in tot.values()) / n, len(sel) // n))
keys = ("rle", "cat", "Cat", "copy", "Copy", "cumsum", "scan", "fill", "Fill", "u8_to", "decode")
for k, (c, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    if any(s in k for s in keys):
        print("%-92s n/step=%6.1f us/step=%9.1f avg_us=%8.2f" % (k, c / n, us / n, us / c))
mc = glob.glob("gpurun_out/prof_${TAG}/**/*memory_copy_trace.csv", recursive=True)
if mc:
    rows = list(csv.DictReader(open(mc[0])))
    byk = collections.defaultdict(lambda: [0, 0.0, 0])
    for r in rows:
        d = r.get("Direction", "?")
        byk[d][0] += 1
        byk[d][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for d, (c, us, _) in byk.items():
        print("memcpy %-22s n=%6d total %.1f ms avg %.1f us" % (d, c, us / 1e3, us / c))
PY
