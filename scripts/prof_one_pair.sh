#!/bin/bash
# Where one model([pair]) call spends its time: kernel trace of scripts/one_pair_profile.py, totals per call over the last 40 calls.
# Usage: bash scripts/prof_one_pair.sh <tag> [tune]
TAG=${1:-b1}; shift
R=$PWD; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG} -o one -- python $R/scripts/one_pair_profile.py 50 "$@" > $R/gpurun_out/prof_${TAG}.log 2>&1
cd $R
grep "one pair" gpurun_out/prof_${TAG}.log
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
busy = sum(v[1] for v in tot.values()) / n
# time with no kernel running: union of the kernel intervals
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in sel)
cov, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        cov += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
cov += cur_e - cur_s
print("per call: wall %.1f us, kernel-sum %.1f us, some-kernel-running %.1f us, launches %d" % (wall, busy, cov / 1e3 / n, len(sel) // n))
with open("gpurun_out/one_pair_${TAG}.tsv", "w") as out:
    for k, (c, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        line = "%-92s n/call=%6.1f us/call=%9.1f avg_us=%8.2f" % (k, c / n, us / n, us / c)
        out.write(line + "\n")
for l in open("gpurun_out/one_pair_${TAG}.tsv").read().split("\n")[:40]: print(l)
# ordered timeline of the LAST complete call: start offset, duration, gap since the previous kernel on the same queue ended, queue
last = rows[starts[-2]:starts[-1]]
t0 = int(last[0]["Start_Timestamp"])
qend, qn = {}, {}
with open("gpurun_out/one_pair_${TAG}_timeline.tsv", "w") as out:
    out.write("start_us\tdur_us\tgap_same_queue_us\tqueue\tkernel\n")
    for r in last:
        s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")
        qi = qn.setdefault(q, len(qn))
        gap = (s - qend[q]) / 1e3 if q in qend else 0.0
        qend[q] = e
        out.write("%.1f\t%.1f\t%.1f\t%d\t%s\n" % ((s - t0) / 1e3, (e - s) / 1e3, gap, qi, r["Kernel_Name"].split("(")[0].replace("void nps::", "").replace("nps::", "")[:80]))
PY
