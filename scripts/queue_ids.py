"""(Queue_Id, Stream_Id) pairs of a rocprofv3 kernel trace and what runs where (main = the stem's stream, side = the pose net's).
usage: queue_ids.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = rows[len(rows) // 2:]
kind = {}
for r in sel:
    k = (r["Queue_Id"], r["Stream_Id"])
    n = r["Kernel_Name"]
    if "stem_fused" in n:
        kind[k] = "main"
    elif "gn_stats" in n and kind.get(k) != "main":
        kind.setdefault(k, "side")
cnt = collections.Counter((r["Queue_Id"], r["Stream_Id"]) for r in sel)
byq = collections.defaultdict(list)
for (q, s), c in sorted(cnt.items()):
    byq[q].append("%s%s(%d)" % (kind.get((q, s), "?"), s, c))
for q in sorted(byq):
    print("queue %s: %s" % (q, "  ".join(byq[q])))
