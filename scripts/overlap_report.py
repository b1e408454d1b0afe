"""What bounds the four-in-flight loop?  Reads a rocprofv3 --kernel-trace CSV of `bench.py` (steady steps, four batches in flight) and the
rocm-smi samples taken meanwhile, and writes ONE JSON summary (profiles/r6_overlap.json):

  * wall time and sum of kernel time over N steady steps, the share of the wall with 0 / 1 / 2 / >= 3 kernels running;
  * per kernel: launches per step, mean duration in the loop, share of the wall it is running, share of the wall it runs ALONE, and the mean
    number of OTHER kernels running next to it (time-weighted);
  * per kernel: workgroup size, grid, static + dynamic LDS bytes per workgroup, VGPRs (from the trace's own columns) -> what can co-reside;
  * mean power and engine clock over the same run.

usage: python scripts/overlap_report.py <kernel_trace.csv> <smi_samples.txt> <out.json> [steps]"""
import collections
import csv
import json
import re
import sys


def short(name):
    return name.split("(")[0].replace("void nps::", "").replace("nps::", "")[:80]


def main():
    trace, smi, out = sys.argv[1:4]
    nsteps = int(sys.argv[4]) if len(sys.argv) > 4 else 40
    rows = list(csv.DictReader(open(trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    first = [i for i, r in enumerate(rows) if "stem_fused_kernel" in r["Kernel_Name"]]
    # the last 3 stem launches belong to instrumented single steps; take nsteps steps in front of them, away from the warm-up
    hi = first[-4]
    lo = first[-4 - nsteps]
    t0, t1 = int(rows[lo]["Start_Timestamp"]), int(rows[hi]["Start_Timestamp"])
    sel = [r for r in rows if int(r["End_Timestamp"]) > t0 and int(r["Start_Timestamp"]) < t1]
    ev = []
    for i, r in enumerate(sel):
        s, e = max(int(r["Start_Timestamp"]), t0), min(int(r["End_Timestamp"]), t1)
        ev.append((s, 1, i))
        ev.append((e, 0, i))
    ev.sort()
    active = set()
    by_count = collections.Counter()
    run = collections.Counter()
    alone = collections.Counter()
    others = collections.Counter()
    prev = t0
    for t, kind, i in ev:
        dt = t - prev
        if dt > 0:
            n = len(active)
            by_count[min(n, 4)] += dt
            for j in active:
                k = short(sel[j]["Kernel_Name"])
                run[k] += dt
                others[k] += dt * (n - 1)
                if n == 1:
                    alone[k] += dt
        prev = t
        if kind:
            active.add(i)
        else:
            active.discard(i)
    wall = t1 - t0
    dur = collections.Counter()
    cnt = collections.Counter()
    meta = {}
    for r in sel:
        k = short(r["Kernel_Name"])
        dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        cnt[k] += 1
        if k not in meta:
            g = lambda a: int(r.get(a, 0) or 0)
            wg = g("Workgroup_Size_X") * max(g("Workgroup_Size_Y"), 1) * max(g("Workgroup_Size_Z"), 1) or g("Workgroup_Size")
            grid = g("Grid_Size_X") * max(g("Grid_Size_Y"), 1) * max(g("Grid_Size_Z"), 1) or g("Grid_Size")
            meta[k] = {"workgroup_size": wg, "workgroups": grid // wg if wg else None, "lds_bytes_per_workgroup": g("LDS_Block_Size"),
                       "vgpr": g("VGPR_Count"), "accum_vgpr": g("Accum_VGPR_Count"), "sgpr": g("SGPR_Count"), "scratch_bytes": g("Scratch_Size")}
    pw, ck = [], []
    for line in open(smi):
        m = re.findall(r"Power \(W\):\s*([0-9.]+)", line)
        c = re.findall(r"sclk clock level:.*?\((\d+)Mhz\)", line)
        pw += [float(x) for x in m]
        ck += [int(x) for x in c]
    # samples while the benchmark loop runs = those above half of the maximum (idle ~250 W)
    thr = 0.5 * max(pw) if pw else 0
    busy = [i for i, p in enumerate(pw) if p > thr]
    kernels = {}
    for k in sorted(dur, key=lambda k: -dur[k]):
        kernels[k] = {"launches_per_step": round(cnt[k] / nsteps, 2), "mean_us_in_loop": round(dur[k] / cnt[k] / 1e3, 2),
                      "us_per_step": round(dur[k] / nsteps / 1e3, 1), "share_of_wall_running": round(run[k] / wall, 4),
                      "share_of_wall_alone": round(alone[k] / wall, 4), "mean_other_kernels_next_to_it": round(others[k] / run[k], 2) if run[k] else None,
                      **meta[k]}
    res = {"steps": nsteps, "wall_ms_per_step": round(wall / nsteps / 1e6, 4), "sum_kernel_ms_per_step": round(sum(dur.values()) / nsteps / 1e6, 4),
           "share_of_wall_by_kernels_running": {("%d" % n if n < 4 else ">=4"): round(by_count[n] / wall, 4) for n in range(5)},
           "mean_kernels_running": round(sum(run.values()) / wall, 3),
           "power_w": {"mean_under_load": round(sum(pw[i] for i in busy) / len(busy), 1) if busy else None, "max": max(pw) if pw else None, "samples": len(busy)},
           "sclk_mhz": {"mean_under_load": round(sum(ck[i] for i in busy if i < len(ck)) / max(1, len([i for i in busy if i < len(ck)])), 0) if ck else None},
           "kernels_over_100us": {k: v for k, v in kernels.items() if v["mean_us_in_loop"] >= 100},
           "kernels": kernels}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: res[k] for k in ("wall_ms_per_step", "sum_kernel_ms_per_step", "share_of_wall_by_kernels_running", "mean_kernels_running", "power_w", "sclk_mhz")}))
    for k, v in list(kernels.items())[:30]:
        print("%-70s n=%5.1f us=%7.1f run=%.3f alone=%.3f others=%s lds=%s wg=%s" % (k[:70], v["launches_per_step"], v["mean_us_in_loop"], v["share_of_wall_running"],
                                                                                   v["share_of_wall_alone"], v["mean_other_kernels_next_to_it"], v["lds_bytes_per_workgroup"], v["workgroups"]))


if __name__ == "__main__":
    main()
