#!/bin/bash
# HBM traffic of the conv kernels inside the real benchmark (one batch in flight, so per-dispatch counters are clean).
# Separate passes for FETCH_SIZE and WRITE_SIZE (TCC slot limits); counters only (no trace domains).
# Every path argument is made absolute BEFORE the cd (round 4 passed a relative --routing: bench.py did not find it under /tmp,
# re-tuned under the profiler and the PMC pass counted a different kernel mix than the benchmark line).
R=$PWD; export TMPDIR=/tmp
ARGS=(); ROUTING=""
while [ $# -gt 0 ]; do
  if [ "$1" = "--routing" ]; then ROUTING=$(realpath "$2"); ARGS+=(--routing "$ROUTING"); shift 2; else ARGS+=("$1"); shift; fi
done
if [ -z "$ROUTING" ]; then ROUTING=$R/profiles/routing_r5.json; ARGS+=(--routing "$ROUTING"); fi
[ -f "$ROUTING" ] || { echo "pmc_bench.sh: routing file $ROUTING does not exist" >&2; exit 2; }
export PMC_ROUTING_FILE=$ROUTING
set -- "${ARGS[@]}"
cd /tmp
[ -n "$PMC_SKIP_COLLECT" ] && set --          # (re-summarise the CSVs of an earlier collection)
for c in $([ -n "$PMC_SKIP_COLLECT" ] || echo FETCH_SIZE WRITE_SIZE); do
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_bench/$c -o pmc -- python $R/bench.py --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline --no-accuracy --no-fp32-path --no-boundary --no-other-configs --no-tape "$@" > $R/gpurun_out/pmc_bench_$c.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections, json
tot=collections.defaultdict(lambda: collections.defaultdict(float)); calls=collections.Counter()
for f in glob.glob("gpurun_out/pmc_bench/*/*counter_collection.csv"):
    rows=sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
    # only the 6 real forward passes at the end (each starts with the preprocess kernel): the load-time autotuner's trial
    # launches before them are not part of a step
    starts=[int(r["Dispatch_Id"]) for r in rows if (("stem_fused_kernel<1" in r["Kernel_Name"] or "stem_fused_kernel<2" in r["Kernel_Name"]) or "preprocess" in r["Kernel_Name"]) and r["Counter_Name"]==rows[0]["Counter_Name"]]
    first=sorted(set(starts))[-6]          # warmup 1 + 2 timed + 3 instrumented forward passes
    for r in rows:
        if int(r["Dispatch_Id"]) < first: continue
        k=r["Kernel_Name"].split("(")[0].replace("void ","").strip()[-80:]
        tot[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        if r["Counter_Name"]=="FETCH_SIZE": calls[k]+=1
steps=6   # warmup 1 + 2 timed + 3 instrumented forward passes
rows=[]
for k,v in tot.items():
    # FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 (MI355X_MICROARCH.md)
    rd=2*v.get("FETCH_SIZE",0)*1024/steps; wr=v.get("WRITE_SIZE",0)*1024/steps
    rows.append((rd+wr, k, rd, wr, calls[k]/steps))
rows.sort(reverse=True)
print("per step: kernel, launches, HBM read GB (2x FETCH_SIZE), write GB")
for t,k,rd,wr,n in rows[:16]:
    print("%-80s %6.1f  %7.3f  %7.3f"%(k,n,rd/1e9,wr/1e9))
# the family bench.py's per-launch timer sees (`roofline.conv_family`): every bf16 conv2d / bottleneck-tail launch - NOT the fused stem and the
# res2 3x3 kernel, which are separate entry points
fam=[r for r in rows if any(s in r[1] for s in ("conv_igemm", "pw_chain", "conv3x3_halo")) and "float" not in r[1]]
import os
out={"routing_file": os.path.relpath(os.environ["PMC_ROUTING_FILE"]),
     "note": "HBM bytes per forward pass of 32 pairs (bench.py --inflight 1, kernel routing from the routing file, the 6 forward passes after the tuning), rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; "
             "read = 2 x FETCH_SIZE x 1024 (gfx950 correction, MI355X_MICROARCH.md), write = WRITE_SIZE x 1024",
     "bf16_conv_family": {"launches_per_step": sum(r[4] for r in fam), "read_bytes_per_step": sum(r[2] for r in fam), "write_bytes_per_step": sum(r[3] for r in fam)},
     "kernels": {r[1]: {"launches_per_step": r[4], "read_bytes_per_step": r[2], "write_bytes_per_step": r[3]} for r in rows[:40]}}
f=out["bf16_conv_family"]
print("bf16 conv family per step: %.0f launches, read %.2f GB, write %.2f GB" % (f["launches_per_step"], f["read_bytes_per_step"]/1e9, f["write_bytes_per_step"]/1e9))
json.dump(out, open("gpurun_out/pmc_traffic.json","w"), indent=1)
PY
