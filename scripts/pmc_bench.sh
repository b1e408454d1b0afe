#!/bin/bash
# HBM traffic of the conv kernels inside the real benchmark (one batch in flight, so per-dispatch counters are clean).
# Separate passes for FETCH_SIZE and WRITE_SIZE (TCC slot limits); counters only (no trace domains).
R=$PWD; export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_bench/$c -o pmc -- python $R/bench.py --steps 2 --warmup 1 --inflight 1 --no-autotune --no-cpu-baseline --no-accuracy > $R/gpurun_out/pmc_bench_$c.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
tot=collections.defaultdict(lambda: collections.defaultdict(float)); calls=collections.Counter()
for f in glob.glob("gpurun_out/pmc_bench/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0][-70:]
        tot[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        if r["Counter_Name"]=="FETCH_SIZE": calls[k]+=1
steps=4   # warmup 1 + 2 timed + 1 instrumented
rows=[]
for k,v in tot.items():
    # FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 (MI355X_MICROARCH.md)
    rd=2*v.get("FETCH_SIZE",0)*1024/steps; wr=v.get("WRITE_SIZE",0)*1024/steps
    rows.append((rd+wr, k, rd, wr, calls[k]/steps))
rows.sort(reverse=True)
print("per step: kernel, launches, HBM read GB (2x FETCH_SIZE), write GB")
for t,k,rd,wr,n in rows[:14]:
    print("%-72s %6.1f  %7.3f  %7.3f"%(k,n,rd/1e9,wr/1e9))
conv=[r for r in rows if "conv_igemm" in r[1]]
print("conv kernels total per step: read %.2f GB write %.2f GB"%(sum(r[2] for r in conv)/1e9, sum(r[3] for r in conv)/1e9))
PY
