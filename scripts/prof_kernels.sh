#!/bin/bash
# kernel-trace stats of the default bench command (no counters); prints the top kernels.  Usage: bash scripts/prof_kernels.sh <tag> [bench args]
TAG=${1:-tmp}; shift
R=$PWD; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG} -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-accuracy "$@" > $R/gpurun_out/prof_${TAG}.log 2>&1
cd $R
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_${TAG}/bench_kernel_stats.csv")))
for r in rows[:28]:
    print("%-84s calls=%6s tot_ms=%9.3f avg_us=%8.2f"%(r["Name"].split("(")[0][-84:], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3))
PY
