#!/bin/bash
# A/B of environment switches on the default bench line: bash scripts/ab_env.sh "VAR1=1" "VAR2=1 VAR3=1" ...  (first run = no switch)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() { env $1 python bench.py --no-cpu-baseline --no-accuracy 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%-50s %8.1f pairs/s  %.3f ms' % ('$1', d['value'], d['ms_per_step']))"; }
run "NOPE_BASE=1"
for v in "$@"; do run "$v"; done
run "NOPE_BASE=1"
