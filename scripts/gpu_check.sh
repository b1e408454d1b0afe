#!/bin/bash
# Run on the GPU box (via gpurun): GPU test-suite, the default bench line, and a rocprofv3 kernel-trace of
# the same bench command.  Usage: bash scripts/gpu_check.sh <tag> [skip-tests]
set -u
TAG=${1:-r1}
R=$PWD
mkdir -p gpurun_out
if [ "${2:-}" != "skip-tests" ]; then
  python -m pytest tests -x -q -m gpu 2>&1 | tail -15
fi
python bench.py --stages --layers gpurun_out/layers_${TAG}.tsv 2>&1 | tail -1 > gpurun_out/bench_${TAG}.json
cat gpurun_out/bench_${TAG}.json
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG} -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-accuracy > $R/gpurun_out/prof_${TAG}.log 2>&1
cd $R
find gpurun_out/prof_${TAG} -name "*kernel_stats*" | head
