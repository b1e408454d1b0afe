#!/bin/bash
# fp8 backbone mode: kernel + stage tests, then the bench with and without --fp8 (same box, back to back)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py tests/test_stages_gpu.py -x -q -m gpu -k "fp8" 2>&1 | tail -15
python bench.py --no-cpu-baseline --fp8 --layers gpurun_out/fp8_layers.tsv > gpurun_out/fp8_bench.json 2> gpurun_out/fp8_bench.err; tail -3 gpurun_out/fp8_bench.err; cat gpurun_out/fp8_bench.json
python bench.py --no-cpu-baseline --no-accuracy > gpurun_out/bf16_bench.json 2>/dev/null; cat gpurun_out/bf16_bench.json
