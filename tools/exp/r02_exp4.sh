#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02_exp4_tests.txt 2>&1
grep -E "passed|failed" gpurun_out/r02_exp4_tests.txt | tail -2
MODES="spec" BENCH_ARGS="--no-e2e --steps 20" bash tools/bench_variants.sh 2>&1 | tee gpurun_out/r02_exp4_variants.txt
MODES="spec" BENCH_ARGS="--no-e2e --steps 20" bash tools/bench_variants.sh 2>&1 | tee -a gpurun_out/r02_exp4_variants.txt
