#!/bin/bash
# Soak of the VarDCT frame path on a GPU box: the randomised differential tests (tests/test_gpu_fuzz.py) re-run with
# shifted seeds, then with tripled frame sizes.   usage: tools/soak_vardct.sh [rounds]   -> gpurun_out/soak_vardct.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
R=${1:-8}
: > gpurun_out/soak_vardct.txt
for i in $(seq 1 $R); do
  JXLH_FUZZ_OFFSET=$((i * 100000)) timeout 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -2 | sed "s/^/offset $((i * 100000)): /" >> gpurun_out/soak_vardct.txt
done
for i in 1 2; do
  JXLH_FUZZ_SCALE=3 JXLH_FUZZ_OFFSET=$((i * 7000)) timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -2 | sed "s/^/scale 3 offset $((i * 7000)): /" >> gpurun_out/soak_vardct.txt
done
cat gpurun_out/soak_vardct.txt
