#!/bin/bash
# round 2, experiment 1: Infinity Cache probe + band-pipelined frame_run
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
./tools/mall_probe > gpurun_out/mall_probe.txt 2>&1
JXLH_BAND_ROWS=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/band_parity.txt 2>&1
echo "rc=$?" >> gpurun_out/band_parity.txt
for br in 0 1 2 3 4 8; do
  for inf in 1 2; do
    echo "== band_rows=$br inflight=$inf" >> gpurun_out/band_bench.txt
    JXLH_BAND_ROWS=$br timeout 300 python bench.py --no-cpu --no-e2e --inflight $inf --steps 20 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['roofline']['all_kernels_ms_per_step']
print('step', d['ms_per_step'], 'ev', d['hip_event_ms_per_step_rank0'], {n:(v['ms_per_step'],v['launches_per_step']) for n,v in k.items()})
" >> gpurun_out/band_bench.txt 2>&1
  done
done
cat gpurun_out/mall_probe.txt gpurun_out/band_bench.txt; tail -3 gpurun_out/band_parity.txt
