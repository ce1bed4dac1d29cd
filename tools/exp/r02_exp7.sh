#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for lib in jxl_rs_amd/libjxl_hip.so $(ls jxl_rs_amd/variants/*.so); do
  JXLH_LIBRARY=$PWD/$lib timeout 300 python bench.py --size 16384 --mix all --steps 5 --warmup 1 --no-cpu --no-e2e --no-active --inflight 1 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['roofline']['all_kernels_ms_per_step']
print('$lib'.split('/')[-1], 'step', d['ms_per_step'], 'k1', k['k1_vardct']['ms_per_step'], 'fused', k['k23_fused_filters']['ms_per_step'])
"
done 2>&1 | tee gpurun_out/r02_exp7.txt
