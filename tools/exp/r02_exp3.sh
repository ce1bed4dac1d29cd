#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for m in spec active passthrough; do
  timeout 300 python bench.py --no-cpu --no-e2e --inflight 1 --epf $m --steps 20 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['roofline']['all_kernels_ms_per_step']
print('$m inflight1 step', d['ms_per_step'], {n:v['ms_per_step'] for n,v in k.items()})
"
done
timeout 300 python bench.py --no-cpu --no-e2e --steps 20 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['roofline']['all_kernels_ms_per_step']
print('spec inflight2 step', d['ms_per_step'], {n:v['ms_per_step'] for n,v in k.items()})
"
