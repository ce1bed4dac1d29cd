#!/bin/bash
# runs bench.py (kernel times only) for the product library and every variant under jxl_rs_amd/variants
cd "$(dirname "$0")/.."
modes=${MODES:-"spec passthrough"}
for lib in jxl_rs_amd/libjxl_hip.so $(ls jxl_rs_amd/variants/*.so 2>/dev/null); do
  for m in $modes; do
    JXLH_LIBRARY=$PWD/$lib timeout 200 python bench.py --no-cpu --inflight 1 --epf $m ${BENCH_ARGS} 2>&1 | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); k=d['roofline']['all_kernels_ms_per_step']
    print('$lib'.split('/')[-1], '$m', 'step', d['ms_per_step'], 'k1', k['k1_vardct']['ms_per_step'], 'fused', k.get('k23_fused_filters',{}).get('ms_per_step'))
except Exception as e:
    print('$lib $m FAILED', e)
"
  done
done
