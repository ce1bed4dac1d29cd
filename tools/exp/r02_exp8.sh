#!/bin/bash
cd "$(dirname "$0")/../.."
for lib in jxl_rs_amd/libjxl_hip.so $(ls jxl_rs_amd/variants/*.so); do
  echo "== $lib"
  JXLH_LIBRARY=$PWD/$lib timeout 300 python tools/bench_modular.py 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())['kernels']
print({k:v['ms'] for k,v in d.items()})
"
done 2>&1 | tee gpurun_out/r02_exp8.txt
