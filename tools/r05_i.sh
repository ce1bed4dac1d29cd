#!/bin/bash
O=gpurun_out/r05i; mkdir -p $O
for v in base fs48 fs40d32 fs32 fs32d24 fs16d16 base; do
  if [ $v = base ]; then lib=$PWD/jxl_rs_amd/libjxl_hip.so; else lib=$PWD/jxl_rs_amd/variants/libjxl_hip_$v.so; fi
  echo -n "$v "; JXLH_LIBRARY=$lib timeout 200 python tools/filter_pop_time.py 2>&1 | tail -1
done | tee $O/filter_thresholds.txt
python bench.py --no-cpu --no-strip --no-secondary --reps 3 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05i/bench.json'))
print(d['value'], d['ms_per_step'])
e=d['e2e_pcie_inclusive']
for k,v in e.items():
    if k!='note': print(k, v.get('ms_per_frame'), v.get('value'))
PY
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
