#!/bin/bash
O=gpurun_out/r05m; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_progressive.py -x -q -m gpu 2>&1 | tail -3
python bench.py --no-cpu --no-secondary --no-strip --no-active --reps 3 > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python - <<'PY'
import json
txt=open('gpurun_out/r05m/bench.json').read()
d=json.loads([l for l in txt.splitlines() if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['copy_ceiling_GBs'])
e=d['e2e_pcie_inclusive']
for k,v in e.items():
    if k!='note': print(k, json.dumps({a:b for a,b in v.items() if a in ('ms_per_frame','value','repetitions_ms','ms_per_frame_sync_loop','kernels_ms')}))
PY
