#!/bin/bash
O=gpurun_out/r05n; mkdir -p $O
for s in 1 2 1; do
JXLH_BENCH_SLOTS=$s python bench.py --no-cpu --no-secondary --no-strip --no-active --reps 1 --steps 10 > $O/bench_s$s.json 2> $O/bench.err
python - <<PY
import json
txt=open('gpurun_out/r05n/bench_s$s.json').read()
d=json.loads([l for l in txt.splitlines() if l.startswith('{')][-1])
e=d['e2e_pcie_inclusive']
print('slots=$s', {k:(v.get('ms_per_frame'), v.get('ms_per_frame_sync_loop')) for k,v in e.items() if k!='note'})
PY
done
