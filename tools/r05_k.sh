#!/bin/bash
O=gpurun_out/r05k; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; grep -E "passed|failed|error" $O/gpu_tests.txt | tail -3
python bench.py --no-cpu --no-secondary --reps 3 > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05k/bench.json'))
print(d['value'], d['ms_per_step'], d['copy_ceiling_GBs'], d['chain_counter_traffic_bytes'], d['chain_frac_of_copy_ceiling'])
r=d['roofline']
print({k:v['ms_per_step'] for k,v in r['all_kernels_ms_per_step'].items()})
for k in ('epf_population_all_active','epf_population_half_active'):
    print(k, {a:(b['ms_per_step'] if isinstance(b,dict) else b) for a,b in r[k].items()})
print('strip', json.dumps(r.get('strip_kernel'))[:600])
e=d['e2e_pcie_inclusive']
for k,v in e.items():
    if k!='note': print(k, json.dumps(v)[:400])
PY
