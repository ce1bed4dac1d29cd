#!/bin/bash
O=gpurun_out/r05o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "type or special or all or fuzz or afv" 2>&1 | tail -2
python bench.py --no-cpu --no-e2e --no-active --no-strip --reps 1 --steps 10 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
txt=open('gpurun_out/r05o/bench.json').read()
d=json.loads([l for l in txt.splitlines() if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['copy_ceiling_GBs'])
s=d['secondary']
for k,v in s.items():
    if 'error' in v: print(k, v); continue
    if 'kernels' in v:
        print(k, v.get('ms_per_step'), {a:b['ms_per_step'] for a,b in v['kernels'].items()}, 'epf3' , json.dumps(v.get('epf_iters_3'))[:300] if v.get('epf_iters_3') else '')
    else:
        print(k, json.dumps({a:(b if not isinstance(b,dict) else {x:b[x] for x in b if x in ('ms','frac','wall_ms')}) for a,b in v.items() if a in ('chain','palette','rct_alone')}))
PY
