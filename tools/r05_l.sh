#!/bin/bash
O=gpurun_out/r05l; mkdir -p $O
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --strong-at-1 --steps 5 --warmup 2 --no-cpu --no-e2e --no-secondary --no-strip --no-active > $O/strong1.json 2> $O/strong1.err
tail -3 $O/strong1.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05l/strong1.json'))
print('weak', d['value'], d['ms_per_step']); print('strong', json.dumps(d['strong_scaling'])[:500]); print('strong_modular', json.dumps(d['strong_scaling_modular'])[:400])
PY
python bench.py --no-cpu --no-secondary --no-e2e --reps 3 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05l/bench.json'))
print(d['value'], d['ms_per_step'], d['copy_ceiling_GBs'], d['chain_frac_of_copy_ceiling'])
r=d['roofline']
print({k:v['ms_per_step'] for k,v in r['all_kernels_ms_per_step'].items()})
for k in ('epf_population_all_active','epf_population_half_active'):
    print(k, {a:(b['ms_per_step'] if isinstance(b,dict) else b) for a,b in r[k].items()})
s=r['strip_kernel']
print({k:{kk:(vv.get('ms_per_frame') if isinstance(vv,dict) else vv) for kk,vv in v.items()} for k,v in s.items() if isinstance(v,dict)})
PY
