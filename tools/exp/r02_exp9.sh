#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_x
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_x -o t -- python $R/bench.py --steps 10 --warmup 2 --no-cpu --no-e2e --no-active --inflight 1 > $R/gpurun_out/x.json 2> $R/gpurun_out/x.err
python - <<PY
import csv,glob,json
f=glob.glob("$R/gpurun_out/prof_x/**/*kernel_stats.csv", recursive=True)
for row in csv.DictReader(open(f[0])):
    if 'k1_' in row['Name'] or 'k23' in row['Name']: print(row['Calls'], round(float(row['AverageNs'])/1e3,1), row['Name'][:80])
d=json.loads(open("$R/gpurun_out/x.json").read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['roofline']['all_kernels_ms_per_step'].items()})
PY
cd $R; timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "frame or sparse" 2>&1 | tail -2
