#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02_e16k -o t -- python $R/bench.py --size 16384 --mix all --steps 5 --warmup 1 --no-cpu --no-e2e --no-active --inflight 1 > $R/gpurun_out/r02_e16k.json 2> $R/gpurun_out/r02_e16k.err
python - <<PY
import csv,glob,json
f=glob.glob("$R/gpurun_out/prof_r02_e16k/**/*kernel_stats.csv", recursive=True)
for row in csv.DictReader(open(f[0])):
    print(row['Calls'], round(float(row['AverageNs'])/1e3,1), row['Percentage'], row['Name'][:90])
d=json.loads(open("$R/gpurun_out/r02_e16k.json").read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['roofline']['all_kernels_ms_per_step'].items()})
PY
cd $R
timeout 600 python tools/bench_modular.py > gpurun_out/r02_e_modular.json 2>gpurun_out/r02_e_modular.err; cat gpurun_out/r02_e_modular.json
