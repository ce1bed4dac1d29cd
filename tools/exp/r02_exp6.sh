#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -k "transform or 16k or all_types or frame" > gpurun_out/exp6_tests.txt 2>&1; grep -E "passed|failed" gpurun_out/exp6_tests.txt | tail -2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_r02_f16k
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02_f16k -o t -- python $R/bench.py --size 16384 --mix all --steps 5 --warmup 1 --no-cpu --no-e2e --no-active --inflight 1 > $R/gpurun_out/r02_f16k.json 2> $R/gpurun_out/r02_f16k.err
python - <<PY
import csv,glob,json
f=glob.glob("$R/gpurun_out/prof_r02_f16k/**/*kernel_stats.csv", recursive=True)
for row in csv.DictReader(open(f[0])):
    if 'k1_' in row['Name'] or 'k23' in row['Name']: print(row['Calls'], round(float(row['AverageNs'])/1e3,1), row['Percentage'], row['Name'][:90])
d=json.loads(open("$R/gpurun_out/r02_f16k.json").read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['roofline']['all_kernels_ms_per_step'].items()})
PY
