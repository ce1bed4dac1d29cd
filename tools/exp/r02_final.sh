#!/bin/bash
# round-2 measurement set: tests, default bench line, rocprofv3 kernel trace + PMC of the same command, secondary configs
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
TAG=${1:-r02_h}
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_tests.txt 2>&1; grep -E "passed|failed|error" gpurun_out/${TAG}_tests.txt | tail -2
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
bash tools/gpu_profile.sh ${TAG} > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/prof_${TAG} gpurun_out/${TAG}_pmc.txt gpurun_out/${TAG}_traffic.json
for cfg in "c2:--size 4096 --epf-iters 0" "c5:--size 16384 --mix all" "c5e3:--size 16384 --mix all --epf-iters 3" "c3i1:--inflight 1"; do
  name=${cfg%%:*}; args=${cfg#*:}
  timeout 600 python bench.py $args --no-cpu --no-e2e --no-active > gpurun_out/${TAG}_cfg_$name.json 2> gpurun_out/${TAG}_cfg_$name.err
done
timeout 600 python tools/bench_modular.py > gpurun_out/${TAG}_modular.json 2> gpurun_out/${TAG}_modular.err
timeout 300 python tools/bench_delta_palette.py 1024x1024 8192x8192 > gpurun_out/${TAG}_delta_palette.txt 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/${TAG}_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    if "roofline" in d and d["roofline"]:
        print(f, d["value"], d["ms_per_step"], {k:v["ms_per_step"] for k,v in d["roofline"]["all_kernels_ms_per_step"].items()})
    elif "kernels" in d:
        print(f, {k:v["ms"] for k,v in d["kernels"].items()})
PY
