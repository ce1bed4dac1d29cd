#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/r02_b_bench.json 2> gpurun_out/r02_b_bench.err
echo "rc=$?"; tail -c 600 gpurun_out/r02_b_bench.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu --no-e2e > gpurun_out/r02_b_bench_torchrun1.json 2> gpurun_out/r02_b_bench_torchrun1.err
echo "rc=$?"; tail -c 600 gpurun_out/r02_b_bench_torchrun1.err
python - <<'PY'
import json
for f in ("gpurun_out/r02_b_bench.json","gpurun_out/r02_b_bench_torchrun1.json"):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d["roofline"]
        print(f, d["value"], d["ms_per_step"], r["kernel"], r["frac"], r["copy_ceiling"], r["chain_vs_fused_ideal"], d.get("cpu_baseline"))
        print({k:(v["ms_per_step"],v.get("frac")) for k,v in r["all_kernels_ms_per_step"].items()})
        print("active", r["epf_population_all_active"])
    except Exception as e:
        print(f, "FAILED", e)
PY
