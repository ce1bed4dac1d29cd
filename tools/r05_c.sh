#!/bin/bash
# round 5, session c: PMC of the sparse-resident frame + stream priority "slots high"
O=gpurun_out/r05c; mkdir -p $O
JXLH_STREAM_PRIORITY=0,-1 python tools/r05_inflight_probe.py > $O/inflight_slothigh.json 2> $O/inflight_slothigh.err
JXLH_STREAM_PRIORITY=0,-1 python bench.py --no-cpu --no-strip --no-secondary --no-active --reps 3 > $O/bench_slothigh.json 2> $O/bench_slothigh.err
PROFILE_CMD="python $GRAFT_REPO_ROOT/tools/sparse_resident.py" bash tools/gpu_profile.sh r05c_sparse > $O/prof.log 2>&1
python tools/pmc_summary.py gpurun_out/prof_r05c_sparse $O/sparse_pmc.txt
