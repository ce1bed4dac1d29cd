cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout 300 python bench.py --size 16384 --mix all --steps 5 --warmup 1 --no-cpu --no-e2e --no-active --no-secondary --no-strip --reps 3 --inflight 1 2> gpurun_out/large_fused_bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
k=d['roofline']['all_kernels_ms_per_step']
print('step', d['ms_per_step'], d['repetitions']['ms_per_step'], 'k1', k['k1_vardct']['ms_per_step'], 'filters', k['k23_fused_filters']['ms_per_step'])
"; }
run A=1
run JXLH_K1_SIDE=1
run JXLH_K1_SIDE=1 JXLH_LARGE_GRID=256
run JXLH_K1_SIDE=1 JXLH_LARGE_GRID=384
JXLH_K1_SIDE=1 JXLH_LARGE_GRID=256 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "large or MIX_ALL" 2>&1 | tail -2
