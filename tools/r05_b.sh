#!/bin/bash
# round 5, session b: stream priority classes A/B + kernel trace of the sparse-resident frame
O=gpurun_out/r05b; mkdir -p $O
python tools/r05_inflight_probe.py > $O/inflight_prio.json 2> $O/inflight_prio.err
JXLH_STREAM_PRIORITY=0,0 python tools/r05_inflight_probe.py > $O/inflight_flat.json 2> $O/inflight_flat.err
python bench.py --no-cpu --no-strip --no-secondary --no-active --reps 3 > $O/bench_prio.json 2> $O/bench_prio.err
JXLH_STREAM_PRIORITY=0,0 python bench.py --no-cpu --no-strip --no-secondary --no-active --reps 3 > $O/bench_flat.json 2> $O/bench_flat.err
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_sparse -o t -- python $GRAFT_REPO_ROOT/tools/sparse_resident.py > $GRAFT_REPO_ROOT/$O/trace_sparse.log 2>&1
cd $GRAFT_REPO_ROOT
find $O -name "*kernel_stats.csv" | head
