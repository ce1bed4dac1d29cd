#!/bin/bash
# rocprofv3 kernel trace of one bench configuration -> gpurun_out/<tag>_trace.txt   usage: tools/trace16k.sh <tag> [bench args]
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu --no-e2e --no-active --no-secondary --reps 1 --inflight 1 "$@" > $OUT/log.txt 2>&1
DB=$(find $OUT -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $DB $GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace.txt --title "bench.py $*"
head -40 $GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace.txt
