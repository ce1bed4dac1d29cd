#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + PMC passes for bench.py.
# usage: tools/gpu_profile.sh <tag> [bench args...]
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH=${PROFILE_CMD:-"python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu --no-e2e --no-active --no-secondary --reps 1 --inflight 1 $*"}
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $BENCH > $OUT/trace.log 2>&1
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES" \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc$i -o p -- $BENCH > $OUT/pmc$i.log 2>&1
done
find $OUT -name "*.csv" | head -40
