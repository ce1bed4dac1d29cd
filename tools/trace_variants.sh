#!/bin/bash
# kernel-trace averages of selected kernels for the product library and every variant (JXLH_LIBRARY)
#   usage: tools/trace_variants.sh "<kernel name substrings, |-separated>" [bench args]
PAT=$1; shift
cd /tmp && export TMPDIR=/tmp
for lib in $GRAFT_REPO_ROOT/jxl_rs_amd/libjxl_hip.so $(ls $GRAFT_REPO_ROOT/jxl_rs_amd/variants/*.so 2>/dev/null); do
  OUT=/tmp/trv_$(basename $lib .so); rm -rf $OUT; mkdir -p $OUT
  JXLH_LIBRARY=$lib timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu --no-e2e --no-active --no-secondary --inflight 1 "$@" > $OUT/log.txt 2>&1
  DB=$(find $OUT -name "*.db" | head -1)
  echo "== $(basename $lib)"
  python - "$DB" "$PAT" <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
for name, calls, total, avg, pct in c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
    if any(p in name for p in sys.argv[2].split("|")):
        print(f"  {avg:10.1f} us  x{calls}  {name[:110]}")
PY
done
