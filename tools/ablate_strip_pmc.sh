#!/bin/bash
# dynamic instruction counts of the strip kernel per ablation variant (rocprofv3 --pmc), 8K d1 aligned
cd /tmp && export TMPDIR=/tmp
export JXLH_STRIP_DEADLINE_S=2
R=$GRAFT_REPO_ROOT
for v in base ab1 ab2 ab4 ab7 ab8 ab16 ab64 ab88 ab32 ab128 ab255; do
  if [ $v = base ]; then lib=$R/jxl_rs_amd/libjxl_hip.so; else lib=$R/jxl_rs_amd/variants/libjxl_hip_$v.so; fi
  rm -rf /tmp/pmc_$v
  JXLH_LIBRARY=$lib timeout 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d /tmp/pmc_$v -o p -- python $R/tools/strip_time.py --reps 1 --steps 3 > /dev/null 2>&1
  python3 - $v <<'PY'
import csv, glob, sys
from collections import defaultdict
v = sys.argv[1]
acc = defaultdict(list)
for f in glob.glob(f"/tmp/pmc_{v}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k123_strip" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
print(v, {k: round(sum(x) / len(x) / 1e6, 1) for k, x in sorted(acc.items())})
PY
done
