#!/bin/bash
O=gpurun_out/r05g; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_progressive.py tests/test_gpu_parity.py -x -q -m gpu -k "slot or sparse or progress or accumul or rerender or pass or entries" > $O/tests.txt 2>&1
tail -15 $O/tests.txt
python tools/r05_inflight_probe.py 8192 slots > $O/inflight.json 2> $O/inflight.err
tr -d '\n ' < $O/inflight.json; echo
JXLH_NO_DIRECT_ENTRIES=1 python tools/r05_inflight_probe.py 8192 slots 2>/dev/null | tr -d '\n '; echo
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o t -- python $GRAFT_REPO_ROOT/tools/r05_inflight_probe.py 8192 slots > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, collections
rows=list(csv.DictReader(open('gpurun_out/r05g/trace/t_kernel_trace.csv')))
d=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name']
    for k in ('k1_scan','k1_dct8','k1_dct16_32','k1_entries_fallback','k23_fused','k0b','k3_sigma'):
        if k in n: d[k].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in d.items():
    v=sorted(v); print(k, len(v), 'min %.1f p25 %.1f med %.1f'%(v[0], v[len(v)//4], v[len(v)//2]))
PY
