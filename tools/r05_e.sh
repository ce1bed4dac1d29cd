#!/bin/bash
O=gpurun_out/r05e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_progressive.py tests/test_gpu_parity.py -x -q -m gpu -k "slot or sparse or progress or accumul or rerender or pass" > $O/tests.txt 2>&1
tail -3 $O/tests.txt
for v in base wpe2 wpe2pf2 pf0; do
  if [ $v = base ]; then lib=$PWD/jxl_rs_amd/libjxl_hip.so; else lib=$PWD/jxl_rs_amd/variants/libjxl_hip_$v.so; fi
  echo "== $v"; JXLH_LIBRARY=$lib python tools/r05_inflight_probe.py 8192 slots 2> $O/$v.err | tee $O/$v.json | tr -d '\n ' ; echo
done
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o t -- python $GRAFT_REPO_ROOT/tools/r05_inflight_probe.py 8192 slots > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
cd $GRAFT_REPO_ROOT; head -8 $O/trace/t_kernel_stats.csv | cut -c1-150
