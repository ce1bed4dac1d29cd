#!/bin/bash
O=gpurun_out/r05f; mkdir -p $O
for v in base nodq; do
  if [ $v = base ]; then lib=$PWD/jxl_rs_amd/libjxl_hip.so; else lib=$PWD/jxl_rs_amd/variants/libjxl_hip_$v.so; fi
  echo "== $v"; JXLH_LIBRARY=$lib python tools/r05_inflight_probe.py 8192 slots 2> $O/$v.err | tee $O/$v.json | tr -d '\n ' ; echo
done
