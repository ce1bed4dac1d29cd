#!/bin/bash
O=gpurun_out/r05j; mkdir -p $O
for v in base fs48 fs40d32 fs32 fs32d24 fs24d24 base; do
  if [ $v = base ]; then lib=$PWD/jxl_rs_amd/libjxl_hip.so; else lib=$PWD/jxl_rs_amd/variants/libjxl_hip_$v.so; fi
  echo -n "$v "; JXLH_LIBRARY=$lib timeout 200 python tools/filter_pop_time.py 2>&1 | tail -1
done | tee $O/filter_thresholds.txt
python tools/e2e_stream_probe.py 1 2 3 4 8 2 3 > $O/stream.json 2> $O/stream.err; cat $O/stream.json
