timeout 800 python -m pytest tests -m gpu -q --deselect tests/test_gpu_fullsize.py 2>&1 | tail -4
for v in base fs40d40 fs32d32 fs48d24 fs24d24; do
  if [ $v = base ]; then lib=$PWD/jxl_rs_amd/libjxl_hip.so; else lib=$PWD/jxl_rs_amd/variants/libjxl_hip_$v.so; fi
  echo -n "$v "; JXLH_LIBRARY=$lib timeout 200 python tools/filter_pop_time.py 2>&1 | tail -1
done
