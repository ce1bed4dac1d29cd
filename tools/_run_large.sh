cd $GRAFT_REPO_ROOT
for v in base gabpk base gabpk; do
  if [ $v = base ]; then lib=$PWD/jxl_rs_amd/libjxl_hip.so; else lib=$PWD/jxl_rs_amd/variants/libjxl_hip_$v.so; fi
  echo -n "$v "; JXLH_LIBRARY=$lib timeout 200 python tools/filter_pop_time.py 2>&1 | tail -1
done
JXLH_LIBRARY=$PWD/jxl_rs_amd/variants/libjxl_hip_gabpk.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -2
