export JXLH_STRIP_DEADLINE_S=2
for v in base ab1 ab2 ab4 ab7 ab8 ab16 ab64 ab88 ab32 ab128 ab255; do
  if [ $v = base ]; then lib=$PWD/jxl_rs_amd/libjxl_hip.so; else lib=$PWD/jxl_rs_amd/variants/libjxl_hip_$v.so; fi
  echo -n "$v "; JXLH_LIBRARY=$lib timeout 120 python tools/strip_time.py --reps 1 --steps 10 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_frame']['strip_1'], d['kernels_strip']['k123_strip'])"
done
