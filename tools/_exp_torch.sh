echo "--- probe plain"
python tools/e2e_marks_probe.py 2 2>&1 | tail -1
echo "--- probe with torch.cuda initialised first"
python - <<'PY' 2>&1 | tail -1
import sys, runpy
import torch
print('torch cuda', torch.cuda.is_available(), file=sys.stderr)
torch.cuda.synchronize()
sys.argv = ['e2e_marks_probe.py', '2']
runpy.run_path('tools/e2e_marks_probe.py', run_name='__main__')
PY
