python tools/e2e_marks_probe.py 2 2>&1 | tail -1
JXLH_BENCH_SLOTS=2 python bench.py --no-cpu --no-secondary --no-strip --no-active --reps 1 --steps 10 > /tmp/b.json 2>/dev/null
python - <<'PY'
import json
txt=open('/tmp/b.json').read()
d=json.loads([l for l in txt.splitlines() if l.startswith('{')][-1])
e=d['e2e_pcie_inclusive']
print('bench slots=2', d['ms_per_step'], {k:(v.get('ms_per_frame'), v.get('ms_per_frame_sync_loop'), v.get('repetitions_ms')) for k,v in e.items() if k.startswith('slots')})
PY
python tools/e2e_marks_probe.py 2 2>&1 | tail -1
