#!/usr/bin/env python3
"""PCIe-inclusive frame time of the slot-bucketed transport when ONE context streams consecutive frames: frame i + 1 is
submitted while frame i runs (the upload goes under frame i's kernels), the host syncs every `depth` frames only."""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import jxl_rs_amd
from jxl_rs_amd import synth

size = 8192
E12 = os.environ.get("JXLH_PROBE_E12", "0") != "0"   # 12-bit entries packed two per three bytes (JXLH_GROUP_ENTRIES12)
from jxl_rs_amd import lib as _jl
FLAGS = _jl.GROUP_COMPLETE | (_jl.GROUP_ENTRIES12 if E12 else 0)
ESZ = 1 if E12 else 2   # bytes per element of the entries array (bytes for the packed form)
wl = synth.make_vardct(size, size, mix=synth.MIX_D1, seed=3, unique_groups=24, epf_iters=2)
ng = wl.coeffs.shape[0]
cache, es, cs, ns = {}, [], [], []
for g in range(ng):
    k = g % 24
    if k not in cache:
        cache[k] = synth.to_slots(wl.coeffs[g], E12)
    es.append(cache[k][0]); cs.append(cache[k][1].reshape(-1)); ns.append(cache[k][2])
off = np.concatenate([[0], np.cumsum([len(x) for x in es])]).astype(np.int64)
tot = int(off[-1])
ns = np.concatenate(ns).astype(np.uint32)
ids = np.arange(ng, dtype=np.uint32)
out = {"MB_per_frame": round((tot * ESZ + ng * 3072) / 1e6, 1), "entries12": E12}
c = jxl_rs_amd.Context(0, n_slots=2)
c.frame_begin(synth.apply_opts(c.default_params(size, size), wl))
c.set_dequant_tables(wl.tables); c.set_lf_quantized(*wl.lf_q)
c.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
pe, pe_a = c.alloc_pinned(tot * ESZ)
pc, pc_a = c.alloc_pinned(ng * 3072)
pe.view(np.uint8 if E12 else np.uint16)[:tot] = np.concatenate(es)
pc[:] = np.concatenate(cs)


def submit(nslots=2):
    per = (ng + nslots - 1) // nslots
    for sl in range(nslots):
        g0, g1 = sl * per, min(ng, (sl + 1) * per)
        c.submit_groups_slots(ids[g0:g1], pe_a + int(off[g0]) * ESZ, pc_a + g0 * 3072, ns[3 * g0:3 * g1], None, slot=sl, flags=FLAGS)


submit(); c.frame_run(); c.sync()
want = [float(np.asarray(p, dtype=np.float64).sum()) for p in c.read_planes()]
for depth in ([int(a) for a in sys.argv[1:]] or (4, 8, 12, 24, 6, 8, 12)):
    for _ in range(4):
        submit(); c.frame_run()
    c.sync()
    t0 = time.perf_counter()
    for i in range(24):
        submit(); c.frame_run()
        if (i + 1) % depth == 0:
            c.sync()
    c.sync()
    out.setdefault(f"ms_per_frame_sync_every_{depth}", []).append(round((time.perf_counter() - t0) / 24 * 1e3, 3))
# rolling window instead of draining syncs (round 5, jxlh_ctx_mark / jxlh_ctx_wait_mark): after frame i + 1 has been
# submitted and enqueued, wait for frame i - lag + 1 only
for lag in (1, 2, 3):
    for _ in range(4):
        submit(); c.frame_run()
    c.sync()
    for rep in range(3):
        marks = []
        t0 = time.perf_counter()
        for i in range(24):
            submit(); c.frame_run()
            marks.append(c.mark())
            if len(marks) > lag:
                c.wait_mark(marks[-1 - lag])
        c.sync()
        out.setdefault(f"ms_per_frame_wait_mark_lag_{lag}", []).append(round((time.perf_counter() - t0) / 24 * 1e3, 3))
got = [float(np.asarray(p, dtype=np.float64).sum()) for p in c.read_planes()]
out["planes_identical_to_single_frame"] = got == want
print(json.dumps(out))
c.close()
if len(sys.argv) > 1:
    sys.exit(0)
# host-side time of each call inside the streaming loop
c = jxl_rs_amd.Context(0, n_slots=2)
c.frame_begin(synth.apply_opts(c.default_params(size, size), wl))
c.set_dequant_tables(wl.tables); c.set_lf_quantized(*wl.lf_q)
c.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
for _ in range(4):
    submit(); c.frame_run()
c.sync()
ts = []
t0 = time.perf_counter()
for i in range(12):
    a = time.perf_counter(); submit(); b = time.perf_counter(); c.frame_run(); d = time.perf_counter()
    ts.append((round((a - t0) * 1e3, 3), round((b - a) * 1e3, 3), round((d - b) * 1e3, 3)))
c.sync()
print("t_start_ms, submit_ms, frame_run_ms:", ts, "total", round((time.perf_counter() - t0) * 1e3, 3))
