#!/usr/bin/env python3
"""PCIe-inclusive frame time of the slot-bucketed transport with NC contexts, each streaming its frames behind a rolling
window of marks (jxlh_ctx_mark / jxlh_ctx_wait_mark): context c submits and enqueues frame i, then waits for ITS frame
i - lag.  usage: e2e_marks_probe.py [NC ...]   env JXLH_PROBE_E12=1: 12-bit packed entries, JXLH_PROBE_SLOTS: slot streams"""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import jxl_rs_amd
from jxl_rs_amd import synth, lib as jl

size = 8192
E12 = os.environ.get("JXLH_PROBE_E12", "0") != "0"
NSLOTS = int(os.environ.get("JXLH_PROBE_SLOTS", "2"))
FLAGS = jl.GROUP_COMPLETE | (jl.GROUP_ENTRIES12 if E12 else 0)
ESZ = 1 if E12 else 2
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
out = {"MB_per_frame": round((tot * ESZ + ng * 3072) / 1e6, 1), "entries12": E12, "slot_streams": NSLOTS}


def make():
    c = jxl_rs_amd.Context(0, n_slots=NSLOTS)
    c.frame_begin(synth.apply_opts(c.default_params(size, size), wl))
    c.set_dequant_tables(wl.tables); c.set_lf_quantized(*wl.lf_q)
    c.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
    return c


ctxs = [make() for _ in range(3)]
pe, pe_a = ctxs[0].alloc_pinned(tot * ESZ)
pc, pc_a = ctxs[0].alloc_pinned(ng * 3072)
pe.view(np.uint8 if E12 else np.uint16)[:tot] = np.concatenate(es)
pc[:] = np.concatenate(cs)
per = (ng + NSLOTS - 1) // NSLOTS


def submit(c):
    for sl in range(NSLOTS):
        g0, g1 = sl * per, min(ng, (sl + 1) * per)
        c.submit_groups_slots(ids[g0:g1], pe_a + int(off[g0]) * ESZ, pc_a + g0 * 3072, ns[3 * g0:3 * g1], None, slot=sl, flags=FLAGS)


for c in ctxs:
    submit(c); c.frame_run(); c.sync()
want = [float(np.asarray(p, dtype=np.float64).sum()) for p in ctxs[0].read_planes()]
frames = 36
for nc in ([int(a) for a in sys.argv[1:]] or (1, 2, 3)):
    for lag in (1, 2):
        res = []
        for rep in range(4):
            marks = [[] for _ in range(nc)]
            t0 = time.perf_counter()
            for i in range(frames):
                k = i % nc
                c = ctxs[k]
                submit(c); c.frame_run()
                marks[k].append(c.mark())
                if len(marks[k]) > lag:
                    c.wait_mark(marks[k][-1 - lag])
            for c in ctxs[:nc]:
                c.sync()
            res.append(round((time.perf_counter() - t0) / frames * 1e3, 3))
        out[f"contexts_{nc}_lag_{lag}"] = res
# uploads serialised by the host: before a context's next submission the OTHER context's upload must have finished
# (jxlh_slot_wait), so the two contexts stay in anti-phase -- one uploads while the other computes -- by construction
for lag in (1, 2):
    res = []
    for rep in range(5):
        marks = [[] for _ in range(2)]
        t0 = time.perf_counter()
        for i in range(frames):
            k = i % 2
            c, o = ctxs[k], ctxs[1 - k]
            for sl in range(NSLOTS):
                o.slot_wait(sl)
            submit(c); c.frame_run()
            marks[k].append(c.mark())
            if len(marks[k]) > lag:
                c.wait_mark(marks[k][-1 - lag])
        for c in ctxs[:2]:
            c.sync()
        res.append(round((time.perf_counter() - t0) / frames * 1e3, 3))
    out[f"contexts_2_serialised_uploads_lag_{lag}"] = res
# the same ordering on the DEVICE (jxlh_slot_after): the host does not block between the contexts' uploads
for lag in (1, 2):
    res = []
    for rep in range(5):
        marks = [[] for _ in range(2)]
        t0 = time.perf_counter()
        for i in range(frames):
            k = i % 2
            c, o = ctxs[k], ctxs[1 - k]
            for sl in range(NSLOTS):
                for so in range(NSLOTS):
                    c.slot_after(sl, o, so)
            submit(c); c.frame_run()
            marks[k].append(c.mark())
            if len(marks[k]) > lag:
                c.wait_mark(marks[k][-1 - lag])
        for c in ctxs[:2]:
            c.sync()
        res.append(round((time.perf_counter() - t0) / frames * 1e3, 3))
    out[f"contexts_2_device_ordered_uploads_lag_{lag}"] = res
got = [float(np.asarray(p, dtype=np.float64).sum()) for p in ctxs[0].read_planes()]
out["planes_identical_to_single_frame"] = got == want
print(json.dumps(out))
