#!/usr/bin/env python3
"""Slot-resident 8K d1 frames, TWO contexts round robin (the pattern of bench.py's slots_resident_no_pcie leg): ms per frame,
median of 5 repetitions of 20 frames.  For A/B runs of library variants (JXLH_LIBRARY)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import jxl_rs_amd
from jxl_rs_amd import synth
size = 8192
wl = synth.make_vardct(size, size, mix=synth.MIX_D1, seed=3, unique_groups=24, epf_iters=2)
ng = wl.coeffs.shape[0]
cache, e, cn, ns = {}, [], [], []
for g in range(ng):
    k = g % 24
    if k not in cache:
        cache[k] = synth.to_slots(wl.coeffs[g])
    e.append(cache[k][0]); cn.append(cache[k][1].reshape(-1)); ns.append(cache[k][2])
e, cn, ns = np.concatenate(e), np.concatenate(cn), np.concatenate(ns)
ctxs = [jxl_rs_amd.Context(0, n_slots=1) for _ in range(2)]
for c in ctxs:
    c.frame_begin(synth.apply_opts(c.default_params(size, size), wl))
    c.set_dequant_tables(wl.tables); c.set_lf_quantized(*wl.lf_q)
    c.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
    c.submit_groups_slots(np.arange(ng, dtype=np.uint32), e, cn, ns, None)
    c.slot_wait(0)
    c.frame_run()
for c in ctxs:
    c.sync()
t = time.perf_counter()
while time.perf_counter() - t < 0.1:
    for i in range(10):
        ctxs[i % 2].frame_run()
    for c in ctxs:
        c.sync()
reps = []
for _ in range(5):
    t0 = time.perf_counter()
    for i in range(20):
        ctxs[i % 2].frame_run()
    for c in ctxs:
        c.sync()
    reps.append((time.perf_counter() - t0) * 1e3 / 20)
print("two contexts, slot-resident ms/frame:", [round(v, 4) for v in sorted(reps)], "median", round(sorted(reps)[2], 4))
