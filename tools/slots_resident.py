#!/usr/bin/env python3
"""Frame submitted once in the slot-bucketed form (jxlh_submit_groups_slots), then re-run N times: the transforms read
the entries in place.  For kernel traces / counters of the entries form."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import jxl_rs_amd
from jxl_rs_amd import synth
size = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
wl = synth.make_vardct(size, size, mix=synth.MIX_D1, seed=3, unique_groups=24, epf_iters=2)
ng = wl.coeffs.shape[0]
c = jxl_rs_amd.Context(0, n_slots=1)
c.frame_begin(synth.apply_opts(c.default_params(size, size), wl))
c.set_dequant_tables(wl.tables); c.set_lf_quantized(*wl.lf_q)
c.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
cache, e, cn, ns = {}, [], [], []
for g in range(ng):
    k = g % 24
    if k not in cache:
        cache[k] = synth.to_slots(wl.coeffs[g])
    e.append(cache[k][0]); cn.append(cache[k][1].reshape(-1)); ns.append(cache[k][2])
c.submit_groups_slots(np.arange(ng, dtype=np.uint32), np.concatenate(e), np.concatenate(cn), np.concatenate(ns), None)
c.slot_wait(0)
for _ in range(3):
    c.frame_run()
c.sync()
N = 10
t0 = time.perf_counter()
for _ in range(N):
    c.frame_run()
c.sync()
print("slot-resident ms/frame:", (time.perf_counter() - t0) / N * 1e3)
c.kernel_timing(True)
for _ in range(N):
    c.frame_run()
c.sync()
print({k: round(v[0] / N, 4) for k, v in c.kernel_times().items()})
