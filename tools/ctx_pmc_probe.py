#!/usr/bin/env python3
"""Six contexts with the same dense 8192^2 d1 frame: the placement probe's rating of each context's buffers
(jxlh_probe_placement) beside the HIP-event times of the real K1 and filter launches on them.  One context after the
other, N frames each (a kernel trace / PMC pass can be cut into per-context segments by dispatch order)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import jxl_rs_amd
from jxl_rs_amd import synth
size, N = 8192, int(os.environ.get("PROBE_FRAMES", "30"))
NC = int(os.environ.get("PROBE_CONTEXTS", "6"))
TRIALS = int(os.environ.get("PROBE_TRIALS", "1"))
wl = synth.make_vardct(size, size, mix=synth.MIX_D1, seed=3, unique_groups=24, epf_iters=2)
ctxs, ratings = [], []
for _ in range(NC):
    c = jxl_rs_amd.Context(0, n_slots=1)
    if TRIALS > 1 and len(ctxs) % 2 == 1:   # every other context picks its placement among TRIALS candidates
        c.tune_placement(TRIALS)
    c.frame_begin(synth.apply_opts(c.default_params(size, size), wl))
    c.set_dequant_tables(wl.tables); c.set_lf_quantized(*wl.lf_q)
    c.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
    for g in range(wl.coeffs.shape[0]):
        c.submit_group(g, wl.coeffs[g])
    c.slot_wait(0)
    ratings.append(c.probe_placement() if hasattr(c.L, "jxlh_probe_placement") else (0.0, 0.0))
    ctxs.append(c)
for rnd in range(2):
    for i, c in enumerate(ctxs):
        c.kernel_timing_reset(); c.kernel_timing(True)
        for _ in range(N):
            c.frame_run()
        c.sync()
        kt = {k: round(v[0] / N, 4) for k, v in c.kernel_times().items()}
        c.kernel_timing(False)
        print(f"round {rnd} context {i}: k1 {kt.get('k1_vardct')} filters {kt.get('k23_fused_filters')}   probe: k1-like "
              f"{ratings[i][0]:.4f} filter-like {ratings[i][1]:.4f} sum {sum(ratings[i]):.4f}  tuned {c.tune_placement()[1]}", flush=True)
