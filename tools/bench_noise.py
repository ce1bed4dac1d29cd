#!/usr/bin/env python3
"""Kernel times of the noise synthesis on an 8192^2 VarDCT d1 frame."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jxl_rs_amd
from jxl_rs_amd import synth

size = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
wl = synth.make_vardct(size, size, mix=synth.MIX_D1, seed=3, unique_groups=24, epf_iters=2)
c = jxl_rs_amd.Context(0, n_slots=1)
p = synth.apply_opts(c.default_params(size, size), wl)
p.noise = 1
for i in range(8):
    p.noise_lut[i] = 0.05 + 0.02 * i
c.frame_begin(p)
c.set_dequant_tables(wl.tables)
c.set_lf_quantized(*wl.lf_q)
c.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
for g in range(wl.coeffs.shape[0]):
    c.submit_group(g, wl.coeffs[g])
c.slot_wait(0)
for _ in range(2):
    c.frame_run()
c.sync()
c.kernel_timing(True)
N = 10
for _ in range(N):
    c.frame_run()
c.sync()
kt = {k: round(v[0] / N, 4) for k, v in c.kernel_times().items()}
px = float(size) * size
print(json.dumps({"kernels_ms": kt,
                  "generate_GB_per_s": round(px * 12 / kt["k_noise_generate"] / 1e6, 1),
                  "apply_GB_per_s_algorithmic(36B/px)": round(px * 36 / kt["k_noise_apply"] / 1e6, 1)}))
