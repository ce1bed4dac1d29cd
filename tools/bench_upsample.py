#!/usr/bin/env python3
"""Kernel time of the 2x / 4x / 8x upsampling stage inside a frame run: a (8192/n)^2 VarDCT d1 frame upsampled
to 8192^2 (three channels)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jxl_rs_amd
from jxl_rs_amd import synth

out = 8192
res = {}
for n in (2, 4, 8):
    size = out // n
    wl = synth.make_vardct(size, size, mix=synth.MIX_D1, seed=3, unique_groups=24, epf_iters=2)
    c = jxl_rs_amd.Context(0, n_slots=1)
    p = synth.apply_opts(c.default_params(size, size), wl)
    p.upsampling = n
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
    ms = kt["k_upsample"]
    px_out = 3.0 * out * out
    res[f"{n}x"] = {"k_upsample_ms": ms, "out_GB_per_s": round(px_out * 4 / ms / 1e6, 1),
                    "GFLOP_per_s": round(px_out * 50 / ms / 1e6, 1), "frame_kernels_ms": kt}
    c.close()
print(json.dumps(res))
