#!/usr/bin/env python3
"""Kernel times of a chroma-subsampled (JPEG-recompression-like) frame: 8x8 transforms only, no filters,
YCbCr -> RGB8 output.  usage: tools/bench_jpeg420.py [size] [420|422|440|444] [8x8|dct8]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import jxl_rs_amd
from jxl_rs_amd import synth

size = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
sub = sys.argv[2] if len(sys.argv) > 2 else "420"
hs, vs = {"420": ((1, 0, 1), (1, 0, 1)), "422": ((1, 0, 1), (0, 0, 0)), "440": ((0, 0, 0), (1, 0, 1)),
          "444": ((0, 0, 0), (0, 0, 0))}[sub]
mix = {"8x8": synth.MIX_8X8, "dct8": synth.MIX_DCT8}[sys.argv[3] if len(sys.argv) > 3 else "8x8"]
wl = synth.make_vardct(size, size, mix=mix, seed=3, unique_groups=24,
                       epf_iters=0, gab=False, lf_smoothing=False, hshift=hs, vshift=vs)
c = jxl_rs_amd.Context(0, n_slots=1)
p = synth.apply_opts(c.default_params(size, size), wl)
c.frame_begin(p)
c.set_dequant_tables(wl.tables)
c.set_lf_quantized(*wl.lf_q)
c.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
for g in range(wl.coeffs.shape[0]):
    c.submit_group(g, wl.coeffs[g])
c.slot_wait(0)
out = torch.empty((size, size, 3), dtype=torch.uint8, device="cuda:0")
N = 10
for _ in range(2):
    c.frame_run()
    c._chk(c.L.jxlh_frame_read_ycbcr_rgb8(c._ctx, 3, 0, size, out.data_ptr(), size * 3), "rgb8")
c.sync()
c.kernel_timing(True)
for _ in range(N):
    c.frame_run()
    c._chk(c.L.jxlh_frame_read_ycbcr_rgb8(c._ctx, 3, 0, size, out.data_ptr(), size * 3), "rgb8")
c.sync()
kt = {k: round(v[0] / N, 4) for k, v in c.kernel_times().items()}
total = sum(kt.values())
print(json.dumps({"workload": f"{size}x{size} {sub} {sys.argv[3] if len(sys.argv) > 3 else '8x8'} transforms, no filters, RGB8 out", "kernels_ms": kt,
                  "ms_per_frame": round(total, 4), "MP_per_s": round(size * size / total / 1e3, 1)}))
