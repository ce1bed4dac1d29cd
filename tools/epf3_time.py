#!/usr/bin/env python3
"""epf_iters = 3 at size^2 (all transform types, spec population): per-kernel times of one frame run (A/B of library variants)."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import jxl_rs_amd
from jxl_rs_amd import synth
size = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
wl = synth.make_vardct(size, size, mix=synth.MIX_ALL, seed=4, unique_groups=24, epf_iters=3)
c = jxl_rs_amd.Context(0, n_slots=1)
c.frame_begin(synth.apply_opts(c.default_params(size, size), wl))
c.set_dequant_tables(wl.tables); c.set_lf_quantized(*wl.lf_q)
c.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
for g in range(wl.coeffs.shape[0]):
    c.submit_group(g, wl.coeffs[g])
c.slot_wait(0)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.1:
    c.frame_run(); c.sync()
c.kernel_timing_reset(); c.kernel_timing(True)
for _ in range(5):
    c.frame_run()
c.sync(); c.kernel_timing(False)
print(json.dumps({"lib": os.path.basename(os.environ.get("JXLH_LIBRARY", "libjxl_hip.so")),
                  "kernels_ms": {k: round(v[0] / v[1], 4) for k, v in c.kernel_times().items()}}))
