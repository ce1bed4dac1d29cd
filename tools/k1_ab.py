"""K1 / filter kernel times of one library build (JXLH_LIBRARY) on fixed frames: A/B runs of two builds on one box.
usage: JXLH_LIBRARY=<so> python tools/k1_ab.py"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import jxl_rs_amd  # noqa: E402
from jxl_rs_amd import synth  # noqa: E402

out = {}
for name, size, mix, seed in (("8k_d1", 8192, synth.MIX_D1, 3), ("16k_all", 16384, synth.MIX_ALL, 4)):
    wl = synth.make_vardct(size, size, mix=mix, seed=seed, unique_groups=24 if size <= 8192 else 32, epf_iters=2, gab=True)
    c = jxl_rs_amd.Context(0, n_slots=1)
    c.frame_begin(synth.apply_opts(c.default_params(size, size), wl))
    c.set_dequant_tables(wl.tables)
    c.set_lf_quantized(*wl.lf_q)
    c.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
    for g in range(wl.coeffs.shape[0]):
        c.submit_group(g, wl.coeffs[g])
    c.slot_wait(0)
    for _ in range(3):
        c.frame_run()
    c.sync()
    c.kernel_timing_reset()
    c.kernel_timing(True)
    for _ in range(8):
        c.frame_run()
    c.sync()
    kt = c.kernel_times()
    c.kernel_timing(False)
    out[name] = {k: round(v[0] / 8, 4) for k, v in kt.items() if k in ("k1_vardct", "k23_fused_filters")}
    c.close()
print(json.dumps(out))
