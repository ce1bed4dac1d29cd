"""Fused filter kernel time for three EPF populations (spec draw, a random half of the blocks filtered, every block filtered)
on the 8K d1 frame; usage: JXLH_LIBRARY=<variant> python tools/filter_pop_time.py"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np  # noqa: E402
import jxl_rs_amd  # noqa: E402
from jxl_rs_amd import synth  # noqa: E402

size = 8192
wl = synth.make_vardct(size, size, mix=synth.MIX_D1, seed=3, unique_groups=24, epf_iters=2, gab=True)
c = jxl_rs_amd.Context(0, n_slots=1)
c.frame_begin(synth.apply_opts(c.default_params(size, size), wl))
c.set_dequant_tables(wl.tables)
c.set_lf_quantized(*wl.lf_q)
for g in range(wl.coeffs.shape[0]):
    c.submit_group(g, wl.coeffs[g])
c.slot_wait(0)
pick = np.random.default_rng(53).random(wl.epf_map.shape) < 0.5
pops = {"spec": (wl.raw_quant, wl.epf_map),
        "half": (np.where(pick, np.minimum(wl.raw_quant, 4), wl.raw_quant), np.where(pick, 7, 0).astype(wl.epf_map.dtype)),
        "all": (np.minimum(wl.raw_quant, 4), np.full_like(wl.epf_map, 7))}
out = {}
# JXLH_POP_ORDER=spec,all,half,all,half: the order matters (a population measured right after the all-filtered one runs
# on a chip that has just drawn more power: round 5 found the driver line's "half slower than all" to be that)
order = os.environ.get("JXLH_POP_ORDER", "spec,half,all").split(",")
for idx, name in enumerate(order):
    rq, em = pops[name]
    c.set_hf_meta(wl.transform_map, rq, em, wl.ytox, wl.ytob)
    for _ in range(3):
        c.frame_run()
    c.sync()
    c.kernel_timing_reset()
    c.kernel_timing(True)
    for _ in range(10):
        c.frame_run()
    c.sync()
    kt = c.kernel_times()
    c.kernel_timing(False)
    out[f"{idx}:{name}"] = round(kt["k23_fused_filters"][0] / 10, 4)
print(json.dumps(out))
c.close()
