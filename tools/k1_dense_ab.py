"""Dense-slab K1 (and the whole frame) at 4096^2 (config 2: EPF off) and 8192^2 (config 3) for one library build.
usage: JXLH_LIBRARY=<so> python tools/k1_dense_ab.py"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import jxl_rs_amd
from jxl_rs_amd import synth
out = {}
for name, size, epf in (("4k_epf0", 4096, 0), ("8k_d1", 8192, 2)):
    wl = synth.make_vardct(size, size, mix=synth.MIX_D1, seed=3, unique_groups=24, epf_iters=epf, gab=True)
    c = jxl_rs_amd.Context(0, n_slots=1)
    c.frame_begin(synth.apply_opts(c.default_params(size, size), wl))
    c.set_dequant_tables(wl.tables); c.set_lf_quantized(*wl.lf_q)
    c.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
    for g in range(wl.coeffs.shape[0]):
        c.submit_group(g, wl.coeffs[g])
    c.slot_wait(0)
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.08:
        c.frame_run(); c.sync()
    t0 = time.perf_counter()
    for _ in range(20):
        c.frame_run()
    c.sync()
    wall = (time.perf_counter() - t0) / 20 * 1e3
    c.kernel_timing_reset(); c.kernel_timing(True)
    for _ in range(10):
        c.frame_run()
    c.sync()
    kt = c.kernel_times(); c.kernel_timing(False)
    out[name] = {"frame_ms": round(wall, 4), "k1_ms": round(kt["k1_vardct"][0] / 10, 4)}
    c.close()
print(json.dumps(out))
