import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import jxl_rs_amd
from jxl_rs_amd import synth
size=16384
wl = synth.make_vardct(size, size, mix=synth.MIX_ALL, seed=3, unique_groups=32, epf_iters=2, gab=True)
c = jxl_rs_amd.Context(0, n_slots=1)
c.frame_begin(synth.apply_opts(c.default_params(size, size), wl))
c.set_dequant_tables(wl.tables); c.set_lf_quantized(*wl.lf_q)
c.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
for g in range(wl.coeffs.shape[0]): c.submit_group(g, wl.coeffs[g])
c.slot_wait(0)
for _ in range(5): c.frame_run()
c.sync()
c.kernel_timing_reset(); c.kernel_timing(True)
for _ in range(6): c.frame_run()
c.sync()
print({k: round(v[0]/6,4) for k,v in c.kernel_times().items()})
