#!/usr/bin/env python3
"""K1 of a frame resident in the slot-bucketed form, for a list of coefficient-density factors (and an outlier frame):
HIP-event time of the transforms, fallback share.  For A/B runs of library variants (JXLH_LIBRARY) and kernel traces.
  python tools/slots_sweep.py [size] [factor ...]      factor: 0.5 1 2 4 ... or `out` (1e-5 of entries at 2000..30000)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import jxl_rs_amd
from jxl_rs_amd import synth, lib as jl
size = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
factors = sys.argv[2:] or ["1", "out", "2", "4"]
wl = synth.make_vardct(size, size, mix=synth.MIX_D1, seed=3, unique_groups=24, epf_iters=2)
ng = wl.coeffs.shape[0]
rng = np.random.default_rng(603)
has_packer = hasattr(jl.load(), "jxlh_host_pack_slots")


def variant(c, f):
    c = c.copy()
    nz = c != 0
    if f == "out":   # (applied per GROUP below: one out-of-range coefficient in 1e-5 x entries-of-the-frame groups)
        idx = np.flatnonzero(c.reshape(-1))
        c.reshape(-1)[rng.choice(idx)] = int(rng.integers(2000, 30001)) * int(rng.choice([-1, 1]))
        return c
    f = float(f)
    if f < 1:
        c[nz & (rng.random(c.shape) >= f)] = 0
    elif f > 1:
        p_new = min(1.0, nz.mean() * (f - 1) / (1 - nz.mean()))
        add = ~nz & (rng.random(c.shape) < p_new)
        c[add] = ((1 + rng.geometric(0.5, size=c.shape)) * rng.choice([-1, 1], size=c.shape))[add]
    return c


c = jxl_rs_amd.Context(0, n_slots=1)
for f in factors:
    cache, e, cn, ns = {}, [], [], []
    pack = lambda v: jl.host_pack_slots(v, 0) if has_packer else synth.to_slots(v, split=True)
    hit = set()
    if f == "out":   # 1e-5 of the frame's ~17.3 M entries: 173 groups get one outlier each
        hit = set(rng.choice(ng, size=173, replace=False).tolist())
    for g in range(ng):
        k = g % 24
        if g in hit:
            q = pack(variant(wl.coeffs[g], f))
        else:
            if k not in cache:
                cache[k] = pack(variant(wl.coeffs[g], f if f != "out" else "1"))
            q = cache[k]
        assert len(q[3]) == 0
        e.append(q[0]); cn.append(q[1].reshape(-1)); ns.append(q[2])
    c.frame_begin(synth.apply_opts(c.default_params(size, size), wl))
    c.set_dequant_tables(wl.tables); c.set_lf_quantized(*wl.lf_q)
    c.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
    c.submit_groups_slots(np.arange(ng, dtype=np.uint32), np.concatenate(e), np.concatenate(cn), np.concatenate(ns), None)
    c.slot_wait(0)
    t_warm = time.perf_counter()
    while time.perf_counter() - t_warm < 0.08:
        c.frame_run()
        c.sync()
    N = 10
    t0 = time.perf_counter()
    for _ in range(N):
        c.frame_run()
    c.sync()
    wall = (time.perf_counter() - t0) / N * 1e3
    c.kernel_timing_reset(); c.kernel_timing(True)
    for _ in range(N):
        c.frame_run()
    c.sync()
    kt = {k: round(v[0] / N, 4) for k, v in c.kernel_times().items()}
    c.kernel_timing(False)
    extra = ""
    if hasattr(c.L, "jxlh_frame_k1_counters"):
        cnt = c.k1_counters()
        extra = " fallback %d / %d batches" % (sum(cnt["fallback_batches"].values()), sum(cnt["batches"].values()))
        if os.environ.get("SWEEP_RAW"):
            extra += " raw %s %s" % (list(cnt["fallback_batches"].values()), list(cnt["dense_route_varblocks"].values()))
        if os.environ.get("SWEEP_CLASSES"):
            extra += " " + " ".join("%s %d/%d" % (k, cnt["fallback_batches"][k], v) for k, v in cnt["batches"].items())
    print(f"factor {f}: frame {wall:.4f} ms, k1 {kt.get('k1_vardct')} ms, filters {kt.get('k23_fused_filters')}{extra}", flush=True)
