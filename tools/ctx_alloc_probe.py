#!/usr/bin/env python3
"""Does a context's position in the process's allocation history matter?  K1 / filters HIP-event times of the 8192^2 d1
dense frame on context A (created first, like bench.py's headline contexts), again after ~6 s of GPU work, then on a
context B created afterwards, then on A again.   python tools/ctx_alloc_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import jxl_rs_amd
from jxl_rs_amd import synth
size = 8192
wl = synth.make_vardct(size, size, mix=synth.MIX_D1, seed=3, unique_groups=24, epf_iters=2)


def make():
    c = jxl_rs_amd.Context(0, n_slots=1)
    c.frame_begin(synth.apply_opts(c.default_params(size, size), wl))
    c.set_dequant_tables(wl.tables); c.set_lf_quantized(*wl.lf_q)
    c.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
    for g in range(wl.coeffs.shape[0]):
        c.submit_group(g, wl.coeffs[g])
    c.slot_wait(0)
    return c


def times(c, tag):
    for _ in range(5):
        c.frame_run()
    c.sync()
    c.kernel_timing_reset(); c.kernel_timing(True)
    N = 20
    for _ in range(N):
        c.frame_run()
    c.sync()
    kt = {k: round(v[0] / N, 4) for k, v in c.kernel_times().items()}
    c.kernel_timing(False)
    t0 = time.perf_counter()
    for _ in range(N):
        c.frame_run()
    c.sync()
    wall = (time.perf_counter() - t0) / N * 1e3
    pl, _ = c.device_planes()
    cb = c.coeff_buffer()[0]
    print(f"{tag}: k1 {kt.get('k1_vardct')} filters {kt.get('k23_fused_filters')} wall {wall:.4f}  coeffs {cb:#x} planes {pl[0]:#x} {pl[1]:#x} {pl[2]:#x}", flush=True)


def warm(c, sec=1.0):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < sec:
        for _ in range(50):
            c.frame_run()
        c.sync()


def rebegin_dense(c):
    c.frame_begin(synth.apply_opts(c.default_params(size, size), wl))
    c.set_dequant_tables(wl.tables); c.set_lf_quantized(*wl.lf_q)
    c.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
    for g in range(wl.coeffs.shape[0]):
        c.submit_group(g, wl.coeffs[g])
    c.slot_wait(0)


from jxl_rs_amd import lib as jl
import ctypes as C
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]


def placement(c, tag):
    """device-to-device copy coeffs -> plane (268 MB) and a fill of a plane, GB/s"""
    pl, _ = c.device_planes()
    cb = c.coeff_buffer()[0]
    n = size * size * 4
    out = []
    for dst in pl:
        hip.hipMemcpy(dst, cb, n, 3); hip.hipDeviceSynchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            hip.hipMemcpy(dst, cb, n, 3)
        hip.hipDeviceSynchronize()
        out.append(2 * n * 10 / (time.perf_counter() - t0) / 1e9)
    for off in (0, n, 2 * n):
        src = cb + off
        hip.hipMemcpy(pl[0], src, n, 3); hip.hipDeviceSynchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            hip.hipMemcpy(pl[0], src, n, 3)
        hip.hipDeviceSynchronize()
        out.append(2 * n * 10 / (time.perf_counter() - t0) / 1e9)
    print(f"   placement {tag}: coeffs[0:268MB] -> plane 0/1/2: {out[0]:.0f} {out[1]:.0f} {out[2]:.0f} GB/s; coeffs third 0/1/2 -> plane 0: {out[3]:.0f} {out[4]:.0f} {out[5]:.0f}", flush=True)


a = make()
b = make()
warm(a); times(a, "A (first context), warm")
warm(b); times(b, "B (second context), warm")
# a context that first held the slot-bucketed form, then the dense slabs (bench.py's sweep contexts)
e = jxl_rs_amd.Context(0, n_slots=1)
e.frame_begin(synth.apply_opts(e.default_params(size, size), wl))
e.set_dequant_tables(wl.tables); e.set_lf_quantized(*wl.lf_q)
e.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
cache = {}
ents, cnts, ns = [], [], []
for g in range(wl.coeffs.shape[0]):
    k = g % 24
    if k not in cache:
        cache[k] = jl.host_pack_slots(wl.coeffs[g], 0)
    q = cache[k]
    ents.append(q[0]); cnts.append(q[1].reshape(-1)); ns.append(q[2])
e.submit_groups_slots(np.arange(wl.coeffs.shape[0], dtype=np.uint32), np.concatenate(ents), np.concatenate(cnts), np.concatenate(ns), None)
e.slot_wait(0)
warm(e); times(e, "E, slot form, warm")
rebegin_dense(e)
warm(e); times(e, "E, re-begun with dense slabs, warm")
rebegin_dense(a)
warm(a); times(a, "A, re-begun with dense slabs, warm")
f = make()
warm(f); times(f, "F (created last), warm")
times(a, "A at the end")
for c, t in ((a, "A"), (b, "B"), (e, "E"), (f, "F")):
    placement(c, t)
