#!/usr/bin/env python3
"""Times the pieces of the PCIe-inclusive path separately (H2D of the sparse pairs, the device
zero-fill + scatter, the reconstruction kernels) to see what overlaps."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import jxl_rs_amd
from jxl_rs_amd import synth

size = 8192
wl = synth.make_vardct(size, size, mix=synth.MIX_D1, seed=3, unique_groups=24, epf_iters=2)
ng = wl.coeffs.shape[0]
c = jxl_rs_amd.Context(0, n_slots=4)
c.frame_begin(synth.apply_opts(c.default_params(size, size), wl))
c.set_dequant_tables(wl.tables); c.set_lf_quantized(*wl.lf_q)
c.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
cache, runs, ns = {}, [], []
for g in range(ng):
    k = g % 24
    if k not in cache:
        cache[k] = synth.to_sparse(wl.coeffs[g])
    runs.append(cache[k][0]); ns.append(cache[k][1])
total = sum(len(r) for r in runs)
pin, addr = c.alloc_pinned(total * 4)
pin.view(np.uint32)[:total] = np.concatenate(runs)
ns = np.concatenate(ns).astype(np.uint32)
offs = np.concatenate([[0], np.cumsum(ns.reshape(ng, 3).sum(axis=1))]).astype(np.int64)
ids = np.arange(ng, dtype=np.uint32)
for nslots in (1, 2, 4):
    per = (ng + nslots - 1) // nslots
    def submit():
        for sl in range(nslots):
            g0, g1 = sl * per, min(ng, (sl + 1) * per)
            c.submit_groups_sparse(ids[g0:g1], addr + int(offs[g0]) * 4, ns[3 * g0:3 * g1], None, slot=sl)
    c.frame_begin(synth.apply_opts(c.default_params(size, size), wl))
    submit(); [c.slot_wait(s) for s in range(nslots)]
    c.frame_begin(synth.apply_opts(c.default_params(size, size), wl))  # drop pending
    c.set_dequant_tables(wl.tables); c.set_lf_quantized(*wl.lf_q)
    c.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
    t0 = time.perf_counter()
    for _ in range(5):
        c.frame_begin(synth.apply_opts(c.default_params(size, size), wl))
        submit(); [c.slot_wait(s) for s in range(nslots)]
    el = (time.perf_counter() - t0) / 5
    print(f"H2D only, {nslots} slots: {el*1e3:.3f} ms  ({total*4/el/1e9:.1f} GB/s)")
c.frame_begin(synth.apply_opts(c.default_params(size, size), wl))
c.set_dequant_tables(wl.tables); c.set_lf_quantized(*wl.lf_q)
c.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
c.kernel_timing(True)
for _ in range(5):
    submit(); c.frame_run(); c.sync()
kt = c.kernel_times()
print({k: round(v[0] / 5, 4) for k, v in kt.items()})
t0 = time.perf_counter()
for _ in range(5):
    submit(); c.frame_run(); c.sync()
print("serial submit+run+sync:", (time.perf_counter() - t0) / 5 * 1e3, "ms")
# ---- two contexts, pipelined
c2 = jxl_rs_amd.Context(0, n_slots=4)
c2.frame_begin(synth.apply_opts(c2.default_params(size, size), wl))
c2.set_dequant_tables(wl.tables); c2.set_lf_quantized(*wl.lf_q)
c2.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
c.kernel_timing(False)
cs = [c, c2]
def submit_to(cc):
    for sl in range(nslots):
        g0, g1 = sl * per, min(ng, (sl + 1) * per)
        cc.submit_groups_sparse(ids[g0:g1], addr + int(offs[g0]) * 4, ns[3 * g0:3 * g1], None, slot=sl)
for cc in cs:
    submit_to(cc); cc.frame_run()
for cc in cs:
    cc.sync()
t0 = time.perf_counter()
N = 20
for i in range(N):
    cc = cs[i % 2]
    cc.sync(); submit_to(cc); cc.frame_run()
for cc in cs:
    cc.sync()
print("pipelined 2 contexts:", (time.perf_counter() - t0) / N * 1e3, "ms/frame")
# ---- does a large second pinned allocation change the picture? (bench.py holds the dense slabs too)
big, big_addr = c.alloc_pinned(wl.coeffs.nbytes)
big.view(np.int32)[:] = wl.coeffs.reshape(-1)
t0 = time.perf_counter()
for i in range(N):
    cc = cs[i % 2]
    cc.sync(); submit_to(cc); cc.frame_run()
for cc in cs:
    cc.sync()
print("pipelined, after 805 MB pinned alloc:", (time.perf_counter() - t0) / N * 1e3, "ms/frame")
