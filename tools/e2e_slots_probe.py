#!/usr/bin/env python3
"""Pieces of the PCIe-inclusive path with the slot-bucketed 2-byte transport (jxlh_submit_groups_slots): the H2D + pack
alone, the kernels alone (per-kernel HIP events), the serial and the pipelined frame time with 2 / 3 contexts."""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import jxl_rs_amd
from jxl_rs_amd import synth

size = 8192
wl = synth.make_vardct(size, size, mix=synth.MIX_D1, seed=3, unique_groups=24, epf_iters=2)
ng = wl.coeffs.shape[0]
cache, es, cs, ns = {}, [], [], []
for g in range(ng):
    k = g % 24
    if k not in cache:
        cache[k] = synth.to_slots(wl.coeffs[g])
    es.append(cache[k][0]); cs.append(cache[k][1].reshape(-1)); ns.append(cache[k][2])
off = np.concatenate([[0], np.cumsum([len(x) for x in es])]).astype(np.int64)
tot = int(off[-1])
ns = np.concatenate(ns).astype(np.uint32)
ids = np.arange(ng, dtype=np.uint32)
out = {"MB_per_frame": round((tot * 2 + ng * 3072) / 1e6, 1)}


def make(nslots):
    c = jxl_rs_amd.Context(0, n_slots=nslots)
    c.frame_begin(synth.apply_opts(c.default_params(size, size), wl))
    c.set_dequant_tables(wl.tables); c.set_lf_quantized(*wl.lf_q)
    c.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
    return c


c0 = make(4)
pe, pe_a = c0.alloc_pinned(tot * 2)
pc, pc_a = c0.alloc_pinned(ng * 3072)
pe.view(np.uint16)[:tot] = np.concatenate(es)
pc[:] = np.concatenate(cs)


def submit(c, nslots):
    per = (ng + nslots - 1) // nslots
    for sl in range(nslots):
        g0, g1 = sl * per, min(ng, (sl + 1) * per)
        if g0 < g1:
            c.submit_groups_slots(ids[g0:g1], pe_a + int(off[g0]) * 2, pc_a + g0 * 3072, ns[3 * g0:3 * g1], None, slot=sl)


def resetup(c):
    c.frame_begin(synth.apply_opts(c.default_params(size, size), wl))
    c.set_dequant_tables(wl.tables); c.set_lf_quantized(*wl.lf_q)
    c.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)


for nslots in (1, 2, 4):
    resetup(c0)
    submit(c0, nslots); c0.frame_run(); c0.sync()
    t0 = time.perf_counter()
    for _ in range(6):
        c0.frame_begin(synth.apply_opts(c0.default_params(size, size), wl))
        submit(c0, nslots)
        [c0.slot_wait(s) for s in range(nslots)]
    out[f"h2d_and_pack_only_ms_{nslots}slots"] = round((time.perf_counter() - t0) / 6 * 1e3, 3)
resetup(c0)
submit(c0, 2); c0.frame_run(); c0.sync()
hs = []
for _ in range(6):
    t0 = time.perf_counter()
    submit(c0, 2)
    hs.append(time.perf_counter() - t0)
    c0.frame_run(); c0.sync()
out["host_side_submit_call_ms"] = round(sum(hs) / 6 * 1e3, 3)
c0.kernel_timing(True)
for _ in range(5):
    submit(c0, 2); c0.frame_run(); c0.sync()
out["kernels_ms"] = {k: round(v[0] / 5, 4) for k, v in c0.kernel_times().items()}
c0.kernel_timing(False)
t0 = time.perf_counter()
for _ in range(6):
    submit(c0, 2); c0.frame_run(); c0.sync()
out["serial_ms"] = round((time.perf_counter() - t0) / 6 * 1e3, 3)
for ne in (1, 2):
    cx = [c0] + [make(2) for _ in range(ne - 1)]
    for c in cx[1:]:
        submit(c, 2); c.frame_run(); c.sync()
    for i in range(ne + 4):
        c = cx[i % ne]; c.sync(); submit(c, 2); c.frame_run()
    [c.sync() for c in cx]
    t0 = time.perf_counter()
    for i in range(12):
        c = cx[i % ne]; c.sync(); submit(c, 2); c.frame_run()
    [c.sync() for c in cx]
    out[f"pipelined_ms_{ne}ctx"] = round((time.perf_counter() - t0) / 12 * 1e3, 3)
    # the same without waiting for the context's previous frame on the host: the library orders the slot streams
    # behind the consumers of its buffers with events (sp_expanded, k1_done)
    t0 = time.perf_counter()
    for i in range(12):
        c = cx[i % ne]; submit(c, 2); c.frame_run()
    [c.sync() for c in cx]
    out[f"pipelined_nosync_ms_{ne}ctx"] = round((time.perf_counter() - t0) / 12 * 1e3, 3)
    for c in cx[1:]:
        c.close()
# deeper pipelines with ONE slot stream per context (same number of streams as 2 contexts x 2 slots)
for ne, nsl in ((3, 1), (4, 1), (2, 1)):
    cx = [make(nsl) for _ in range(ne)]
    for c in cx:
        submit(c, nsl); c.frame_run(); c.sync()
    for i in range(ne + 4):
        c = cx[i % ne]; c.sync(); submit(c, nsl); c.frame_run()
    [c.sync() for c in cx]
    t0 = time.perf_counter()
    for i in range(12):
        c = cx[i % ne]; c.sync(); submit(c, nsl); c.frame_run()
    [c.sync() for c in cx]
    out[f"pipelined_ms_{ne}ctx_{nsl}slot"] = round((time.perf_counter() - t0) / 12 * 1e3, 3)
    for c in cx:
        c.close()
print(json.dumps(out))
c0.close()
