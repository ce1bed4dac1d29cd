#!/usr/bin/env python3
"""Round 5 diagnosis: why is the sparse-resident frame (K1 reads the bucketed pairs: 0.27-0.30 ms instead of 0.39) no
faster than the dense one with two frames in flight?  Dense and sparse resident frames, 1 and 2 contexts in flight,
1 and 2 slot streams per context (the runtime maps streams onto a few hardware queues round robin: two main streams on
one queue do not overlap), per-kernel HIP-event tables."""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import jxl_rs_amd
from jxl_rs_amd import synth

size = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
wl = synth.make_vardct(size, size, mix=synth.MIX_D1, seed=3, unique_groups=24, epf_iters=2, gab=True, lf_smoothing=True)
ng = wl.coeffs.shape[0]
cache, runs, ns = {}, [], []
for g in range(ng):
    k = g % 24
    if k not in cache:
        cache[k] = synth.to_sparse(wl.coeffs[g])
    runs.append(cache[k][0]); ns.append(cache[k][1])
runs = np.concatenate(runs); ns = np.concatenate(ns)
ids = np.arange(ng, dtype=np.uint32)


scache, sent, scnt, sn = {}, [], [], []
for g in range(ng):
    k = g % 24
    if k not in scache:
        scache[k] = synth.to_slots(wl.coeffs[g])
    sent.append(scache[k][0]); scnt.append(scache[k][1].reshape(-1)); sn.append(scache[k][2])
sent = np.concatenate(sent); scnt = np.concatenate(scnt); sn = np.concatenate(sn)


def make(nslots, sparse):
    c = jxl_rs_amd.Context(0, n_slots=nslots)
    c.frame_begin(synth.apply_opts(c.default_params(size, size), wl))
    c.set_dequant_tables(wl.tables); c.set_lf_quantized(*wl.lf_q)
    c.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
    if sparse == "slots":
        c.submit_groups_slots(ids, sent, scnt, sn, None)
    elif sparse:
        c.submit_groups_sparse(ids, runs, ns, None)
    else:
        for g in range(ng):
            c.submit_group(g, wl.coeffs[g])
    c.slot_wait(0)
    c.frame_run(); c.sync()
    return c


def timed(ctxs, n=20, reps=5):
    for i in range(4):
        ctxs[i % len(ctxs)].frame_run()
    for c in ctxs:
        c.sync()
    res = []
    for _ in range(reps):
        t0 = time.perf_counter()
        for i in range(n):
            ctxs[i % len(ctxs)].frame_run()
        for c in ctxs:
            c.sync()
        res.append((time.perf_counter() - t0) / n * 1e3)
    return round(sorted(res)[len(res) // 2], 4)


out = {}
modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["dense", "pairs", "slots"]
for mode in modes:
    sparse = {"dense": False, "pairs": True, "slots": "slots"}[mode]
    for nslots in (1, 2):
        cs = [make(nslots, sparse) for _ in range(2)]
        key = f"{mode}_slots{nslots}"
        out[key] = {"inflight1": timed(cs[:1]), "inflight2": timed(cs)}
        if nslots == 1:
            c = cs[0]
            c.kernel_timing(True)
            for _ in range(10):
                c.frame_run()
            c.sync()
            out[key]["kernels_ms"] = {k: round(v[0] / 10, 4) for k, v in c.kernel_times().items()}
            c.kernel_timing(False)
        for c in cs:
            c.close()
print(json.dumps(out, indent=1))
