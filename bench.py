#!/usr/bin/env python3
"""bench.py -- megapixels/s of the JPEG XL VarDCT reconstruction hot path on MI355X.

A "step" is one pass of the whole device chain (K0b LF smoothing, K3 sigma map, K1
dequant+CfL+LLF+IDCT, Gaborish, EPF1, EPF2) over one synthetic 8192x8192 VarDCT d1 frame
(BASELINE.json configs[2]) whose inputs -- 805 MB of i32 coefficients, the HfMetadata maps,
the quantised LF, the dequant tables -- are resident in HBM before the timed region starts.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--size 8192] [--cpu-sample 8192]

N > 1 (launched by torch.distributed.run, one rank per GPU) measures BOTH multi-GPU forms in one run:
  * weak (the top-level `value`): frames are independent units, every rank reconstructs its own frame of the
    same shape with no data-path collective; value = all ranks' pixels / max-over-ranks time;
  * strong (`strong_scaling`): ONE frame cut into bands of group rows, one band per rank -- transforms on the
    own band, halo exchange of the edge block rows (ncclSend/ncclRecv), filters, in-place ncclAllGather of
    the finished planes so that every rank holds the whole frame (the layout north_star describes).  The
    RCCL communicator is the library's own (jxlh_comm_init); torch.distributed only launches, broadcasts the
    128-byte id, and does the barrier / max-over-ranks of the timing protocol.  The weak line is measured first; if the
    sharded legs have not come back after JXLH_BENCH_STRONG_TIMEOUT_S seconds (default 240) rank 0 prints the weak
    line with them marked as timed out and every rank leaves.

Rank 0 prints ONE JSON line: metric/value/... + "roofline" (every kernel of the chain with its own
algorithmic bytes and fraction, HIP-event timed on the kernels' own stream; the chain against the
fused-ideal numerator; the all-blocks-filtered EPF population next to the spec one; measured copy
ceilings) + "cpu_baseline" (the CPU oracle, a C port of the reference, on the same frame size).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FUSED_IDEAL_BYTES_PER_PX = 24.4  # compulsory traffic of the whole chain as ONE kernel
# algorithmic (compulsory) HBM bytes per pixel of each kernel, SURVEY.md section 8(d) / DESIGN.md
ALGO_BYTES_PER_PX = {
    "k1_vardct": 24.3,         # 12 B coeffs in + 12 B planes out + maps/LF (scan + class kernels)
    "k2_gaborish": 24.0,       # 3 ch x (4 in + 4 out)
    "k3a_epf0": 24.06, "k3b_epf1": 24.06, "k3c_epf2": 24.06,
    "k23_fused_filters": 24.06,
}


PLACEMENT_TRIALS = [12]  # --placement-trials: candidate buffer sets a VarDCT context picks its placement from (setup)
PLACEMENT_LOG = []


def new_context(jxl_rs_amd, device, n_slots=1, trials=None, tag=""):
    """A context whose first allocation of the large buffers is a pick among several placements
    (jxlh_ctx_tune_placement: where the driver puts them moves K1 by up to 10 %, profiles/r06_q_context_placement.txt).
    Part of setup, outside every timed region; `setup.placement` in the JSON line says what was picked."""
    c = jxl_rs_amd.Context(device, n_slots=n_slots)
    t = PLACEMENT_TRIALS[0] if trials is None else trials
    if t > 1 and hasattr(c.L, "jxlh_ctx_tune_placement"):
        c.tune_placement(t)
        begin = c.frame_begin
        done = []

        def frame_begin(*a, **k):  # the pick happens inside the first frame_begin: note what it saw
            r = begin(*a, **k)
            if not done:
                done.append(1)
                ratings, pick = c.tune_placement()
                if pick >= 0:
                    PLACEMENT_LOG.append({"context": tag, "candidates": len(ratings), "picked": pick,
                                          "picked_ms": [round(v, 4) for v in ratings[pick]],
                                          "worst_ms": [round(max(r[0] for r in ratings), 4), round(max(r[1] for r in ratings), 4)]})
            return r
        c.frame_begin = frame_begin
    return c


def placement_summary():
    return list(PLACEMENT_LOG)


def resolve_traffic(dominant, root=ROOT):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this same command
    (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE; tools/gpu_profile.sh, tools/pmc_summary.py).  PMC needs its own
    profiler run, so the value is the latest committed measurement THAT HOLDS THE DOMINANT KERNEL'S ENTRIES (other
    profiles -- the Modular kernels, secondary configs -- have their own r*_traffic.json files), not live.
    Returns (bytes or None, file or None)."""
    import glob
    # "k1_vardct" is the scan + class kernels: their PMC entries are k1_scan, k1_dct8, ...
    prefix = "k1_" if dominant == "k1_vardct" else dominant
    for path in sorted(glob.glob(os.path.join(root, "profiles", "r*_traffic.json")), reverse=True):
        try:
            pmc = json.load(open(path))
        except (OSError, ValueError):
            continue
        if pmc.get("_workload", "8192 d1") != "8192 d1":
            continue
        parts = [int(v["hbm_bytes"]) for k, v in pmc.items() if k.startswith(prefix) and isinstance(v, dict)]
        if parts:
            return sum(parts), os.path.relpath(path, root)
    return None, None


def chain_traffic(workload="8192 d1", root=ROOT):
    """HBM bytes of the WHOLE chain per frame (every kernel's FETCH x2 + WRITE) from the newest committed counter file of
    that workload ("8192 d1": dense slabs resident; "8192 d1 slots": the slot-bucketed form resident).
    Returns (bytes or None, file or None)."""
    import glob
    for path in sorted(glob.glob(os.path.join(root, "profiles", "r*_traffic.json")), reverse=True):
        try:
            pmc = json.load(open(path))
        except (OSError, ValueError):
            continue
        if pmc.get("_workload", "8192 d1") != workload:
            continue
        parts = [int(v["hbm_bytes"]) for k, v in pmc.items() if isinstance(v, dict) and "hbm_bytes" in v]
        if any(k.startswith("k1_") for k in pmc) and any(k.startswith("k23_fused") for k in pmc):
            return sum(parts), os.path.relpath(path, root)
    return None, None


def time_vardct_config(jxl_rs_amd, synth, np, device, size, mix, epf_iters, seed, steps, warmup=2, extra_epf=None):
    """One secondary VarDCT configuration, ONE frame in flight, inputs HBM-resident: wall-clock step time (sync on both
    sides) + the per-kernel HIP-event table with every kernel against its own algorithmic bytes.  extra_epf: also
    time the same frame with that epf_iters (new frame epoch, coefficients re-submitted)."""
    t0 = time.time()
    # every transform type of the mix must be in the frame (BASELINE configs[4] names DCT256X256; with 32 distinct
    # group tilings a 4 %-of-area type can miss the draw): further seeds until it is, and say which seed it was
    want_types = set(mix.keys())
    for used_seed in range(seed, seed + 40):
        wl = synth.make_vardct(size, size, mix=mix, seed=used_seed, unique_groups=24 if size <= 8192 else 32,
                               epf_iters=epf_iters, gab=True, lf_smoothing=True)
        have = set(np.unique(wl.transform_map[wl.transform_map >= 128] & 127).tolist())
        if want_types <= have:
            break
    assert want_types <= have, f"types {sorted(want_types - have)} missing from the synthetic frame"
    gen_s = time.time() - t0
    ctx = new_context(jxl_rs_amd, device, trials=min(PLACEMENT_TRIALS[0], 12 if size <= 8192 else 6), tag=f"secondary {size}")
    npx = size * size

    def measure(iters):
        wl.opts["epf_iters"] = iters
        ctx.frame_begin(synth.apply_opts(ctx.default_params(size, size), wl))
        ctx.set_dequant_tables(wl.tables)
        ctx.set_lf_quantized(*wl.lf_q)
        ctx.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
        for g in range(wl.coeffs.shape[0]):
            ctx.submit_group(g, wl.coeffs[g])
        ctx.slot_wait(0)
        for _ in range(warmup):
            ctx.frame_run()
        ctx.sync()
        t_warm = time.perf_counter()   # ... and until the device has been busy for 80 ms (see kernel_table in main)
        while time.perf_counter() - t_warm < 0.08:
            ctx.frame_run()
            ctx.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.frame_run()
        ctx.sync()
        ms = (time.perf_counter() - t0) * 1e3 / steps
        ctx.kernel_timing_reset()
        ctx.kernel_timing(True)
        nprof = max(2, min(steps, 5))
        for _ in range(nprof):
            ctx.frame_run()
        ctx.sync()
        kt = ctx.kernel_times()
        ctx.kernel_timing(False)
        kernels = {}
        for name, (kms, n) in kt.items():
            k = {"ms_per_step": round(kms / nprof, 4), "launches_per_step": n // nprof}
            if name in ALGO_BYTES_PER_PX:
                ab = ALGO_BYTES_PER_PX[name] * npx
                k["algorithmic_bytes"] = int(ab)
                k["achieved_GBs"] = round(ab / (kms / nprof * 1e-3) / 1e9, 1)
                k["frac"] = round(k["achieved_GBs"] / HBM_PEAK_GBS, 4)
            kernels[name] = k
        chain_ms = sum(v["ms_per_step"] for v in kernels.values())
        ideal = (24.3 if iters == 0 else FUSED_IDEAL_BYTES_PER_PX) * npx
        return {"epf_iters": iters, "ms_per_step": round(ms, 4), "value": round(npx / 1e6 / (ms / 1e3), 1), "unit": "MP/s",
                "steps": steps, "frames_in_flight": 1, "kernels": kernels, "sum_of_kernels_ms": round(chain_ms, 4),
                "chain_vs_fused_ideal": {"algorithmic_bytes": int(ideal),
                                         "frac": round(ideal / (chain_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}

    types = sorted(set(np.unique(wl.transform_map[wl.transform_map >= 128] & 127).tolist()))
    out = {"workload": f"{size}x{size} VarDCT, {len(types)} transform types present, CfL, LF smoothing, Gaborish, "
                       f"EPF iters={epf_iters}, inputs HBM-resident", "transform_types": types, "seed": used_seed,
           "host_generate_s": round(gen_s, 2)}
    out.update(measure(epf_iters))
    if extra_epf is not None:
        out[f"epf_iters_{extra_epf}"] = measure(extra_epf)
    ctx.close()
    return out


def time_modular_config(jxl_rs_amd, np, device, size, steps, cores, cpu=True):
    """BASELINE configs[3]: the default squeeze chain of a size x size image on three channels + YCoCg RCT (one
    jxlh_unsqueeze_chain call, the sequence tests/test_gpu_fullsize.py holds to the oracle at this size), and the
    256-colour palette expansion.  Fractions: the chain against SURVEY 8(d)'s 16 B per final sample (and against what
    its levels really write, 8 B per written sample); the palette against 16 B/px."""
    from jxl_rs_amd.modular import ModularChain
    from jxl_rs_amd.lib import DeviceArray
    ctx = jxl_rs_amd.Context(device, n_slots=1)
    t0 = time.time()
    ch = ModularChain(ctx, size, size, seed=84)
    gen_s = time.time() - t0
    npx = size * size

    def timed(fn, reps):
        # 60 ms of untimed calls first: after the host-side set-up the device has idled and its first calls run 15-25 %
        # slower than the steady state (the same effect as between the EPF populations of the headline)
        t_warm = time.perf_counter()
        while time.perf_counter() - t_warm < 0.06:
            fn()
            ctx.sync()
        ctx.timer_start()
        for _ in range(reps):
            fn()
        return ctx.timer_stop() / reps

    chain_ms = timed(ch.run_chain, steps)
    # the same call with one launch per streamed level instead of the dataflow launch (JXLH_CHAIN_FLOW=0: what rounds
    # 2-4 timed, now with the vector movers), for the A/B on the box the line is measured on
    os.environ["JXLH_CHAIN_FLOW"] = "0"
    levels_ms = timed(ch.run_chain, steps)
    del os.environ["JXLH_CHAIN_FLOW"]
    t0 = time.perf_counter()
    for _ in range(steps):
        ch.run_chain()
    ctx.sync()
    wall_ms = (time.perf_counter() - t0) * 1e3 / steps
    ctx.kernel_timing_reset()
    ctx.kernel_timing(True)
    ch.run_chain()
    ctx.sync()
    ctx.kernel_timing(False)
    rng = np.random.default_rng(256)
    pal = rng.integers(0, 256, size=(3, 256)).astype(np.int32)
    idx = rng.integers(0, 256, size=(size, size)).astype(np.int32)
    d_idx, d_pal, d_out = (DeviceArray(idx, device=device), DeviceArray(pal, device=device),
                           DeviceArray(nbytes=3 * npx * 4, device=device))
    pal_ms = timed(lambda: ctx._chk(ctx.L.jxlh_palette(ctx._ctx, d_idx.ptr, npx, d_pal.ptr, 256, 256, 3, 8, d_out.ptr),
                                    "palette"), steps)
    rct_ms = timed(lambda: ctx._chk(ctx.L.jxlh_rct(ctx._ctx, ch.d_out[0].ptr, ch.d_out[1].ptr, ch.d_out[2].ptr, npx, 6, 0),
                                    "rct"), steps)
    final = 16.0 * 3 * npx
    out = {"workload": f"{size}x{size} x 3 ch i32 Modular: default squeeze chain ({len(ch.steps)} steps) + YCoCg RCT "
                       f"(one jxlh_unsqueeze_chain call), 256-colour palette; device-resident",
           "dtype": "i32", "host_generate_s": round(gen_s, 2),
           "chain": {"ms": round(chain_ms, 4), "wall_ms": round(wall_ms, 4), "value": round(npx / 1e6 / (chain_ms / 1e3), 1),
                     "unit": "MP/s", "algorithmic_bytes": int(final), "bytes_per_final_sample": 16,
                     "achieved_GBs": round(final / (chain_ms * 1e-3) / 1e9, 1),
                     "frac": round(final / (chain_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "level_by_level_ms": round(levels_ms, 4),
                     "launches": "first levels in LDS + the streamed levels as one dataflow launch + last level fused with the RCT",
                     "bytes_written_levels": int(8.0 * ch.samples_written),
                     "dependent_steps_on_the_longest_line": int(sum(ow if hz else oh for hz, ow, oh in ch.steps) // 2)},
           "palette": {"ms": round(pal_ms, 4), "algorithmic_bytes": int(16.0 * npx),
                       "frac": round(16.0 * npx / (pal_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
           "rct_alone": {"ms": round(rct_ms, 4), "algorithmic_bytes": int(24.0 * npx),
                         "frac": round(24.0 * npx / (rct_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
           "pmc_file": "profiles/r05_k_modular_pmc.txt", "timeline_file": "profiles/r05_i_modular_flow.txt"}
    if cpu:
        # the oracle's step-by-step chain + RCT on the same planes, one thread per channel (ctypes releases the GIL)
        from concurrent.futures import ThreadPoolExecutor
        from oracle.oracle import Oracle
        o = Oracle(fused=True)

        def one(c):
            cur = ch.base[c]
            for (hz, ow, oh), res in zip(ch.steps, ch.residuals):
                cur = o.unsqueeze_h(cur, res[c], ow) if hz else o.unsqueeze_v(cur, res[c], oh)
            return cur

        t0 = time.perf_counter()
        with ThreadPoolExecutor(3) as ex:
            planes = list(ex.map(one, range(3)))
        o.rct(planes, 6, 0)
        el = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(npx / 1e6 / el, 2), "unit": "MP/s", "cores": 3, "kind": "port",
                               "sample": f"one {size}x{size} x 3 chain + RCT on the C oracle, one thread per channel "
                                         f"(the recurrence is serial along a line; scalar C, not the reference's SIMD)",
                               "seconds": round(el, 3)}
    for d in (d_idx, d_pal, d_out):
        d.free()
    ch.free()
    ctx.close()
    return out


def slot_content_sweep(jxl_rs_amd, synth, np, ectx, wl, size, steps, reps, cores, seed):
    """The slot-bucketed form on content that is NOT the synthetic d1 frame: for each variant of
    the frame's coefficients -- values far outside the entries' 10 bits (split into repeated in-range entries by the C
    packer), the coefficient density x0.5 / x2 / x4, one group submitted as a dense slab -- the frame resident in the
    slot-bucketed form (two contexts round robin like the headline, median of `reps` repetitions of `steps` frames),
    K1's HIP-event time, the share of K1's batches that left the direct path for the dense dequantisation pass, and
    the SAME coefficients resident as dense slabs beside it."""
    from concurrent.futures import ThreadPoolExecutor
    from jxl_rs_amd import lib as jl
    ng = wl.coeffs.shape[0]
    uniq = {}          # unique group contents (make_vardct(unique_groups=24) reuses them round robin)
    which = []
    for g in range(ng):
        key = g % 24
        if key not in uniq or not np.array_equal(wl.coeffs[g], wl.coeffs[uniq[key]]):
            uniq[key] = g
        which.append(uniq[key])
    base = {g: wl.coeffs[g] for g in sorted(set(which))}
    rng = np.random.default_rng(seed + 600)

    def densify(c, factor):
        c = c.copy()
        nz = c != 0
        if factor < 1:
            c[nz & (rng.random(c.shape) >= factor)] = 0
            return c
        p_new = min(1.0, nz.mean() * (factor - 1) / max(1e-9, 1 - nz.mean()))
        add = ~nz & (rng.random(c.shape) < p_new)
        mag = 1 + rng.geometric(0.5, size=c.shape)
        c[add] = (mag * rng.choice([-1, 1], size=c.shape))[add]
        return c

    def outlier_group(c):
        """one coefficient of the group at +-2000 ... 30000"""
        c = c.copy()
        nz = np.flatnonzero(c.reshape(-1))
        c.reshape(-1)[rng.choice(nz)] = int(rng.integers(2000, 30001)) * int(rng.choice([-1, 1]))
        return c

    variants = [("d1_clean", lambda c: c, None),
                ("outliers_1e-5_of_entries_2000_to_30000", None, None),
                ("density_x0.5", lambda c: densify(c, 0.5), None),
                ("density_x2", lambda c: densify(c, 2.0), None),
                ("density_x4", lambda c: densify(c, 4.0), None),
                ("one_group_dense_slab", lambda c: c, 0)]
    NE = len(ectx)

    def begin(c):
        c.frame_begin(synth.apply_opts(c.default_params(size, size), wl))
        c.set_dequant_tables(wl.tables)
        c.set_lf_quantized(*wl.lf_q)
        c.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)

    def resident_ms():
        for i in range(6):
            ectx[i % NE].frame_run()
        for c in ectx:
            c.sync()
        t_warm = time.perf_counter()
        while time.perf_counter() - t_warm < 0.05:
            for i in range(10):
                ectx[i % NE].frame_run()
            for c in ectx:
                c.sync()
        out = []
        for _ in range(max(1, reps)):
            t0 = time.perf_counter()
            for i in range(steps):
                ectx[i % NE].frame_run()
            for c in ectx:
                c.sync()
            out.append((time.perf_counter() - t0) * 1e3 / steps)
        return sorted(out)[(len(out) - 1) // 2]

    def k1_ms(c0):
        c0.kernel_timing_reset(); c0.kernel_timing(True)
        for _ in range(10):
            c0.frame_run()
        c0.sync()
        kt = c0.kernel_times(); c0.kernel_timing(False)
        return {k: round(v[0] / 10, 4) for k, v in kt.items()}

    res = {}
    for name, fn, dense_group in variants:
        var = {g: np.ascontiguousarray((fn or (lambda c: c))(c), dtype=np.int32) for g, c in base.items()}
        packed = {g: jl.host_pack_slots(c, group_id=0) for g, c in var.items()}
        assert all(len(q[3]) == 0 for q in packed.values()), "the packer split every value: nothing in `wide`"
        slotted = [g for g in range(ng) if g != dense_group]
        per_group = {g: packed[which[g]] for g in slotted}
        dense_of = lambda g: var[which[g]]
        if fn is None:
            # 1e-5 of the FRAME's entries: that many groups (picked at random) get ONE out-of-range coefficient each
            total = sum(int(per_group[g][2].sum()) for g in slotted)
            hit = rng.choice(slotted, size=max(1, int(round(total * 1e-5))), replace=False)
            changed = {int(g): np.ascontiguousarray(outlier_group(var[which[g]]), dtype=np.int32) for g in hit}
            for g, c in changed.items():
                per_group[g] = jl.host_pack_slots(c, group_id=0)
                assert len(per_group[g][3]) == 0
            dense_of = lambda g: changed.get(g, var[which[g]])
        ents = np.concatenate([per_group[g][0] for g in slotted])
        cnts = np.concatenate([per_group[g][1].reshape(-1) for g in slotted])
        ns = np.concatenate([per_group[g][2] for g in slotted])
        ids = np.asarray(slotted, dtype=np.uint32)
        for c in ectx:
            begin(c)
            c.submit_groups_slots(ids, ents, cnts, ns, None)
            if dense_group is not None:
                c.submit_group(dense_group, dense_of(dense_group))
            c.slot_wait(0)
            c.frame_run()
        for c in ectx:
            c.sync()
        slots_ms = resident_ms()
        kt = k1_ms(ectx[0])
        cnt = ectx[0].k1_counters()
        nb = sum(cnt["batches"].values())
        fb = sum(cnt["fallback_batches"].values())
        leg = {"slots_resident_ms_per_frame": round(slots_ms, 4), "k1_ms": kt.get("k1_vardct"),
               "entries_per_frame": int(sum(int(per_group[g][2].sum()) for g in slotted)),
               "nonzero_share_of_coefficients": round(float(np.mean([np.mean(var[g] != 0) for g in var])), 4),
               "fallback_share_of_batches": round(fb / max(1, nb), 4),
               "fallback_batches_by_class": {k: v for k, v in cnt["fallback_batches"].items() if v},
               "dense_route_varblocks": int(sum(cnt["dense_route_varblocks"].values())),
               "other_kernels_ms": {k: v for k, v in kt.items() if k.startswith("k_")} or None}
        # the same coefficients resident as dense slabs
        for c in ectx:
            begin(c)
            for g in range(ng):
                c.submit_group(g, dense_of(g))
            c.slot_wait(0)
            c.frame_run()
        for c in ectx:
            c.sync()
        leg["dense_resident_ms_per_frame"] = round(resident_ms(), 4)
        leg["dense_k1_ms"] = k1_ms(ectx[0]).get("k1_vardct")
        leg["slots_vs_dense"] = round(leg["slots_resident_ms_per_frame"] / leg["dense_resident_ms_per_frame"], 4)
        res[name] = leg
    clean = res["d1_clean"]["slots_resident_ms_per_frame"]
    for name, leg in res.items():
        leg["vs_clean_slots_frame"] = round(leg["slots_resident_ms_per_frame"] / clean, 4)
    # ---- what producing the form costs on the host: jxlh_host_pack_slots on every group of the
    # frame (dense i32 slab -> entries + slot counts), `cores` threads (ctypes releases the GIL), and on one thread
    # One C call per thread for its share of the frame (jxlh_host_pack_slots_many; ctypes releases the GIL): round 6's
    # first figure (37 ms on 16 cores) called jxlh_host_pack_slots once per group from Python threads and mostly
    # measured the interpreter lock around 1024 wrapper calls -- kept beside it as `per_group_python_calls_ms`.
    per = -(-ng // cores)
    shares = [(t * per, min(ng, (t + 1) * per)) for t in range(cores) if t * per < ng]
    big = max(g1 - g0 for g0, g1 in shares)
    ents_many = [np.empty(big * (3 * 65536 + 65536), np.uint16) for _ in shares]
    cnt_many = [np.empty((big, 3, 1024), np.uint8) for _ in shares]
    n_many = [np.zeros((big, 3), np.uint32) for _ in shares]
    ents_buf = [np.empty(3 * 65536 + 65536, np.uint16) for _ in range(cores)]
    cnt_buf = [np.empty((3, 1024), np.uint8) for _ in range(cores)]

    def pack_share(t, g0, g1):
        jl.host_pack_slots_many(wl.coeffs[g0:g1], np.arange(g0, g1, dtype=np.uint32), entries=ents_many[t],
                                slot_counts=cnt_many[t][:g1 - g0], n=n_many[t][:g1 - g0])

    def pack_range(t, g0, g1):
        for g in range(g0, g1):
            jl.host_pack_slots(wl.coeffs[g], group_id=g, entries=ents_buf[t], slot_counts=cnt_buf[t])

    def pack_frame(nthreads, fn):
        sh = shares if nthreads > 1 else [(0, ng)]
        t0 = time.perf_counter()
        if nthreads == 1:
            for g0 in range(0, ng, big):  # (the one-thread run reuses thread 0's buffers share by share)
                fn(0, g0, min(ng, g0 + big))
        else:
            with ThreadPoolExecutor(nthreads) as ex:
                list(ex.map(lambda a: fn(a[0], a[1][0], a[1][1]), enumerate(sh)))
        return (time.perf_counter() - t0) * 1e3

    has_many = hasattr(jl.load(), "jxlh_host_pack_slots_many")
    fn = pack_share if has_many else pack_range
    pack_frame(cores, fn)
    allc = sorted(pack_frame(cores, fn) for _ in range(3))[1]
    one = pack_frame(1, fn)
    res["host_pack"] = {"host_pack_ms_per_frame": round(allc, 2), "cores": cores, "one_thread_ms_per_frame": round(one, 1),
                        "per_group_python_calls_ms": round(sorted(pack_frame(cores, pack_range) for _ in range(3))[1], 2),
                        "what": "jxlh_host_pack_slots_many (C; AVX2 zero scan where the host has it, SSE2 otherwise), one call per "
                                "thread for its share of the frame's %d dense i32 group slabs (805 MB read) -> u16 entries + u8 "
                                "slot counts; a decoder that appends from its entropy loop (jxlh_slot_writer_*) pays per "
                                "coefficient instead of per slab byte" % ng}
    return res


def usable_cores():
    """Host cores this process may actually use: the affinity mask capped by the container's CPU quota
    (cgroup v2 cpu.max / v1 cfs quota).  os.cpu_count() reports the machine (256 on the GPU boxes) while the
    container is throttled to its quota (16 there): 256 threads then time-slice 16 cores' worth of CPU time."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, -(-q // per)))
        except (OSError, ValueError):
            pass
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--placement-trials", type=int, default=12,
                    help="candidate buffer placements a VarDCT context picks from at setup (jxlh_ctx_tune_placement); 1 = off")
    ap.add_argument("--size", type=int, default=8192)
    ap.add_argument("--cpu-sample", type=int, default=8192)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-strong", action="store_true", help="N > 1: skip the sharded-frame (strong scaling) leg")
    ap.add_argument("--strong-at-1", action="store_true",
                    help="run the sharded-frame leg with a single rank too (validation of the RCCL path on a 1-GPU box; "
                         "needs a torch.distributed.run launch)")
    ap.add_argument("--no-active", action="store_true",
                    help="skip the all-blocks-filtered EPF population (profiling runs: one population per kernel name)")
    ap.add_argument("--no-e2e", action="store_true",
                    help="skip the PCIe-inclusive legs (pinned host coefficients -> finished planes)")
    ap.add_argument("--strip", action="store_true",
                    help="also run the strip-kernel A/B block (roofline.strip_kernel; opt-in since round 6: a closed negative result)")
    ap.add_argument("--no-sweep", action="store_true",
                    help="skip e2e_pcie_inclusive.slot_form_content_sweep (outliers / density / one dense group on the slot form)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the `secondary` block (BASELINE configs 2, 4 and 5, one frame in flight each)")
    ap.add_argument("--reps", type=int, default=5,
                    help="repetitions of the timed region of K steps: `value` / `ms_per_step` are the MEDIAN repetition, "
                         "min and max are reported beside it (boxes and runs differ by several per cent)")
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--inflight", type=int, default=2,
                    help="frames in flight per GPU (contexts with their own stream and buffers, used round-robin)")
    ap.add_argument("--epf", default="spec", choices=["spec", "active", "passthrough"],
                    help="EPF sigma population: spec = SURVEY 8(d) draws (raw_quant U[2,16], sharpness U{0..7}: 93%% of "
                         "blocks fall below MIN_SIGMA and pass through); active = every block filtered "
                         "(sharpness 7); passthrough = none")
    ap.add_argument("--epf-iters", type=int, default=2, choices=[0, 1, 2, 3],
                    help="EPF iterations (config 3: 2; config 2: 0; 3 adds EPF0 and uses the per-stage filter kernels)")
    ap.add_argument("--mix", default="d1", choices=["d1", "dct8", "all"],
                    help="transform-type mix of the synthetic frame (d1 = BASELINE config 3)")
    args = ap.parse_args()
    PLACEMENT_TRIALS[0] = max(1, args.placement_trials)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: one rank per GPU under torch.distributed.run (what the driver's command
        # line does explicitly).  Never a silent single-rank measurement labelled N.
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            sys.exit(f"bench.py --gpus {args.gpus}: {have} HIP device(s) visible on this box "
                     f"(torch.cuda.device_count()); need {args.gpus} -- refusing to measure fewer ranks than asked for")
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)

    import numpy as np
    import torch
    import jxl_rs_amd
    from jxl_rs_amd import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run (also with a single rank)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        try:  # device_id: no guessing of the rank -> GPU mapping (a heterogeneous guess can hang the first collective)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
        except TypeError:
            dist.init_process_group("nccl", rank=rank, world_size=world)
    n_gpus = max(world, 1)
    if n_gpus != args.gpus:
        sys.exit(f"bench.py --gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                 f"--nproc-per-node {args.gpus} (or run `python bench.py --gpus {args.gpus}` without a launcher)")

    size = args.size
    # ---- synthetic frame (config 3): d1-like type mix DCT8..32, CfL, LF smoothing, Gaborish, EPF x2
    t0 = time.time()
    mix = {"d1": synth.MIX_D1, "dct8": synth.MIX_DCT8, "all": synth.MIX_ALL}[args.mix]
    wl = synth.make_vardct(size, size, mix=mix, seed=args.seed + rank, unique_groups=24,
                           epf_iters=args.epf_iters, gab=True, lf_smoothing=True)
    gen_s = time.time() - t0
    if args.epf == "active":
        wl.epf_map[:] = 7
        wl.raw_quant[:] = np.minimum(wl.raw_quant, 4)
    elif args.epf == "passthrough":
        wl.epf_map[:] = 0
    ctxs = []
    h2d_s = 0.0
    for _ in range(max(1, args.inflight)):
        c = new_context(jxl_rs_amd, local_rank, tag=f"headline {len(ctxs)}")
        params = synth.apply_opts(c.default_params(size, size), wl)
        c.frame_begin(params)
        c.set_dequant_tables(wl.tables)
        c.set_lf_quantized(*wl.lf_q)
        c.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
        t0 = time.time()
        for g in range(wl.coeffs.shape[0]):
            c.submit_group(g, wl.coeffs[g])
        c.slot_wait(0)
        h2d_s = time.time() - t0
        ctxs.append(c)
    ctx = ctxs[0]
    step_no = [0]

    ygroups = wl.ygroups

    def step():
        ctxs[step_no[0] % len(ctxs)].frame_run(0, ygroups)
        step_no[0] += 1

    def sync_all():
        for c in ctxs:
            c.sync()

    def barrier():
        if dist is not None:
            dist.barrier()
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    def max_over_ranks(seconds):
        if dist is None:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ---- weak leg (N = 1: THE measurement): K steps of whole frames, one frame per rank; the timed region is
    # repeated --reps times (barrier + synchronize on both sides of each, max over ranks of each) and the MEDIAN
    # repetition is what `value` / `ms_per_step` report
    for _ in range(args.warmup):
        step()
    sync_all()
    rep_ms, rep_ev = [], []
    for _ in range(max(1, args.reps)):
        barrier()
        t_wall0 = time.perf_counter()
        ctx.timer_start()
        for _ in range(args.steps):
            step()
        ev = ctx.timer_stop()
        sync_all()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        wall_s = time.perf_counter() - t_wall0
        barrier()
        rep_ms.append(max_over_ranks(wall_s) * 1e3 / args.steps)
        rep_ev.append(ev)
    order = sorted(range(len(rep_ms)), key=lambda i: rep_ms[i])
    mid = order[(len(order) - 1) // 2]
    ms_per_step = rep_ms[mid]
    ev_ms = rep_ev[mid]
    value = size * size * n_gpus / 1e6 / (ms_per_step / 1e3)

    chain_bytes, chain_file = chain_traffic("8192 d1") if size == 8192 and args.mix == "d1" else (None, None)

    def result_line(strong, strong_modular, roofline, cpu, e2e, secondary):
        return {
            "metric": "megapixels/sec decoded (8K VarDCT d1 reconstruction: dequant+CfL+IDCT+LF smoothing+Gaborish+EPF)",
            "value": round(value, 1), "unit": "MP/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{size}x{size} VarDCT {args.mix} mix full pipeline (CfL, LF smoothing, "
                                   f"Gaborish, EPF iters={args.epf_iters}), inputs HBM-resident",
                       "groups": int(wl.coeffs.shape[0]),
                       "sharding": "independent frames per GPU, no collective (strong_scaling: one frame in bands of "
                                   "group rows, halo exchange + RCCL all-gather)" if world > 1 else "single GPU",
                       "frames_in_flight_per_gpu": max(1, args.inflight), "epf_population": args.epf},
            "strong_scaling": strong, "strong_scaling_modular": strong_modular,
            "hip_event_ms_per_step_rank0": round(ev_ms / args.steps, 4),
            "repetitions": {"n": len(rep_ms), "of_steps": args.steps, "reported": "median",
                            "ms_per_step": [round(v, 4) for v in rep_ms], "min_ms_per_step": round(min(rep_ms), 4),
                            "max_ms_per_step": round(max(rep_ms), 4),
                            "value_at_min_ms": round(size * size * n_gpus / 1e6 / (min(rep_ms) / 1e3), 1),
                            "value_at_max_ms": round(size * size * n_gpus / 1e6 / (max(rep_ms) / 1e3), 1)},
            "setup": {"host_generate_s": round(gen_s, 2), "h2d_coeffs_s": round(h2d_s, 2),
                      "placement_trials": PLACEMENT_TRIALS[0], "placement": placement_summary(),
                      "placement_note": "jxlh_ctx_tune_placement: each VarDCT context's large buffers are the best-rated of "
                                        "`placement_trials` candidate allocations (two byte-mover probes, ms = [k1-like, "
                                        "filter-like]); setup only, --placement-trials 1 turns it off"},
            "roofline": roofline, "cpu_baseline": cpu, "e2e_pcie_inclusive": e2e, "secondary": secondary,
            # what makes rounds on different boxes comparable: the box's own copy rate (float4 read + write kernel,
            # frame-sized), the bytes the chain moved by counters (committed profile of this command), and the rate
            # the pipelined step moved them at as a fraction of that copy rate
            "copy_ceiling_GBs": (roofline or {}).get("copy_ceiling_GBs"),
            # the bytes the chain moved by counters (the newest committed PMC profile of this command: counters need
            # their own profiler run) over the pipelined step time, against the 8 TB/s peak and against the 6.3 TB/s the
            # guide gives as achievable (MI355X_MICROARCH.md); round 5's "fraction of the copy ceiling" read 1.02 and is gone
            "chain_counter_traffic_bytes": chain_bytes, "chain_counter_traffic_source": chain_file,
            "chain_counter_frac_of_8TBs": (round(chain_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if chain_bytes else None),
            "chain_counter_frac_of_6p3TBs": (round(chain_bytes / (ms_per_step * 1e-3) / 1e9 / 6300.0, 4) if chain_bytes else None),
            # the two kernels of the chain side by side (HIP events, one frame in flight), and the filters on the
            # population where every block is filtered
            "k1": (roofline or {}).get("k1_summary"), "filters": (roofline or {}).get("filters_summary"),
            "filters_all_blocks_filtered": (roofline or {}).get("filters_all_active_summary"),
            "one_frame_in_flight": (roofline or {}).get("one_frame_in_flight"),
        }

    # The sharded (strong) legs below are the only part of an N > 1 run in which ranks wait for each other inside
    # collectives; the weak line is measured by now.  If they have not come back after JXLH_BENCH_STRONG_TIMEOUT_S
    # seconds (a rank that failed, a transport that hangs), rank 0 prints the weak line with the legs marked as timed
    # out and every rank leaves -- the driver gets its line instead of a hung job.
    watchdog = None
    if (world > 1 or (args.strong_at_1 and dist is not None)) and not args.no_strong:
        import threading
        limit_s = float(os.environ.get("JXLH_BENCH_STRONG_TIMEOUT_S", "240"))

        def give_up():
            if rank == 0:
                msg = {"error": f"no result after {limit_s:g} s (JXLH_BENCH_STRONG_TIMEOUT_S); weak line printed without it"}
                print(json.dumps(result_line(msg, msg, None, None, None, None)), flush=True)
            os._exit(0)

        watchdog = threading.Timer(limit_s, give_up)
        watchdog.daemon = True
        watchdog.start()

    # ---- strong leg (N > 1): one frame sharded by bands of group rows, halo exchange + all-gather in the timed region
    strong = None
    if (world > 1 or (args.strong_at_1 and dist is not None)) and not args.no_strong:
        try:
            from jxl_rs_amd import lib as jl
            swl = wl if rank == 0 else synth.make_vardct(size, size, mix=mix, seed=args.seed, unique_groups=24,
                                                         epf_iters=args.epf_iters, gab=True, lf_smoothing=True)
            if args.epf == "active":
                swl.epf_map[:] = 7
                swl.raw_quant[:] = np.minimum(swl.raw_quant, 4)
            elif args.epf == "passthrough":
                swl.epf_map[:] = 0
            box = [jl.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            setup_error = None
            try:
                sctx = new_context(jxl_rs_amd, local_rank, tag="strong scaling")
                sctx.comm_init(box[0], rank, world)      # before frame_begin: it sizes the planes for the gather
                sctx.frame_begin(synth.apply_opts(sctx.default_params(size, size), swl))
                sctx.set_dequant_tables(swl.tables)
                sctx.set_lf_quantized(*swl.lf_q)        # LF / maps / tables are replicated (20 MB)
                sctx.set_hf_meta(swl.transform_map, swl.raw_quant, swl.epf_map, swl.ytox, swl.ytob)
                _, _, row0, row1 = sctx.comm_band()
                for g in range(row0 * swl.xgroups, row1 * swl.xgroups):   # only the own band's coefficient groups
                    sctx.submit_group(g, swl.coeffs[g])
                sctx.slot_wait(0)
            except Exception as e:
                setup_error = f"{type(e).__name__}: {e}"
            # all ranks take the sharded leg or none does: a rank that failed to set up must not leave the others
            # waiting in a collective
            okf = torch.tensor([0 if setup_error else 1], dtype=torch.int32,
                               device=f"cuda:{local_rank}" if torch.cuda.is_available() else "cpu")
            dist.all_reduce(okf, op=dist.ReduceOp.MIN)
            if int(okf.item()) == 0:
                raise RuntimeError(setup_error or "another rank failed to set up the sharded frame")

            def sstep():
                sctx.frame_run_sharded()
                sctx.frame_allgather()

            for _ in range(max(1, args.warmup)):
                sstep()
            sctx.sync()
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                sstep()
            sctx.sync()
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            s_wall = time.perf_counter() - t0
            barrier()
            s_wall = max_over_ranks(s_wall)
            # ---- the second strong form (round 6): gather the CONVERTED image (interleaved 8-bit sRGB, a quarter of the
            # bytes) instead of the f32 planes -- jxlh_frame_allgather_output
            o_wall = None
            try:
                from jxl_rs_amd.lib import DeviceArray
                kk = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_kat.json")))["output_stage"]
                bias = np.float32(kk["opsin_bias"])
                xyb_params = np.concatenate([np.asarray(kk["opsin_inverse_matrix"], np.float32),
                                             np.full(3, np.cbrt(bias), np.float32), np.full(3, bias, np.float32),
                                             np.ones(1, np.float32)])
                desc = jxl_rs_amd.Context.output_desc(xyb_params=xyb_params, bits=8, channels=3)
                per = (ygroups + world - 1) // world
                img = DeviceArray(nbytes=world * per * 256 * size * 3, device=local_rank)

                def ostep():
                    sctx.frame_run_sharded()
                    sctx.frame_allgather_output(desc, img.ptr, size * 3)

                for _ in range(2):
                    ostep()
                sctx.sync()
                barrier()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    ostep()
                sctx.sync()
                o_wall = max_over_ranks(time.perf_counter() - t0)
                barrier()
                img.free()
            except Exception as e:
                o_wall = f"{type(e).__name__}: {e}"
            # without the gather: what the sharded compute alone costs (band K1 + exchange + filters)
            for _ in range(2):
                sctx.frame_run_sharded()
            sctx.sync()
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                sctx.frame_run_sharded()
            sctx.sync()
            c_wall = max_over_ranks(time.perf_counter() - t0)
            barrier()
            # every rank must hold the same, complete frame; rank 0 also holds the single-GPU result of this very frame
            sstep()
            sctx.sync()
            got = sctx.read_planes()
            digest = int(sum(int(np.sum(pl.view(np.uint32), dtype=np.uint64)) for pl in got) & 0x7FFFFFFFFFFFFFFF)
            dd = torch.tensor([digest, -digest], dtype=torch.int64, device=f"cuda:{local_rank}")
            dist.all_reduce(dd, op=dist.ReduceOp.MAX)
            same_everywhere = int(dd[0].item()) == -int(dd[1].item())
            matches_single = None
            if rank == 0:
                ctx.frame_run(0, ygroups)
                ctx.sync()
                ref = ctx.read_planes()
                matches_single = all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(got, ref))
                del ref
            del got
            s_ms = s_wall * 1e3 / args.steps
            strong = {"value": round(size * size / 1e6 / (s_ms / 1e3), 1), "unit": "MP/s", "ms_per_step": round(s_ms, 4),
                      "scaling": "strong", "n_gpus": world,
                      "ms_per_step_without_gather": round(c_wall * 1e3 / args.steps, 4),
                      "allgather_MB_total": round(3 * size * size * 4 / 1e6, 1),
                      "output_gather_rgb8": ({"ms_per_step": round(o_wall * 1e3 / args.steps, 4),
                                              "value": round(size * size / 1e6 / (o_wall / args.steps), 1), "unit": "MP/s",
                                              "allgather_MB_total": round(3 * size * size / 1e6, 1),
                                              "what": "the same sharded frame, bands converted to interleaved 8-bit sRGB on their "
                                                      "rank and gathered (jxlh_frame_allgather_output)"}
                                             if isinstance(o_wall, float) else {"error": o_wall}),
                      "halo_exchange_KB_per_edge": round(3 * 8 * swl.xblocks * 8 * 4 / 1e3, 1),
                      "band_group_rows_per_rank": (ygroups + world - 1) // world,
                      "rccl_ranks_in_communicator": int(sctx.comm_band()[1]),
                      "frame_identical_on_all_ranks": bool(same_everywhere),
                      "frame_bit_equal_to_single_gpu_run": matches_single,
                      "what": "ONE frame: per rank K1 on its band of group rows, ncclSend/ncclRecv of the edge block rows, "
                              "fused filters on the band, in-place ncclAllGather of the 3 finished planes (library-owned "
                              "RCCL communicator); every rank ends with the whole frame"}
            barrier()  # every rank is done with the communicator before any rank tears it down
            sctx.comm_destroy()
            sctx.close()
        except Exception as e:  # the weak line must survive a failure of the sharded leg
            strong = {"error": f"{type(e).__name__}: {e}"}

    # ---- Modular across GPUs (BASELINE configs[3]: "Squeeze + RCT + Palette, 1 -> 8 GPU group shard + RCCL
    # all-gather"): the squeeze chain replicated on every rank (its recurrence is serial along whole lines), the RCT
    # and the palette expansion on the rank's own sample share, six in-place all-gathers over the library's communicator
    strong_modular = None
    if (world > 1 or (args.strong_at_1 and dist is not None)) and not args.no_strong:
        try:
            from jxl_rs_amd import lib as jl
            from jxl_rs_amd.modular import ModularChain
            box = [jl.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            mctx = jxl_rs_amd.Context(local_rank, n_slots=1)
            mctx.comm_init(box[0], rank, world)
            msize = 8192
            mch = ModularChain(mctx, msize, msize, seed=84, world=world)

            def mstep():
                mch.run_pipeline_rccl(rank)

            for _ in range(2):
                mstep()
            mctx.sync()
            barrier()
            t0 = time.perf_counter()
            for _ in range(max(3, args.steps // 2)):
                mstep()
            mctx.sync()
            m_wall = max_over_ranks(time.perf_counter() - t0)
            barrier()
            # every rank must hold the same planes
            planes, pal = mch.pipeline_result()
            digest = int(sum(int(np.sum(pl.view(np.uint32), dtype=np.uint64)) for pl in planes + pal) & 0x7FFFFFFFFFFFFFFF)
            dd = torch.tensor([digest, -digest], dtype=torch.int64, device=f"cuda:{local_rank}")
            dist.all_reduce(dd, op=dist.ReduceOp.MAX)
            m_ms = m_wall * 1e3 / max(3, args.steps // 2)
            strong_modular = {"value": round(msize * msize / 1e6 / (m_ms / 1e3), 1), "unit": "MP/s", "ms_per_step": round(m_ms, 4),
                              "scaling": "strong", "n_gpus": world, "rccl_ranks_in_communicator": world,
                              "planes_identical_on_all_ranks": int(dd[0].item()) == -int(dd[1].item()),
                              "allgather_MB_total": round(6 * msize * msize * 4 / 1e6, 1),
                              "what": f"{msize}x{msize} x 3 ch: default squeeze chain replicated per rank, YCoCg RCT and 256-colour "
                                      "palette on the rank's sample share, six in-place ncclAllGather (library-owned communicator)"}
            del planes, pal
            barrier()
            mch.free()
            mctx.comm_destroy()
            mctx.close()
        except Exception as e:
            strong_modular = {"error": f"{type(e).__name__}: {e}"}

    if watchdog is not None:
        watchdog.cancel()

    # ---- per-kernel HIP-event timing (separate steps; not part of the timed region)
    roofline = None
    if rank == 0:
        npx = size * size

        def kernel_table():
            # untimed frames first (80 ms of them): after the host-side work between two populations (numpy, set_hf_meta's
            # synchronous copies) the device has idled and its first tens of frames run 15-25 % slower than the steady
            # state -- round 4's "half-filtered slower than all-filtered" was the population measured cold
            # (profiles/r05_d_filter_populations.txt)
            t_warm = time.perf_counter()
            while time.perf_counter() - t_warm < 0.08:   # ~100 frames: the device's clocks are back up
                for _ in range(10):
                    ctx.frame_run(0, ygroups)
                ctx.sync()
            ctx.kernel_timing_reset()
            ctx.kernel_timing(True)
            nprof = max(3, min(args.steps, 10))
            for _ in range(nprof):
                ctx.frame_run(0, ygroups)
            ctx.sync()
            kt = ctx.kernel_times()
            ctx.kernel_timing(False)
            out = {}
            for name, (ms, n) in kt.items():
                k = {"ms_per_step": round(ms / nprof, 4), "launches_per_step": n // nprof}
                if name in ALGO_BYTES_PER_PX:  # each kernel against ITS OWN compulsory traffic
                    ab = ALGO_BYTES_PER_PX[name] * npx
                    k["algorithmic_bytes"] = int(ab)
                    k["achieved_GBs"] = round(ab / (ms / nprof * 1e-3) / 1e9, 1)
                    k["frac"] = round(k["achieved_GBs"] / HBM_PEAK_GBS, 4)
                out[name] = k
            return out

        kernels = kernel_table()
        # the population where EVERY 8x8 block is filtered (sharpness 7, raw_quant <= 4), same frame otherwise: the
        # spec draws leave 93 % of the blocks below EPF's MIN_SIGMA (they pass through, like epf1.rs:72-78)
        active = None
        if args.epf == "spec" and args.epf_iters > 0 and not args.no_active:
            ctx.set_hf_meta(wl.transform_map, np.minimum(wl.raw_quant, 4), np.full_like(wl.epf_map, 7), wl.ytox, wl.ytob)
            ak = kernel_table()
            ctx.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
            ctx.frame_run(0, ygroups)
            ctx.sync()
            active = {k: ak[k] for k in ak if k in ALGO_BYTES_PER_PX}
            active["sum_of_kernels_ms"] = round(sum(v["ms_per_step"] for v in ak.values()), 4)
        # ... and the population in between: a random half of the blocks filtered (real d1 content lies between the spec
        # draw and every-block-filtered)
        half = None
        if args.epf == "spec" and args.epf_iters > 0 and not args.no_active:
            pick = np.random.default_rng(args.seed + 50).random(wl.epf_map.shape) < 0.5
            ctx.set_hf_meta(wl.transform_map, np.where(pick, np.minimum(wl.raw_quant, 4), wl.raw_quant),
                            np.where(pick, 7, 0).astype(wl.epf_map.dtype), wl.ytox, wl.ytob)
            hk = kernel_table()
            ctx.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
            ctx.frame_run(0, ygroups)
            ctx.sync()
            half = {k: hk[k] for k in hk if k in ALGO_BYTES_PER_PX}
            half["sum_of_kernels_ms"] = round(sum(v["ms_per_step"] for v in hk.values()), 4)
            half["blocks_filtered"] = round(float(pick.mean()), 3)
        # ---- the strip kernel (JXLH_FRAME_STRIP, k_strip.hip): the whole chain as ONE persistent launch with no
        # intermediate planes in HBM, on this frame (whatever its tilings let the strip kernel transform itself) and on
        # the same workload with every varblock aligned to its own size (SURVEY 8(d)'s tiling law, what libjxl's encoder
        # emits: every tile is the strip kernel's).  Opt-in, bit-identical, and SLOWER than the two-kernel path on MI355X
        # today (both forms are bound by instruction issue): reported, never `value`.
        strip = None
        if args.strip and args.epf_iters <= 2:
            from jxl_rs_amd import lib as jl
            strip = {}
            for tag, swl in (("this_frame", wl), ("aligned_tilings", None)):
                try:
                    if swl is None:
                        swl = synth.make_vardct(size, size, mix=mix, seed=args.seed, unique_groups=24,
                                                epf_iters=args.epf_iters, gab=True, lf_smoothing=True, aligned=True)
                        if args.epf == "active":
                            swl.epf_map[:] = 7
                            swl.raw_quant[:] = np.minimum(swl.raw_quant, 4)
                    res = {}
                    planes_of = {}
                    for name, flags in (("two_kernels", 0), ("strip", jl.FRAME_STRIP)):
                        sc = jxl_rs_amd.Context(local_rank, n_slots=1)
                        sp = synth.apply_opts(sc.default_params(size, size), swl)
                        sp.flags = flags
                        sc.frame_begin(sp)
                        sc.set_dequant_tables(swl.tables)
                        sc.set_lf_quantized(*swl.lf_q)
                        sc.set_hf_meta(swl.transform_map, swl.raw_quant, swl.epf_map, swl.ytox, swl.ytob)
                        for g in range(swl.coeffs.shape[0]):
                            sc.submit_group(g, swl.coeffs[g])
                        sc.slot_wait(0)
                        for _ in range(3):
                            sc.frame_run()
                        sc.sync()
                        t0 = time.perf_counter()
                        n = max(5, min(args.steps, 20))
                        for _ in range(n):
                            sc.frame_run()
                        sc.sync()
                        res[name] = {"ms_per_frame": round((time.perf_counter() - t0) * 1e3 / n, 4), "frames_in_flight": 1}
                        if name == "strip":
                            ran, tiles, by_class = sc.frame_path()
                            res[name].update({"strip_kernel_ran": ran, "tiles": tiles, "tiles_left_to_class_kernels": by_class})
                        planes_of[name] = [int(np.sum(pl.view(np.uint32), dtype=np.uint64)) for pl in sc.read_planes()]
                        sc.close()
                    res["bit_identical_checksums"] = planes_of["two_kernels"] == planes_of["strip"]
                    strip[tag] = res
                except Exception as e:
                    strip[tag] = {"error": f"{type(e).__name__}: {e}"}
            strip["counters"] = "profiles/r04_a_strip_pmc_aligned.txt, r04_a_strip_ablation.txt"
        cand = {k: v for k, v in kernels.items() if k in ALGO_BYTES_PER_PX}
        if cand:
            dom = max(cand, key=lambda k: cand[k]["ms_per_step"])
            traffic, traffic_file = (None, None)
            if size == 8192 and args.mix == "d1":
                traffic, traffic_file = resolve_traffic(dom)
            chain_ms = sum(v["ms_per_step"] for v in kernels.values())
            ideal = FUSED_IDEAL_BYTES_PER_PX * npx
            roofline = {"kernel": dom, "bound": "hbm", "achieved": cand[dom]["achieved_GBs"], "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": cand[dom]["frac"], "traffic": traffic,
                        "traffic_source": traffic_file,
                        "algorithmic_bytes_per_launch": cand[dom]["algorithmic_bytes"],
                        "avg_launch_ms": cand[dom]["ms_per_step"], "all_kernels_ms_per_step": kernels,
                        # the whole IDCT+EPF stage against north_star's numerator: what ONE fused kernel would move
                        "chain_vs_fused_ideal": {
                            "algorithmic_bytes": int(ideal), "bytes_per_px": FUSED_IDEAL_BYTES_PER_PX,
                            "sum_of_kernels_ms": round(chain_ms, 4),
                            "achieved_GBs": round(ideal / (chain_ms * 1e-3) / 1e9, 1),
                            "frac": round(ideal / (chain_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                            "ms_per_step_pipelined": round(ms_per_step, 4),
                            "frac_pipelined": round(ideal / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                        "epf_population_all_active": active, "epf_population_half_active": half,
                        "strip_kernel": strip}
        # ---- measured copy ceilings: a float4 read+write kernel, frame-sized (one plane set in, one
        # out: the traffic shape of K1 and of the filters) and Infinity-Cache-resident
        if roofline is not None:
            big = ctx.probe_copy_bandwidth(3 * npx * 4, reps=10)
            small = ctx.probe_copy_bandwidth(64 << 20, reps=50)
            roofline["copy_ceiling_GBs"] = round(big, 1)
            roofline["copy_ceiling"] = {"frame_sized": {"bytes_each_way": 3 * npx * 4, "GBs": round(big, 1)},
                                        "infinity_cache_resident": {"bytes_each_way": 64 << 20, "GBs": round(small, 1)},
                                        "kernel": "float4 grid-stride copy, the better of plain and nt accesses (jxlh_probe_copy_bandwidth)"}
            def summary(tab, name):
                k = (tab or {}).get(name)
                return None if not k else {"ms": k["ms_per_step"], "frac": k.get("frac"), "achieved_GBs": k.get("achieved_GBs"),
                                           "algorithmic_bytes": k.get("algorithmic_bytes")}
            roofline["k1_summary"] = summary(kernels, "k1_vardct")
            roofline["filters_summary"] = summary(kernels, "k23_fused_filters")
            roofline["filters_all_active_summary"] = summary(active, "k23_fused_filters")
            # ONE frame in flight: wall clock per frame (sync on both sides of K frames on one context) against the
            # sum of its kernels' HIP-event times -- the difference is what the second context of the headline hides
            for _ in range(5):
                ctx.frame_run(0, ygroups)
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                ctx.frame_run(0, ygroups)
            ctx.sync()
            one_ms = (time.perf_counter() - t0) * 1e3 / args.steps
            chain_ms_ = sum(v["ms_per_step"] for v in kernels.values())
            roofline["one_frame_in_flight"] = {"wall_ms_per_frame": round(one_ms, 4), "sum_of_kernels_ms": round(chain_ms_, 4),
                                               "launches_per_frame": int(sum(v["launches_per_step"] for v in kernels.values())),
                                               "gap_ms": round(one_ms - chain_ms_, 4)}

    # ---- CPU baseline: the oracle (C port of the reference path) on the same frame size
    cpu = None
    if rank == 0 and n_gpus == 1 and not args.no_cpu:  # reported baseline: single-GPU runs only
        from oracle.oracle import Oracle
        o = Oracle(fused=True)
        cs = min(args.cpu_sample, size)
        cwl = wl if cs == size else synth.make_vardct(cs, cs, mix=mix, seed=args.seed, unique_groups=24,
                                                      epf_iters=args.epf_iters)
        p = o.default_params(cs, cs)
        p.epf_iters = args.epf_iters
        lf = o.dequant_lf(p, *cwl.lf_q)
        cores = usable_cores()

        def cpu_run(c, nthreads, buffers):
            o.vardct_frame(p, c.coeffs, c.transform_map, c.raw_quant, c.epf_map, c.ytox, c.ytob, lf, c.tables,
                           num_threads=nthreads, buffers=buffers)
            return o.last_buffers

        bufs = cpu_run(cwl, cores, None)  # warm-up; allocates (and first-touches) the planes once
        reps, t0 = 0, time.perf_counter()
        while True:
            cpu_run(cwl, cores, bufs)
            reps += 1
            el = time.perf_counter() - t0
            if el > 12.0 or reps >= 20:
                break
        all_core = cs * cs * reps / 1e6 / el
        # one thread, on a 2048^2 frame of the same workload (a whole 8K frame would take ~15 s per repetition)
        s1 = min(2048, cs)
        wl1 = synth.make_vardct(s1, s1, mix=mix, seed=args.seed, unique_groups=24, epf_iters=args.epf_iters)
        p1 = o.default_params(s1, s1)
        p1.epf_iters = args.epf_iters
        lf1 = o.dequant_lf(p1, *wl1.lf_q)
        b1 = None
        t1 = []
        for _ in range(3):
            t0 = time.perf_counter()
            o.vardct_frame(p1, wl1.coeffs, wl1.transform_map, wl1.raw_quant, wl1.epf_map, wl1.ytox, wl1.ytob, lf1,
                           wl1.tables, num_threads=1, buffers=b1)
            b1 = o.last_buffers
            t1.append(time.perf_counter() - t0)
        one = s1 * s1 / 1e6 / min(t1[1:])
        cpu = {"value": round(all_core, 2), "unit": "MP/s", "cores": cores, "kind": "port",
               "sample": f"{reps} reps of the same {cs}x{cs} synthetic frame the GPU ran (same type mix / filters), C oracle "
                         f"-O3 x86-64-v3 (scalar restatement, not the reference's SIMD), pthreads over groups and row "
                         f"bands, output planes allocated and touched before the timed region",
               "one_thread": {"value": round(one, 2), "unit": "MP/s", "sample": f"best of 2 timed runs of a {s1}x{s1} frame"},
               "parallel_efficiency": round(all_core / (one * cores), 3),
               "machine_cpus": os.cpu_count(),
               "note": "cores = the affinity mask capped by the container's cgroup CPU quota (what the process can really "
                       "use; round 1 ran 256 threads on a 16-core quota); a thread pool is created per stage, adaptive LF "
                       "smoothing and the sigma map run on one thread, and the oracle is scalar C -- the reference itself "
                       "(Rust, SIMD, rayon) cannot be built in this image"}
        del bufs

    for c in ctxs:
        c.close()
    ctxs = []
    # ---- end-to-end legs: coefficients start in pinned HOST memory every frame.  Reported
    # beside `value`, never as `value`.  Two transports: the reference's dense i32 slabs
    # (jxlh_submit_group) and (position, value) pairs (jxlh_submit_groups_sparse).
    e2e = None
    if rank == 0 and n_gpus == 1 and not args.no_e2e and torch.cuda.is_available():
        e2e = {}
        ng = wl.coeffs.shape[0]
        # slot streams per context: the slot-bucketed legs (and the resident ones) run on contexts with ONE, the older
        # transports afterwards on contexts with JXLH_BENCH_SLOTS (2) -- see the switch in the leg loop
        nslots_legacy = int(os.environ.get('JXLH_BENCH_SLOTS', '2'))
        nslots = 1
        NE = 2  # frames in flight in the PCIe legs (3 contexts measured slower: their 9 streams alias on the
        #         runtime's few hardware queues and serialise)
        ectx = [new_context(jxl_rs_amd, local_rank, n_slots=nslots, tag=f"e2e {i}") for i in range(NE)]
        for c in ectx:
            c.frame_begin(synth.apply_opts(c.default_params(size, size), wl))
            c.set_dequant_tables(wl.tables)
            c.set_lf_quantized(*wl.lf_q)
            c.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
        # sparse form of every group (unique groups converted once), packed into ONE pinned buffer
        cache, runs, ns, total = {}, [], [], 0
        n_wide = 0
        for g in range(ng):
            key = g % 24  # make_vardct(unique_groups=24) reuses group contents round-robin (checked below)
            if key not in cache or not np.array_equal(wl.coeffs[g], wl.coeffs[cache[key][3]]):
                pr, n3, wd = synth.to_sparse(wl.coeffs[g])
                cache[key] = (pr, n3, wd, g)
            pr, n3, wd, _ = cache[key]
            runs.append(pr); ns.append(n3); total += len(pr); n_wide += len(wd)
        assert n_wide == 0, "synthetic d1 coefficients fit i16"
        pin_s, pin_s_addr = ectx[0].alloc_pinned(max(4, total * 4))
        pin_s.view(np.uint32)[:total] = np.concatenate(runs)
        ns = np.concatenate(ns).astype(np.uint32)
        offs = np.concatenate([[0], np.cumsum(ns.reshape(ng, 3).sum(axis=1))]).astype(np.int64)
        pin_d, pin_d_addr = ectx[0].alloc_pinned(wl.coeffs.nbytes)
        pin_d.view(np.int32)[:] = wl.coeffs.reshape(-1)
        ids = np.arange(ng, dtype=np.uint32)
        per_of = lambda: (ng + nslots - 1) // nslots  # groups per slot stream (nslots changes between the legs)
        # the 3-byte form (u16 positions + i8 values) of the same pairs: the synthetic d1 values fit 8 bits
        allp = np.concatenate(runs)
        assert ((allp >> 16).astype(np.uint16).view(np.int16).astype(np.int32).__abs__() < 128).all()
        pin_p, pin_p_addr = ectx[0].alloc_pinned(max(4, total * 2))
        pin_v, pin_v_addr = ectx[0].alloc_pinned(max(4, total))
        pin_p.view(np.uint16)[:total] = (allp & 0xFFFF).astype(np.uint16)
        pin_v.view(np.int8)[:total] = (allp >> 16).astype(np.uint16).view(np.int16).astype(np.int8)

        # the 2-byte form (round 4): 12-bit position inside a 4096-coefficient segment + value nibble, per-segment counts;
        # the values the nibble does not hold ([-8, 7]) in a 3-byte overflow list
        c4, e4, cnt4, p48, v48, n48 = {}, [], [], [], [], []
        for g in range(ng):
            key = g % 24
            if key not in c4 or c4[key][1] != cache[key][3]:
                c4[key] = (synth.to_sparse4(wl.coeffs[cache[key][3]]), cache[key][3])
            q4 = c4[key][0]
            assert len(q4[5]) == 0
            e4.append(q4[0]); cnt4.append(q4[1].reshape(-1)); p48.append(q4[2]); v48.append(q4[3]); n48.append(q4[4])
        off_e = np.concatenate([[0], np.cumsum([len(x) for x in e4])]).astype(np.int64)
        off_o = np.concatenate([[0], np.cumsum([len(x) for x in p48])]).astype(np.int64)
        tot_e, tot_o = int(off_e[-1]), int(off_o[-1])
        pin_e, pin_e_addr = ectx[0].alloc_pinned(max(4, tot_e * 2))
        pin_c, pin_c_addr = ectx[0].alloc_pinned(ng * 48 * 2)
        pin_op, pin_op_addr = ectx[0].alloc_pinned(max(4, tot_o * 2))
        pin_ov, pin_ov_addr = ectx[0].alloc_pinned(max(4, tot_o))
        pin_e.view(np.uint16)[:tot_e] = np.concatenate(e4)
        pin_c.view(np.uint16)[:] = np.concatenate(cnt4)
        if tot_o:
            pin_op.view(np.uint16)[:tot_o] = np.concatenate(p48)
            pin_ov.view(np.int8)[:tot_o] = np.concatenate(v48)
        n48 = np.concatenate(n48).astype(np.uint32)
        bytes4 = tot_e * 2 + ng * 48 * 2 + tot_o * 3

        def submit_sparse4(c):
            for sl in range(nslots):
                g0, g1 = sl * per_of(), min(ng, (sl + 1) * per_of())
                if g0 < g1:
                    c.submit_groups_sparse4(ids[g0:g1], pin_e_addr + int(off_e[g0]) * 2, pin_c_addr + g0 * 96,
                                            pin_op_addr + int(off_o[g0]) * 2, pin_ov_addr + int(off_o[g0]),
                                            n48[3 * g0:3 * g1], None, slot=sl)

        # the slot-bucketed 2-byte form (round 4): 6-bit position inside a 64-coefficient slot + 10-bit value, one u8 count
        # per slot; a frame that arrives entirely in this form is not sorted on the device
        cs_, es_, cnts_, ns_ = {}, [], [], []
        for g in range(ng):
            key = g % 24
            if key not in cs_ or cs_[key][1] != cache[key][3]:
                cs_[key] = (synth.to_slots(wl.coeffs[cache[key][3]]), cache[key][3])
            qs = cs_[key][0]
            assert len(qs[3]) == 0
            es_.append(qs[0]); cnts_.append(qs[1].reshape(-1)); ns_.append(qs[2])
        off_s = np.concatenate([[0], np.cumsum([len(x) for x in es_])]).astype(np.int64)
        tot_s = int(off_s[-1])
        pin_se, pin_se_addr = ectx[0].alloc_pinned(max(4, tot_s * 2))
        pin_sc, pin_sc_addr = ectx[0].alloc_pinned(ng * 3072)
        pin_se.view(np.uint16)[:tot_s] = np.concatenate(es_)
        pin_sc[:] = np.concatenate(cnts_)
        ns_ = np.concatenate(ns_).astype(np.uint32)
        bytes_slots = tot_s * 2 + ng * 3072

        def submit_slots(c, streams=None):
            """streams = slot streams the frame's groups are spread over (default: all the context has)"""
            k = streams or nslots
            per_k = (ng + k - 1) // k
            for sl in range(k):
                g0, g1 = sl * per_k, min(ng, (sl + 1) * per_k)
                if g0 < g1:
                    c.submit_groups_slots(ids[g0:g1], pin_se_addr + int(off_s[g0]) * 2, pin_sc_addr + g0 * 3072,
                                          ns_[3 * g0:3 * g1], None, slot=sl)

        # ... and with 12-bit entries packed two per three bytes (JXLH_GROUP_ENTRIES12: values in [-32, 31])
        from jxl_rs_amd import lib as jl_
        c12, e12, cnt12, n12 = {}, [], [], []
        for g in range(ng):
            key = g % 24
            if key not in c12 or c12[key][1] != cache[key][3]:
                c12[key] = (synth.to_slots(wl.coeffs[cache[key][3]], True), cache[key][3])
            q12 = c12[key][0]
            assert len(q12[3]) == 0
            e12.append(q12[0]); cnt12.append(q12[1].reshape(-1)); n12.append(q12[2])
        off12 = np.concatenate([[0], np.cumsum([len(x) for x in e12])]).astype(np.int64)  # bytes
        tot12 = int(off12[-1])
        pin_12, pin_12_addr = ectx[0].alloc_pinned(max(4, tot12))
        pin_12c, pin_12c_addr = ectx[0].alloc_pinned(ng * 3072)
        pin_12[:tot12] = np.concatenate(e12)
        pin_12c[:] = np.concatenate(cnt12)
        n12 = np.concatenate(n12).astype(np.uint32)
        bytes_12 = tot12 + ng * 3072

        def submit_slots12(c, streams=None):
            k = streams or nslots
            per_k = (ng + k - 1) // k
            for sl in range(k):
                g0, g1 = sl * per_k, min(ng, (sl + 1) * per_k)
                if g0 < g1:
                    c.submit_groups_slots(ids[g0:g1], pin_12_addr + int(off12[g0]), pin_12c_addr + g0 * 3072,
                                          n12[3 * g0:3 * g1], None, slot=sl, flags=jl_.GROUP_COMPLETE | jl_.GROUP_ENTRIES12)

        def submit_sparse(c):
            for sl in range(nslots):
                g0, g1 = sl * per_of(), min(ng, (sl + 1) * per_of())
                if g0 < g1:
                    c.submit_groups_sparse(ids[g0:g1], pin_s_addr + int(offs[g0]) * 4, ns[3 * g0:3 * g1], None, slot=sl)

        def submit_sparse8(c):
            for sl in range(nslots):
                g0, g1 = sl * per_of(), min(ng, (sl + 1) * per_of())
                if g0 < g1:
                    c.submit_groups_sparse8(ids[g0:g1], pin_p_addr + int(offs[g0]) * 2, pin_v_addr + int(offs[g0]),
                                            ns[3 * g0:3 * g1], None, slot=sl)

        def submit_dense(c):
            slab = 3 * 65536 * 4
            for g in range(ng):
                c.submit_group(g, pin_d_addr + g * slab, slot=g % nslots)

        # the same frame resident in HBM as pairs instead of dense slabs (K1 reads the bucketed pairs):
        # what the reconstruction kernels cost in the sparse-transport deployment, without PCIe
        def resident_leg():
            """frames already resident in their sparse form: the headline's pattern (warm-up, K frames round robin over
            the contexts, median of --reps repetitions; round 4 timed the first 20 frames after the buffers' first touch)"""
            for i in range(max(4, args.warmup)):
                ectx[i % NE].frame_run()
            for c in ectx:
                c.sync()
            reps = []
            for _ in range(max(1, args.reps)):
                t0 = time.perf_counter()
                for i in range(args.steps):
                    ectx[i % NE].frame_run()
                for c in ectx:
                    c.sync()
                reps.append((time.perf_counter() - t0) * 1e3 / args.steps)
            ms = sorted(reps)[(len(reps) - 1) // 2]
            return {"value": round(size * size / 1e6 / (ms / 1e3), 1), "unit": "MP/s", "ms_per_frame": round(ms, 4),
                    "frames": args.steps, "repetitions_ms": [round(v, 4) for v in reps], "reported": "median"}

        for c in ectx:
            submit_sparse(c); c.frame_run()
        for c in ectx:
            c.sync()
        e2e["pairs_resident_no_pcie"] = resident_leg()
        e2e["pairs_resident_no_pcie"]["form"] = ("bucketed {u16 pos; i16 val} pair words + slot tables (device sort): what "
                                                 "`sparse_resident_no_pcie` measured up to round 4")
        # ... and resident in the slot-bucketed 2-byte form exactly as jxlh_submit_groups_slots uploads it (round 5: the
        # transforms read the entries in place -- no unpack pass, no slot tables -- and dequantise only the positions
        # that have an entry): what the kernels of the PCIe-inclusive deployment cost, per kernel and by counters
        for c in ectx:
            c.frame_begin(synth.apply_opts(c.default_params(size, size), wl))
            c.set_dequant_tables(wl.tables)
            c.set_lf_quantized(*wl.lf_q)
            c.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
            submit_slots(c); c.frame_run()
        for c in ectx:
            c.sync()
        leg = resident_leg()
        leg["form"] = "u16 entries (pos6 | val10) + u8 slot counts, read in place"
        c0 = ectx[0]
        for _ in range(40):
            c0.frame_run()
        c0.sync()
        c0.kernel_timing_reset(); c0.kernel_timing(True)
        for _ in range(10):
            c0.frame_run()
        c0.sync()
        kt = c0.kernel_times(); c0.kernel_timing(False)
        leg["kernels_ms"] = {k: round(v[0] / 10, 4) for k, v in kt.items()}
        leg["sum_of_kernels_ms"] = round(sum(v[0] for v in kt.values()) / 10, 4)
        tb, tf = chain_traffic("8192 d1 slots") if size == 8192 and args.mix == "d1" else (None, None)
        leg["chain_counter_traffic_bytes"] = tb
        leg["traffic_source"] = tf
        if "k1_vardct" in kt:  # K1's compulsory traffic in this form: 12 B/px of pixels out + the entries / counts / LF in
            k1_ms = kt["k1_vardct"][0] / 10
            leg["k1_pixels_out_GBs"] = round(12.0 * size * size / (k1_ms * 1e-3) / 1e9, 1)
        # `sparse_resident_no_pcie` = the frame resident in the sparse transport a caller is told to use: since round 5 the
        # slot-bucketed form read in place (same key as before, so rounds compare; the pair form keeps its own line above)
        e2e["sparse_resident_no_pcie"] = leg
        e2e["slots_resident_no_pcie"] = leg
        if not args.no_sweep:
            try:
                e2e["slot_form_content_sweep"] = slot_content_sweep(jxl_rs_amd, synth, np, ectx, wl, size, args.steps,
                                                                    min(3, args.reps), usable_cores(), args.seed)
            except Exception as e:  # the headline line must survive a failing leg
                e2e["slot_form_content_sweep"] = {"error": f"{type(e).__name__}: {e}"}
        for c in ectx:   # new epoch for the PCIe legs
            c.frame_begin(synth.apply_opts(c.default_params(size, size), wl))
            c.set_dequant_tables(wl.tables)
            c.set_lf_quantized(*wl.lf_q)
            c.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
        # (the slot-bucketed legs first, on the contexts with ONE slot stream each; the older transports on contexts with
        # two, created after the first pair is closed -- see the switch below)
        legs = (("slots_pos6_val10_no_sort", submit_slots), ("slots_packed12_no_sort", submit_slots12),
                ("sparse_pairs", submit_sparse), ("sparse_pos16_val8", submit_sparse8), ("sparse_seg12_val4", submit_sparse4),
                ("dense_i32", submit_dense), ("slots_to_host_rgb8", None))
        if os.environ.get("JXLH_BENCH_E2E_ORDER") == "swap":  # leg order experiment (first-leg warm-up effects)
            legs = (legs[1], legs[0]) + legs[2:]
        def run_leg(submit, frames, pattern, after_run=None, streams=None):
            """`frames` frames round robin over the contexts, each: submit -> frame_run (-> after_run).  pattern "marks"
            (round 5): the host then waits for the mark of that context's PREVIOUS frame only (jxlh_ctx_wait_mark), so the
            upload of a context's next frame runs under its current frame's kernels; "sync": round 4's loop, a full
            jxlh_ctx_sync of the context before its next submission.  Returns ms per frame."""
            marks = [None] * NE
            ns_leg = streams or nslots  # slot streams this leg's submissions use
            if streams is not None:
                submit_all = submit
                submit = lambda ctx_: submit_all(ctx_, streams)
            t0 = time.perf_counter()
            for i in range(frames):
                k = i % NE
                c = ectx[k]
                if pattern == "sync":
                    c.sync()          # the context's previous frame is finished: its buffers can be refilled
                elif pattern == "marks_after":
                    # the same ordering on the device (jxlh_slot_after): this context's upload starts when the other
                    # context's has landed, the host does not block -- the bus stays busy back to back
                    for o in ectx:
                        if o is not c:
                            for sl in range(ns_leg):
                                for so in range(ns_leg):
                                    c.slot_after(sl, o, so)
                else:
                    # uploads serialised by the host: the OTHER context's upload has finished before this one starts, so
                    # the contexts stay in anti-phase (one uploads while the other computes) by construction.  Without
                    # this the marks loop is bistable: 0.70 or 0.93-1.2 ms per frame, tools/e2e_marks_probe.py
                    for o in ectx:
                        if o is not c:
                            for sl in range(ns_leg):
                                o.slot_wait(sl)
                submit(c)
                c.frame_run()
                if after_run is not None:
                    after_run(k)
                if pattern != "sync":
                    prev, marks[k] = marks[k], c.mark()
                    if prev is not None:
                        c.wait_mark(prev)
            for c in ectx:
                c.sync()
            return (time.perf_counter() - t0) * 1e3 / frames

        def rgb8_leg():
            # full decode-to-host: sparse pairs in, interleaved 8-bit sRGB out (jxlh_frame_read_rgb8) into
            # pinned host memory; frame i's download overlaps frame i+1's upload and kernels
            kk = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_kat.json")))["output_stage"]
            bias = np.float32(kk["opsin_bias"])
            xyb_params = np.concatenate([np.asarray(kk["opsin_inverse_matrix"], np.float32),
                                         np.full(3, np.cbrt(bias), np.float32), np.full(3, bias, np.float32),
                                         np.ones(1, np.float32)])
            rgb_bytes = size * size * 3
            pin_o = [ectx[i].alloc_pinned(rgb_bytes) for i in range(NE)]
            import ctypes as C

            def read_rgb(i):  # queued behind the frame's kernels on the context's stream; complete after c.sync()
                c = ectx[i]
                c._chk(c.L.jxlh_frame_read_rgb8_async(c._ctx, xyb_params.ctypes.data_as(C.c_void_p), 3, 0, size,
                                                      C.c_void_p(pin_o[i][1]), size * 3), "frame_read_rgb8_async")

            frames = 24
            # (the marks loop: a context's next upload starts while its current frame is still being converted and copied
            # out; the output buffer of a context is rewritten only after the mark behind its previous read has been waited for)
            run_leg(submit_slots12, 24, "marks", after_run=read_rgb)
            reps = [run_leg(submit_slots12, frames, "marks", after_run=read_rgb) for _ in range(3)]
            el_ms = sorted(reps)[1]
            e2e["slots_to_host_rgb8"] = {"value": round(size * size / 1e6 / (el_ms / 1e3), 1), "unit": "MP/s",
                                                "ms_per_frame": round(el_ms, 3), "repetitions_ms": [round(v, 3) for v in reps],
                                                "h2d_MB_per_frame": round(bytes_12 / 1e6, 1),
                                                "d2h_MB_per_frame": round(rgb_bytes / 1e6, 1), "frames": frames}

        switched = False
        for name, submit in legs:
            if name == "slots_to_host_rgb8":
                rgb8_leg()
                continue
            if not name.startswith("slots_p") and not switched and nslots != nslots_legacy:
                # The runtime maps streams onto a handful of hardware queues (4 here).  Two contexts with two slot streams
                # each are six streams: a slot stream then shares a queue with a compute stream and its copies wait behind
                # kernels.  Two contexts with ONE slot stream each are four streams, one queue each: the slot-bucketed legs
                # gain 2-11 % (tools/e2e_marks_probe.py with JXLH_PROBE_SLOTS=1 / 2, profiles/r05_m_e2e_slot_streams.txt);
                # the older transports, whose device-side unpack / sort passes sit on the slot streams, prefer two.
                keep_pinned = []  # the legs' pinned host buffers belong to the first context: hand them on
                for c in ectx:
                    keep_pinned += getattr(c, "_pinned", [])
                    c._pinned = []
                    c.close()
                nslots = nslots_legacy
                ectx = [jxl_rs_amd.Context(local_rank, n_slots=nslots) for _ in range(NE)]
                ectx[0]._pinned = keep_pinned
                for c in ectx:
                    c.frame_begin(synth.apply_opts(c.default_params(size, size), wl))
                    c.set_dequant_tables(wl.tables)
                    c.set_lf_quantized(*wl.lf_q)
                    c.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
                switched = True
            frames = 6 if name == "dense_i32" else 48  # (a repetition ends with a drain: one frame time not overlapped)
            # untimed warm-up in the same pipelined pattern: the first frames that stream from a freshly pinned
            # buffer run up to 2x slower (round 4: measured by swapping the order of the legs; round 5: the streaming
            # probe needs 40-70 frames from a new pinned buffer before it settles, tools/e2e_marks_probe.py), whichever
            # leg they belong to -- a decoder reuses its pinned staging buffers for every frame
            # ONE fixed, documented host loop per leg (round 5 reported the best of four variants for the slots
            # legs only).  The slot-bucketed legs: uploads of the two contexts ordered ON THE DEVICE (jxlh_slot_after: a
            # context's upload starts when the other's has landed, the host never blocks) -- what INTEGRATION.md tells a
            # streaming decoder to do with one slot stream per context; the host-ordered loop (jxlh_slot_wait) is the
            # side field.  The older transports keep the host-ordered loop they have always been measured with.
            primary = "marks_after" if name.startswith("slots_") else "marks"
            run_leg(submit, 60 if name != "dense_i32" else NE, primary)
            reps = [run_leg(submit, frames, primary) for _ in range(1 if name == "dense_i32" else 3)]
            ms = sorted(reps)[(len(reps) - 1) // 2]
            order_txt = {"marks": "jxlh_slot_wait (host-ordered uploads)", "marks_after": "jxlh_slot_after (device-ordered uploads)"}
            loop, alt = f"marks + {order_txt[primary]}, {nslots} slot stream{'s' if nslots > 1 else ''} per context", None
            if name.startswith("slots_"):
                alt = {f"device_ordered_{nslots}_streams_ms": [round(v, 3) for v in reps]}
                run_leg(submit, 12, "marks")
                alt[f"host_ordered_{nslots}_streams_ms"] = [round(v, 3) for v in (run_leg(submit, frames, "marks") for _ in range(3))]
            nbytes = {"sparse_pairs": total * 4, "sparse_pos16_val8": total * 3, "sparse_seg12_val4": bytes4,
                      "slots_pos6_val10_no_sort": bytes_slots,
                      "slots_packed12_no_sort": bytes_12}.get(name, wl.coeffs.nbytes)
            e2e[name] = {"value": round(size * size / 1e6 / (ms / 1e3), 1), "unit": "MP/s",
                         "ms_per_frame": round(ms, 3), "h2d_MB_per_frame": round(nbytes / 1e6, 1),
                         "frames": frames, "repetitions_ms": [round(v, 3) for v in reps], "reported": "median",
                         "host_loop": loop}
            if alt:
                e2e[name]["both_upload_orderings"] = alt
            if name.startswith("slots_"):  # round 4's host loop on the same library, for comparison
                run_leg(submit, 12, "sync", streams=nslots)
                e2e[name]["ms_per_frame_sync_loop"] = round(run_leg(submit, 24, "sync", streams=nslots), 3)
        hp = (e2e.get("slot_form_content_sweep") or {}).get("host_pack")
        if hp:  # what the producer of the slot form costs per frame on this box's host, beside every leg that streams it
            for name in list(e2e):
                if name.startswith("slots_") and isinstance(e2e[name], dict) and "h2d_MB_per_frame" in e2e[name]:
                    e2e[name]["host_pack_ms_per_frame"] = hp["host_pack_ms_per_frame"]
                    e2e[name]["host_pack_cores"] = hp["cores"]
        e2e["note"] = ("pinned host coefficients -> H2D on the context's slot streams (ONE per context in the slots_* legs, two in the older transports' legs: see host_loop) -> (pair forms: device unpack / sort; slot-bucketed "
                       "forms: nothing, the transforms read the upload in place) -> K0b/K3/K1/filters; two contexts, each "
                       "streaming its frames behind jxlh_ctx_mark / jxlh_ctx_wait_mark (the host waits for a context's "
                       "previous frame, not for the one it has just enqueued) and starting its upload when the other "
                       "context's has landed (jxlh_slot_wait); planes stay on the device except in "
                       "*_to_host_rgb8, which adds the XYB->sRGB->u8 pass and the asynchronous D2H of the interleaved image "
                       "into pinned memory")
        for c in ectx:
            c.close()

    # ---- the other BASELINE configurations, driver-visible: config 2 (4096^2, EPF off), config 4 (Modular 8192^2),
    # config 5 (16384^2, all 27 types; + epf_iters = 3).  Not `value`: one frame in flight each, a few steps.
    secondary = None
    if rank == 0 and n_gpus == 1 and not args.no_secondary and torch.cuda.is_available():
        secondary = {}
        for key, fn in (
                ("config2_4096_d1_epf0", lambda: time_vardct_config(jxl_rs_amd, synth, np, local_rank, 4096, synth.MIX_D1, 0,
                                                                    args.seed, steps=10)),
                ("config4_modular_8192", lambda: time_modular_config(jxl_rs_amd, np, local_rank, 8192, steps=5,
                                                                     cores=usable_cores(), cpu=not args.no_cpu)),
                ("config5_16384_all_types", lambda: time_vardct_config(jxl_rs_amd, synth, np, local_rank, 16384,
                                                                       synth.MIX_ALL, 2, args.seed, steps=5, extra_epf=3))):
            t0 = time.time()
            try:
                secondary[key] = fn()
            except Exception as e:  # the headline line must survive a failing secondary leg
                secondary[key] = {"error": f"{type(e).__name__}: {e}"}
            secondary[key]["leg_wall_s"] = round(time.time() - t0, 1)

    if rank == 0:
        out = result_line(strong, strong_modular, roofline, cpu, e2e, secondary)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
