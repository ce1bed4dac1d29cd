"""A/B timing of the strip kernel (k123_strip) against the two-kernel path on one box, same frame, alternating.
usage: python tools/strip_time.py [--size 8192] [--steps 20] [--mix d1] [--epf spec|active] [--unaligned] [--epf-iters 2]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np  # noqa: E402
import jxl_rs_amd  # noqa: E402
from jxl_rs_amd import synth  # noqa: E402
from jxl_rs_amd import lib as jl  # noqa: E402


def setup(wl, size, flags, device=0):
    c = jxl_rs_amd.Context(device, n_slots=1)
    p = synth.apply_opts(c.default_params(size, size), wl)
    p.flags = flags
    c.frame_begin(p)
    c.set_dequant_tables(wl.tables)
    c.set_lf_quantized(*wl.lf_q)
    c.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
    for g in range(wl.coeffs.shape[0]):
        c.submit_group(g, wl.coeffs[g])
    c.slot_wait(0)
    return c


def timed(ctxs, steps, warmup=3):
    n = [0]

    def step():
        ctxs[n[0] % len(ctxs)].frame_run()
        n[0] += 1
    for _ in range(warmup):
        step()
    for c in ctxs:
        c.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    for c in ctxs:
        c.sync()
    return (time.perf_counter() - t0) * 1e3 / steps


def ktable(ctx, n=5):
    ctx.kernel_timing_reset()
    ctx.kernel_timing(True)
    for _ in range(n):
        ctx.frame_run()
    ctx.sync()
    kt = ctx.kernel_times()
    ctx.kernel_timing(False)
    return {k: round(ms / n, 4) for k, (ms, _) in kt.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=8192)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--mix", default="d1")
    ap.add_argument("--epf", default="spec")
    ap.add_argument("--epf-iters", type=int, default=2)
    ap.add_argument("--unaligned", action="store_true")
    ap.add_argument("--seed", type=int, default=3)
    a = ap.parse_args()
    mix = {"d1": synth.MIX_D1, "dct8": synth.MIX_DCT8, "all": synth.MIX_ALL}[a.mix]
    wl = synth.make_vardct(a.size, a.size, mix=mix, seed=a.seed, unique_groups=24, epf_iters=a.epf_iters, gab=True,
                           aligned=not a.unaligned)
    if a.epf == "active":
        wl.epf_map[:] = 7
        wl.raw_quant[:] = np.minimum(wl.raw_quant, 4)
    out = {"size": a.size, "mix": a.mix, "aligned": not a.unaligned, "epf": a.epf, "epf_iters": a.epf_iters}
    strip = [setup(wl, a.size, jl.FRAME_STRIP) for _ in range(2)]
    two = [setup(wl, a.size, 0) for _ in range(2)]
    strip[0].frame_run()
    strip[0].sync()
    out["path"] = strip[0].frame_path()
    res = {"strip_1": [], "two_1": [], "strip_2": [], "two_2": []}
    for _ in range(a.reps):
        res["strip_1"].append(round(timed(strip[:1], a.steps), 4))
        res["two_1"].append(round(timed(two[:1], a.steps), 4))
        res["strip_2"].append(round(timed(strip, a.steps), 4))
        res["two_2"].append(round(timed(two, a.steps), 4))
    out["ms_per_frame"] = res
    L = strip[0].L
    if hasattr(L, "jxlh_strip_prof_read"):
        import ctypes as C
        buf = (C.c_ulonglong * 20)()
        L.jxlh_strip_prof_read(buf, 1)
        strip[0].frame_run()
        strip[0].sync()
        L.jxlh_strip_prof_read(buf, 1)
        out["one_launch_us"] = {"first_to_last_start": (buf[17] - buf[16]) / 100.0, "first_start_to_first_end": (buf[18] - buf[16]) / 100.0,
                                "first_start_to_last_end": (buf[19] - buf[16]) / 100.0}
        if hasattr(L, "jxlh_strip_end_read"):
            eb = (C.c_ulonglong * 2048)()
            L.jxlh_strip_end_read(eb)
            t0 = buf[16]
            ends = [(eb[i] - t0) / 100.0 for i in range(512)]
            hw = [eb[1024 + i] for i in range(512)]
            # HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 (gfx9)
            cu = [((h >> 8) & 15) | (((h >> 12) & 1) << 4) | (((h >> 13) & 7) << 5) | (((h >> 32) & 15) << 8) for h in hw]
            out["band_end_us"] = [[round(min(ends[b * 128:(b + 1) * 128])), round(max(ends[b * 128:(b + 1) * 128]))] for b in range(4)]
            out["band0_end_by_strip"] = [round(ends[i]) for i in range(0, 128, 4)]
            out["band2_end_by_strip"] = [round(ends[256 + i]) for i in range(0, 128, 4)]
            from collections import Counter
            cnt = Counter(cu)
            out["wgs_per_cu_hist"] = dict(Counter(cnt.values()))
            out["distinct_cus"] = len(cnt)
            solo = [ends[i] for i in range(512) if cnt[cu[i]] == 1]
            duo = [ends[i] for i in range(512) if cnt[cu[i]] == 2]
            out["end_us_solo_vs_duo"] = [round(sum(solo) / max(1, len(solo))), round(sum(duo) / max(1, len(duo)))]
        for _ in range(5):
            strip[0].frame_run()
        strip[0].sync()
        L.jxlh_strip_prof_read(buf, 0)
        names = ["desc", "tasks", "dequant+llf", "pass1", "pass2", "publish", "wait", "halo", "sync", "save+mirror", "gab",
                 "epf1", "epf2", "restore"]
        tot = sum(buf[i] for i in range(14)) or 1
        out["prof_pct"] = {n: round(100.0 * buf[i] / tot, 1) for i, n in enumerate(names)}
        out["prof_us_per_wg"] = round(tot / 100.0 / 5 / (out["path"][1] / 32 if False else 512), 1)
    out["kernels_strip"] = ktable(strip[0])
    out["kernels_two"] = ktable(two[0])
    # same bits?
    strip[0].frame_run(); strip[0].sync()
    two[0].frame_run(); two[0].sync()
    pa, pb = strip[0].read_planes(), two[0].read_planes()
    out["bit_equal"] = all(np.array_equal(x.view(np.uint32), y.view(np.uint32)) for x, y in zip(pa, pb))
    print(json.dumps(out))
    for c in strip + two:
        c.close()


if __name__ == "__main__":
    main()
