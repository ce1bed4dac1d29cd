#!/usr/bin/env python3
"""GPU micro-benchmark of the Modular kernels at BASELINE config-4 size (8192x8192 x 3 ch i32),
device-resident buffers (torch only provides the device memory)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import jxl_rs_amd
from jxl_rs_amd import lib


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    ctx = jxl_rs_amd.Context(0, 1)
    L = ctx.L
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(1)
    planes = [torch.randint(0, 256, (n, n), dtype=torch.int32, device=dev, generator=g) for _ in range(3)]
    res_h = torch.randint(-8, 9, (n, n // 2), dtype=torch.int32, device=dev, generator=g)
    avg_h = torch.randint(0, 256, (n, n - n // 2), dtype=torch.int32, device=dev, generator=g)
    res_v = torch.randint(-8, 9, (n // 2, n), dtype=torch.int32, device=dev, generator=g)
    avg_v = torch.randint(0, 256, (n - n // 2, n), dtype=torch.int32, device=dev, generator=g)
    out = torch.empty((n, n), dtype=torch.int32, device=dev)
    pal = torch.randint(0, 256, (3, 256), dtype=torch.int32, device=dev, generator=g)
    idx = torch.randint(0, 256, (n, n), dtype=torch.int32, device=dev, generator=g)
    pout = torch.empty((3, n, n), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    P = lambda t: lib._addr(t)
    results = {}

    def timeit(name, fn, bytes_moved, reps=5):
        fn()
        ctx.sync()
        ctx.timer_start()
        for _ in range(reps):
            fn()
        ms = ctx.timer_stop() / reps
        results[name] = {"ms": round(ms, 4), "GB/s": round(bytes_moved / ms / 1e6, 1),
                         "frac_of_8TBs": round(bytes_moved / ms / 1e6 / 8000, 4)}

    timeit("rct_ycocg", lambda: ctx._chk(L.jxlh_rct(ctx._ctx, P(planes[0]), P(planes[1]), P(planes[2]), n * n, 6, 0), "rct"),
           24.0 * n * n)
    timeit("palette_256", lambda: ctx._chk(L.jxlh_palette(ctx._ctx, P(idx), n * n, P(pal), 256, 256, 3, 8, P(pout)), "pal"),
           16.0 * n * n)
    timeit("unsqueeze_h", lambda: ctx._chk(L.jxlh_unsqueeze(ctx._ctx, 1, P(avg_h), avg_h.shape[1], P(res_h), res_h.shape[1],
                                                             n, n, P(out), n), "uh"), 8.0 * n * n)
    timeit("unsqueeze_v", lambda: ctx._chk(L.jxlh_unsqueeze(ctx._ctx, 0, P(avg_v), n, P(res_v), n, n, n, P(out), n), "uv"),
           8.0 * n * n)
    # progressive previews: the smooth steps (no recurrence; read the average once, write the doubled channel)
    avg_q = torch.randint(0, 256, (n - n // 2, n - n // 2), dtype=torch.int32, device=dev, generator=g)
    timeit("smooth_unsqueeze_2d", lambda: ctx.smooth_unsqueeze_dev(2, avg_q, avg_q.shape[1], avg_q.shape[1],
                                                                    avg_q.shape[0], out, n, n, n), 5.0 * n * n)
    timeit("smooth_unsqueeze_h", lambda: ctx.smooth_unsqueeze_dev(0, avg_h, avg_h.shape[1], avg_h.shape[1], n, out, n, n, n),
           6.0 * n * n)
    timeit("smooth_unsqueeze_v", lambda: ctx.smooth_unsqueeze_dev(1, avg_v, n, n, avg_v.shape[0], out, n, n, n), 6.0 * n * n)
    # the three channels of a squeeze step in one launch
    avg3h = [avg_h.clone() for _ in range(3)]; res3h = [res_h.clone() for _ in range(3)]
    avg3v = [avg_v.clone() for _ in range(3)]; res3v = [res_v.clone() for _ in range(3)]
    out3 = [torch.empty((n, n), dtype=torch.int32, device=dev) for _ in range(3)]
    timeit("unsqueeze_h_x3", lambda: ctx.unsqueeze_planes(True, avg3h, res3h, out3, n, n, avg_h.shape[1], res_h.shape[1], n),
           3 * 8.0 * n * n)
    timeit("unsqueeze_v_x3", lambda: ctx.unsqueeze_planes(False, avg3v, res3v, out3, n, n, n, n, n), 3 * 8.0 * n * n)
    # ---- BASELINE config 4 as one unit: the default squeeze chain of an n x n image on three channels
    # (modular/transforms/squeeze.rs:39-105, smallest level first), every step batched over the channels, then RCT
    from jxl_rs_amd import synth
    steps, (cur_w, cur_h) = synth.default_squeeze_steps(n, n)  # (horizontal, out_w, out_h) in decoder order; base size
    cur = [torch.randint(0, 256, (cur_h, cur_w), dtype=torch.int32, device=dev, generator=g) for _ in range(3)]
    plan = []
    for horizontal, ow, oh in steps:
        rw, rh = (ow // 2, oh) if horizontal else (ow, oh // 2)
        res = [torch.randint(-8, 9, (max(rh, 1), max(rw, 1)), dtype=torch.int32, device=dev, generator=g) for _ in range(3)]
        outs = [torch.empty((oh, ow), dtype=torch.int32, device=dev) for _ in range(3)]
        plan.append((horizontal, ow, oh, res, outs))

    def chain():
        avg = cur
        for horizontal, ow, oh, res, outs in plan:
            ctx.unsqueeze_planes(horizontal, avg, res, outs, ow, oh, avg[0].shape[1], res[0].shape[1], ow)
            avg = outs
        ctx._chk(L.jxlh_rct(ctx._ctx, P(avg[0]), P(avg[1]), P(avg[2]), n * n, 6, 0), "rct")

    def chain_fused():
        avg = cur
        for horizontal, ow, oh, res, outs in plan[:-1]:
            ctx.unsqueeze_planes(horizontal, avg, res, outs, ow, oh, avg[0].shape[1], res[0].shape[1], ow)
            avg = outs
        horizontal, ow, oh, res, outs = plan[-1]
        ctx.unsqueeze_rct(horizontal, avg, res, outs, ow, oh, avg[0].shape[1], res[0].shape[1], ow, 6, 0)

    n_small = sum(1 for _, ow, oh, _, _ in plan if ow <= 128 and oh <= 128)   # the levels that fit one LDS launch

    def chain_levels():
        small = [(hz, ow, oh, res, res[0].shape[1]) for hz, ow, oh, res, _ in plan[:n_small]]
        ctx.unsqueeze_levels(small, cur, cur_w, cur_w, cur_h, plan[n_small - 1][4], plan[n_small - 1][1])
        avg = plan[n_small - 1][4]
        for horizontal, ow, oh, res, outs in plan[n_small:-1]:
            ctx.unsqueeze_planes(horizontal, avg, res, outs, ow, oh, avg[0].shape[1], res[0].shape[1], ow)
            avg = outs
        horizontal, ow, oh, res, outs = plan[-1]
        ctx.unsqueeze_rct(horizontal, avg, res, outs, ow, oh, avg[0].shape[1], res[0].shape[1], ow, 6, 0)

    samples = sum(ow * oh for _, ow, oh, _, _ in plan) * 3
    timeit("config4_chain_squeeze_rct", chain, 8.0 * samples + 24.0 * n * n, reps=3)
    results["config4_chain_squeeze_rct"]["steps"] = len(plan)
    timeit("config4_chain_fused_last_step", chain_fused, 8.0 * samples + 0.0, reps=3)
    small = [(hz, ow, oh, res, res[0].shape[1]) for hz, ow, oh, res, _ in plan[:n_small]]
    timeit("config4_small_levels_one_launch", lambda: ctx.unsqueeze_levels(small, cur, cur_w, cur_w, cur_h, plan[n_small - 1][4],
                                                                             plan[n_small - 1][1]), 1.0, reps=5)

    def small_stepwise():
        avg = cur
        for horizontal, ow, oh, res, outs in plan[:n_small]:
            ctx.unsqueeze_planes(horizontal, avg, res, outs, ow, oh, avg[0].shape[1], res[0].shape[1], ow)
            avg = outs
    timeit("config4_small_levels_stepwise", small_stepwise, 1.0, reps=5)
    timeit("config4_chain_levels_fused_last_step", chain_levels, 8.0 * samples + 0.0, reps=3)
    results["config4_chain_levels_fused_last_step"]["levels_in_one_launch"] = n_small
    last = plan[-1]
    timeit("unsqueeze_rct_last_step_x3", lambda: ctx.unsqueeze_rct(last[0], plan[-2][4], last[3], last[4], last[1], last[2],
                                                                    plan[-2][4][0].shape[1], last[3][0].shape[1], last[1], 6, 0),
           3 * 8.0 * n * n)
    results["unsqueeze_rct_last_step_x3"]["horizontal"] = bool(last[0])
    # per-level times of the chain (each step alone, its inputs from the previous level's outputs)
    levels = []
    avg = cur
    for i, (horizontal, ow, oh, res, outs) in enumerate(plan):
        a_in = avg
        fused = i == len(plan) - 1
        if fused:
            fn = lambda: ctx.unsqueeze_rct(horizontal, a_in, res, outs, ow, oh, a_in[0].shape[1], res[0].shape[1], ow, 6, 0)
        else:
            fn = lambda: ctx.unsqueeze_planes(horizontal, a_in, res, outs, ow, oh, a_in[0].shape[1], res[0].shape[1], ow)
        fn(); ctx.sync()
        ctx.timer_start()
        for _ in range(3):
            fn()
        levels.append({"step": ("h" if horizontal else "v") + ("+rct" if fused else ""), "out": [ow, oh],
                       "ms": round(ctx.timer_stop() / 3, 4)})
        avg = outs
    results["config4_levels"] = levels
    results["config4_chain_squeeze_rct"]["MP_per_s"] = round(n * n / results["config4_chain_squeeze_rct"]["ms"] / 1e3, 1)
    print(json.dumps({"size": n, "kernels": results}))


if __name__ == "__main__":
    main()
