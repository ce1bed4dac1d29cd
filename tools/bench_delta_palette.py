#!/usr/bin/env python3
"""Delta-palette wavefront kernel timing: usage tools/bench_delta_palette.py [WxH ...] (gradient predictor, 3 channels)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jxl_rs_amd
from jxl_rs_amd import lib
c = jxl_rs_amd.Context(0, 1)
ND = int(os.environ.get("ND", "8"))
sizes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(1024, 1024), (4096, 4096), (8192, 8192)]
for w, h in sizes:
    rng = np.random.default_rng(1)
    idx = torch.from_numpy(rng.integers(0, 40, size=(h, w)).astype(np.int32)).cuda()
    pal = torch.from_numpy(rng.integers(-10, 256, size=(3, 64)).astype(np.int32)).cuda()
    out = torch.empty((3, h, w), dtype=torch.int32, device="cuda")
    def run():
        c._chk(c.L.jxlh_palette_delta(c._ctx, lib._addr(idx), w, h, lib._addr(pal), 64 - ND, ND, 64, 3, 8, 5, lib._addr(out)), "pd")
    run(); c.sync()
    t0 = time.perf_counter(); run(); c.sync(); t = time.perf_counter() - t0
    steps = w + 3 * h
    print(f"{w}x{h}: {t*1e3:.2f} ms  ({w*h/t/1e6:.1f} MP/s, {t*1e6/steps:.2f} us per wavefront step of {steps})")
    hdr = np.array((16, 10, 7, 7, 7, 0, 0, 13, 12, 12, 12), np.uint32)   # Weighted predictor, default header
    def run_wp():
        c._chk(c.L.jxlh_palette_delta_wp(c._ctx, lib._addr(idx), w, h, lib._addr(pal), 64 - ND, ND, 64, 3, 8, lib._addr(hdr), lib._addr(out)), "wp")
    run_wp(); c.sync()
    t0 = time.perf_counter(); run_wp(); c.sync(); t = time.perf_counter() - t0
    print(f"{w}x{h} weighted: {t*1e3:.2f} ms  ({w*h/t/1e6:.1f} MP/s, {t*1e6/steps:.2f} us per wavefront step)")
