"""Soak of the cross-band protocol of the palette wavefront kernels: many-band images, random content, every run
compared with the raster-order oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import jxl_rs_amd
from oracle.oracle import Oracle
o = Oracle(fused=True)
ctx = jxl_rs_amd.Context(0, 1)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = 0
t0 = time.time()
for it in range(n):
    rng = np.random.default_rng(1000 + it)
    h = int(rng.integers(600, 4200)); w = int(rng.integers(40, 2600)); nb = int(rng.integers(1, 4))
    ncol, nd = 40, 8
    pal = rng.integers(0, 256, size=(nb, ncol + nd)).astype(np.int32)
    pal[:, :nd] = rng.integers(-6, 7, size=(nb, nd))
    idx = rng.integers(0, ncol + nd, size=(h, w)).astype(np.int32)
    idx[rng.random((h, w)) < 0.5] = rng.integers(0, nd)
    pred = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 10, 11, 12, 13]))
    got = ctx.palette_delta(idx, pal, ncol, nd, 8, pred)
    want = o.palette_delta(idx, pal, ncol, nd, 8, pred)
    ok1 = np.array_equal(got, want)
    hdr = [int(v) for v in rng.integers(0, 32, size=7)] + [int(v) for v in rng.integers(0, 16, size=4)]
    got = ctx.palette_delta_wp(idx, pal, ncol, nd, 8, hdr)
    want = o.palette_delta_wp(idx, pal, ncol, nd, nb, 8, hdr)
    ok2 = np.array_equal(got, want)
    bad += (not ok1) + (not ok2)
    print(it, f"{w}x{h}x{nb} pred={pred}", "ok" if ok1 else "DELTA MISMATCH", "ok" if ok2 else "WP MISMATCH", flush=True)
print("mismatches:", bad, "in", n, "iterations,", round(time.time() - t0, 1), "s")
sys.exit(1 if bad else 0)
