"""Soak of the tiled squeeze kernels' 16-byte movers: single steps whose planes are 16-byte aligned with strides that are
multiples of 4 samples (so complete 64-line groups and all but a line's last chunks take the vector path), both directions,
one and three planes, every result against the oracle.  usage: soak_squeeze_vec.py [iterations]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import jxl_rs_amd
from jxl_rs_amd.lib import DeviceArray
from oracle.oracle import Oracle
o = Oracle(fused=True)
ctx = jxl_rs_amd.Context(0, 1)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bad = 0
t0 = time.time()
for it in range(n):
    rng = np.random.default_rng(9000 + it)
    lines = int(rng.integers(1, 12)) * 64 + int(rng.choice([0, 0, 0, 1, 17, 63]))   # mostly whole groups
    length = int(rng.integers(34, 700)) * 8                                           # strides of every plane % 4 == 0
    lim = int(rng.choice([64, 4096, 1 << 20, 1 << 27]))
    why = []
    for horizontal in (True, False):
        nl = lines if horizontal else ((lines + 3) // 4) * 4   # vertical: the lines are columns, the row stride = their count
        avg = rng.integers(-lim, lim, size=(nl, length // 2)).astype(np.int32)
        res = rng.integers(-lim // 8 - 1, lim // 8 + 1, size=(nl, length // 2)).astype(np.int32)
        if horizontal:
            got = ctx.unsqueeze(True, avg, res, length, nl)
            want = o.unsqueeze_h(avg, res, length)
        else:
            at, rt = np.ascontiguousarray(avg.T), np.ascontiguousarray(res.T)
            got = ctx.unsqueeze(False, at, rt, nl, length)
            want = o.unsqueeze_v(at, rt, length)
        if not np.array_equal(got, want):
            d = np.argwhere(got != want)
            why.append(f"{'h' if horizontal else 'v'} lines={nl} length={length} lim={lim}: {len(d)} samples differ, first {d[:3].tolist()}")
    bad += bool(why)
    if why or it % 100 == 99:
        print(it, "MISMATCH " + " | ".join(why) if why else "ok so far", flush=True)
print("soak_squeeze_vec mismatches:", bad, "in", n, "iterations,", round(time.time() - t0, 1), "s")
sys.exit(1 if bad else 0)
