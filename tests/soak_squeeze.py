"""Soak of the squeeze kernels on random shapes: single steps (streamed and one-wave kernels), the fused unsqueeze +
RCT in both directions and tile widths, and whole default chains through jxlh_unsqueeze_levels + steps -- every result
compared with the oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import jxl_rs_amd
from jxl_rs_amd import synth
from oracle.oracle import Oracle
from jxl_rs_amd.lib import DeviceArray
o = Oracle(fused=True)
ctx = jxl_rs_amd.Context(0, 1)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
t0 = time.time()
for it in range(n):
    rng = np.random.default_rng(5000 + it)
    # --- one step, random geometry (long lines likely)
    lines = int(rng.integers(1, 700)); length = int(rng.integers(1, 5000))
    lim = int(rng.choice([64, 4096, 1 << 20, 1 << 27]))
    avg = rng.integers(-lim, lim, size=(lines, (length + 1) // 2)).astype(np.int32)
    res = rng.integers(-lim // 8 - 1, lim // 8 + 1, size=(lines, length // 2)).astype(np.int32)
    why = []
    ok = np.array_equal(ctx.unsqueeze(True, avg, res, length, lines), o.unsqueeze_h(avg, res, length))
    if not ok: why.append("step-h")
    at, rt = np.ascontiguousarray(avg.T), np.ascontiguousarray(res.T)
    okv = np.array_equal(ctx.unsqueeze(False, at, rt, lines, length), o.unsqueeze_v(at, rt, length))
    if not okv: why.append("step-v")
    ok &= okv
    # --- fused unsqueeze + RCT, vertical (wide aligned planes take the 32-column tiles) and horizontal
    op, perm = int(rng.integers(0, 7)), int(rng.integers(0, 6))
    for horizontal in (True, False):
        cols = int(rng.choice([4096, 4128, 64, 132, 1000])) if not horizontal else int(rng.integers(2, 400))
        ln = int(rng.integers(2, 300))
        ow, oh = (ln, cols) if False else ((cols, ln) if not horizontal else (ln * 1, cols))
        # horizontal: ow = length along x; vertical: oh = length along y
        if horizontal:
            ow, oh = int(rng.integers(2, 900)), int(rng.integers(1, 200))
            aw, ah, rw, rh = (ow + 1) // 2, oh, ow // 2, oh
        else:
            ow, oh = cols, int(rng.integers(2, 300))
            aw, ah, rw, rh = ow, (oh + 1) // 2, ow, oh // 2
        host = []
        for c in range(3):
            host.append((rng.integers(-3000, 3000, size=(ah, aw)).astype(np.int32),
                         np.round(rng.laplace(0, 30, size=(rh, rw))).astype(np.int32)))
        da = [DeviceArray(a) for a, _ in host]; dr = [DeviceArray(r if r.size else np.zeros((1, max(rw, 1)), np.int32)) for _, r in host]
        do = [DeviceArray(nbytes=ow * oh * 4) for _ in range(3)]
        ctx.unsqueeze_rct(horizontal, [d.ptr for d in da], [d.ptr for d in dr], [d.ptr for d in do], ow, oh, aw, max(rw, 1), ow, op, perm)
        ctx.sync()
        unsq = [o.unsqueeze_h(a, r, ow) if horizontal else o.unsqueeze_v(a, r, oh) for a, r in host]
        want = o.rct(unsq, op, perm)
        for c in range(3):
            okr = np.array_equal(do[c].download(np.int32, ow * oh).reshape(oh, ow), want[c].reshape(oh, ow))
            if not okr: why.append(f"rct-{'h' if horizontal else 'v'} {ow}x{oh} op{op} perm{perm} c{c}")
            ok &= okr
        for d in da + dr + do:
            d.free()
    # --- a whole default chain: levels call for the small levels, steps for the rest
    w, h = int(rng.integers(9, 900)), int(rng.integers(9, 900))
    base, residuals, steps = synth.make_modular_planes(w, h, seed=it, nchan=3)
    want = [b.copy() for b in base]
    for (hz, ow, oh), res3 in zip(steps, residuals):
        want = [o.unsqueeze_h(want[c], res3[c], ow) if hz else o.unsqueeze_v(want[c], res3[c], oh) for c in range(3)]
    dbase = [DeviceArray(b) for b in base]
    keep, levels = [], []
    for (hz, ow, oh), res3 in zip(steps, residuals):
        ps = []
        for c in range(3):
            r = res3[c] if res3[c].size else np.zeros((1, 1), np.int32)
            d = DeviceArray(r); keep.append(d); ps.append(d.ptr)
        levels.append((hz, ow, oh, ps, max(res3[0].shape[1], 1)))
    dout = [DeviceArray(nbytes=w * h * 4) for _ in range(3)]
    bh, bw = base[0].shape
    ctx.unsqueeze_levels(levels, [d.ptr for d in dbase], bw, bw, bh, [d.ptr for d in dout], w)
    ctx.sync()
    for c in range(3):
        okc = np.array_equal(dout[c].download(np.int32, w * h).reshape(h, w), want[c])
        if not okc: why.append(f"chain c{c}")
        ok &= okc
    for d in dbase + keep + dout:
        d.free()
    bad += not ok
    print(it, "ok" if ok else "MISMATCH " + "; ".join(why), f"step {lines}x{length} lim={lim}; chain {w}x{h}", flush=True)
print("mismatches:", bad, "in", n, "iterations,", round(time.time() - t0, 1), "s")
sys.exit(1 if bad else 0)
