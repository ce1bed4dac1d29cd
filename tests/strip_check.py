"""Ad-hoc parity check of the strip kernel against the oracle (development tool; the tests proper are
tests/test_gpu_strip.py).  usage: python tools/strip_check.py [case ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.dirname(__file__))
from jxl_rs_amd import Context, synth  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402
import helpers  # noqa: E402


def check(ctx, o, name, w, h, mix, aligned, epf, gab, seed=1, unique=None):
    wl = synth.make_vardct(w, h, mix=mix, seed=seed, unique_groups=unique, epf_iters=epf, gab=gab, aligned=aligned)
    t0 = time.time()
    got, _ = helpers.run_gpu_frame(ctx, wl, flags=4)  # JXLH_FRAME_STRIP
    path = ctx.frame_path()
    want, _ = helpers.run_oracle_frame(o, wl, num_threads=16)
    bad = 0
    first = None
    for c in range(3):
        m = got[c].view(np.uint32) != want[c].view(np.uint32)
        n = int(m.sum())
        if n and first is None:
            ys, xs = np.nonzero(m)
            first = (c, int(ys[0]), int(xs[0]), int(ys.min()), int(ys.max()), int(xs.min()), int(xs.max()))
        bad += n
    print(f"{name:40s} {w}x{h} epf{epf} gab{int(gab)} path={path} bad={bad} first={first} ({time.time() - t0:.1f}s)",
          flush=True)
    return bad == 0


def main():
    o = Oracle(fused=True)
    ctx = Context(0, n_slots=1)
    ok = True
    cases = [
        ("dct8 one tile", 64, 64, synth.MIX_DCT8, True, 2, True),
        ("dct8 2x2 tiles", 128, 128, synth.MIX_DCT8, True, 2, True),
        ("dct8 gab only", 128, 128, synth.MIX_DCT8, True, 0, True),
        ("dct8 epf1 only", 128, 128, synth.MIX_DCT8, True, 1, False),
        ("d1 aligned 256", 256, 256, synth.MIX_D1, True, 2, True),
        ("d1 aligned odd size", 200, 136, synth.MIX_D1, True, 2, True),
        ("d1 aligned odd size 2", 333, 77, synth.MIX_D1, True, 2, True),
        ("d1 aligned 1024x768", 1024, 768, synth.MIX_D1, True, 2, True),
        ("d1 aligned gab+epf1", 520, 328, synth.MIX_D1, True, 1, True),
        ("d1 aligned epf1+2 no gab", 520, 328, synth.MIX_D1, True, 2, False),
        ("d1 unaligned (mixed tiles)", 512, 512, synth.MIX_D1, False, 2, True),
        ("all types aligned", 1024, 1024, synth.MIX_ALL, True, 2, True),
        ("all types unaligned", 1024, 1024, synth.MIX_ALL, False, 2, True),
        ("d1 aligned 4096", 4096, 4096, synth.MIX_D1, True, 2, True, 3, 24),
    ]
    sel = sys.argv[1:]
    for cse in cases:
        if sel and not any(s in cse[0] for s in sel):
            continue
        ok &= check(ctx, o, *cse)
    ctx.close()
    print("ALL OK" if ok else "FAILURES")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
