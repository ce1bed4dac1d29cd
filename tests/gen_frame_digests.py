#!/usr/bin/env python3
"""Writes tests/golden/frame_digests.json: SHA-256 digests of the CPU oracle's output (FMA build) for a fixed list of
seeded synthetic workloads -- whole VarDCT frames (every stage list the device path takes, chroma subsampling, frame
upsampling, noise, 8-bit output) and Modular chains (squeeze + RCT, palette).  The fixtures freeze today's oracle:
tests/test_golden_frames.py checks that the oracle still reproduces them (CPU) and that the device output has the
same digests (GPU), so neither side can drift unnoticed, together or alone.  `inputs` is the digest of the generated
workload itself, so a failure tells a changed generator from a changed reconstruction.

usage: python tests/gen_frame_digests.py [--check]      (--check: compare instead of writing)"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

OUT = os.path.join(ROOT, "tests", "golden", "frame_digests.json")

# name, width, height, mix, seed, make_vardct options
VARDCT_CASES = [
    ("d1_default", 600, 520, "MIX_D1", 11, dict(epf_iters=2, gab=True, lf_smoothing=True)),
    ("all_types_epf3", 520, 776, "MIX_ALL", 12, dict(epf_iters=3, gab=True, lf_smoothing=True)),
    ("all_types_epf1_no_gab", 333, 257, "MIX_ALL", 13, dict(epf_iters=1, gab=False, lf_smoothing=True)),
    ("dct8_no_filters", 300, 200, "MIX_DCT8", 14, dict(epf_iters=0, gab=False, lf_smoothing=False)),
    ("gab_only_ragged", 71, 513, "MIX_D1", 15, dict(epf_iters=0, gab=True, lf_smoothing=True)),
    ("tiny", 5, 3, "MIX_DCT8", 16, dict(epf_iters=2, gab=True, lf_smoothing=True)),
    ("subsampled_420", 400, 304, "MIX_8X8", 17, dict(epf_iters=2, gab=True, lf_smoothing=False, hshift=(1, 0, 1), vshift=(1, 0, 1))),
    ("subsampled_422_plain", 250, 250, "MIX_8X8", 18, dict(epf_iters=0, gab=False, lf_smoothing=False, hshift=(1, 0, 1), vshift=(0, 0, 0))),
]
MODULAR_CASES = [("chain_ycocg", 300, 260, 21, (6, 0)), ("chain_rct_perm", 517, 129, 22, (3, 4)), ("chain_no_rct", 64, 700, 23, None)]


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode() + str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()


def vardct_case(oracle, name, w, h, mix, seed, opts):
    import helpers
    from jxl_rs_amd import synth
    wl = synth.make_vardct(w, h, mix=getattr(synth, mix), seed=seed, **opts)
    planes, lf_sm = helpers.run_oracle_frame(oracle, wl)
    inputs = sha(wl.coeffs, wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob, *wl.lf_q, *wl.tables)
    return wl, planes, lf_sm, {"inputs": inputs, "planes": [sha(p) for p in planes], "lf": [sha(l) for l in lf_sm]}


def modular_case(oracle, name, w, h, seed, rct):
    from jxl_rs_amd import synth
    base, residuals, steps = synth.make_modular_planes(w, h, seed=seed)
    cur = [b.copy() for b in base]
    for (hz, ow, oh), res in zip(steps, residuals):
        cur = [oracle.unsqueeze_h(cur[c], res[c], ow) if hz else oracle.unsqueeze_v(cur[c], res[c], oh) for c in range(3)]
    out = oracle.rct(cur, *rct) if rct is not None else cur
    inputs = sha(*base, *[r for lvl in residuals for r in lvl])
    return (base, residuals, steps), out, {"inputs": inputs, "planes": [sha(p) for p in out]}


def generate():
    from oracle.oracle import Oracle
    oracle = Oracle(fused=True)
    doc = {"_what": "SHA-256 of dtype + shape + bytes of the oracle's (FMA build) outputs; tests/gen_frame_digests.py",
           "vardct": {}, "modular": {}}
    for name, w, h, mix, seed, opts in VARDCT_CASES:
        doc["vardct"][name] = vardct_case(oracle, name, w, h, mix, seed, opts)[3]
    for name, w, h, seed, rct in MODULAR_CASES:
        doc["modular"][name] = modular_case(oracle, name, w, h, seed, rct)[2]
    return doc


if __name__ == "__main__":
    doc = generate()
    if "--check" in sys.argv:
        want = json.load(open(OUT))
        bad = [k for sec in ("vardct", "modular") for k in doc[sec] if doc[sec][k] != want[sec].get(k)]
        print("differs:", bad if bad else "nothing")
        sys.exit(1 if bad else 0)
    with open(OUT, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    print("wrote", OUT, len(doc["vardct"]), "frames,", len(doc["modular"]), "chains")
