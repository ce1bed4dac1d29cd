#!/usr/bin/env python3
"""How often does a context's placement pick find a fast candidate?  N contexts (held), each tuned with T trials:
candidates seen and the pick's ratings.   python tools/placement_rate.py [contexts] [trials]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import jxl_rs_amd
from jxl_rs_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 12
wl = synth.make_vardct(256, 256, mix=synth.MIX_D1, seed=1)
keep, out = [], []
for i in range(n):
    c = jxl_rs_amd.Context(0, n_slots=1)
    c.tune_placement(trials)
    p = c.default_params(8192, 8192)
    c.frame_begin(p)
    ratings, pick = c.tune_placement()
    out.append((len(ratings), round(ratings[pick][0], 4), round(ratings[pick][1], 4)))
    keep.append(c)
print(out, "slow picks:", sum(1 for o in out if o[1] > 0.295), flush=True)
