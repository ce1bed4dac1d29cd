"""Soak of the dataflow squeeze launch (k6_unsqueeze_flow): the same chain run many times back to back -- every run must
reproduce the first run's planes, which are checked against the level-by-level launches (JXLH_CHAIN_FLOW=0) and, for the
smaller shapes, against the oracle.  A race between a level's stores and the next level's loads (publication before
acknowledgement, a stale line in another XCD's L2) would show up as a run that differs.  With "noise" as second argument a second context renders VarDCT frames on the same GPU from another host thread the whole time
(other kernels take workgroup slots and memory bandwidth at random moments).  usage: soak_chain_flow.py [runs] [noise]"""
import os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import jxl_rs_amd
import helpers
from jxl_rs_amd.modular import ModularChain
from oracle.oracle import Oracle

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
o = Oracle(fused=True)
ctx = jxl_rs_amd.Context(0, 1)
stop = False
if len(sys.argv) > 2 and sys.argv[2] == "noise":
    import threading
    from jxl_rs_amd import synth
    wl = synth.make_vardct(4096, 4096, mix=synth.MIX_ALL, seed=9, unique_groups=16, epf_iters=2)
    nctx = jxl_rs_amd.Context(0, 1)
    nctx.frame_begin(synth.apply_opts(nctx.default_params(4096, 4096), wl))
    nctx.set_dequant_tables(wl.tables); nctx.set_lf_quantized(*wl.lf_q)
    nctx.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
    for g in range(wl.coeffs.shape[0]):
        nctx.submit_group(g, wl.coeffs[g])
    nctx.slot_wait(0)

    def noise():
        k = 0
        while not stop:
            for _ in range(1 + k % 4):
                nctx.frame_run()
            nctx.sync()
            if k % 3 == 0:
                time.sleep(0.0007 * (k % 5))
            k += 1
    th = threading.Thread(target=noise, daemon=True)
    th.start()
if len(sys.argv) > 2 and sys.argv[2] == "chains":
    # a second context running its OWN dataflow launches the whole time: two persistent ticket kernels share the device
    import threading
    cctx = jxl_rs_amd.Context(0, 1)
    other = ModularChain(cctx, 4096, 4096, seed=77, rct=(6, 0))
    os.environ["JXLH_CHAIN_FLOW"] = "0"
    other.run_chain(); cctx.sync()
    other_ref = [zlib.crc32(p.tobytes()) for p in other.result()]
    other_bad = [0]

    def chains():
        k = 0
        while not stop:
            for _ in range(1 + k % 3):
                other.run_chain()
            cctx.sync()
            if k % 7 == 0 and [zlib.crc32(p.tobytes()) for p in other.result()] != other_ref:
                other_bad[0] += 1
            k += 1
    th = threading.Thread(target=chains, daemon=True)
    th.start()
bad = 0
t0 = time.time()
for (w, h, rct, check_oracle) in ((8192, 8192, (6, 0), False), (8192, 8192, None, False), (4096, 2048, (3, 4), True),
                                  (2051, 4100, None, True), (1024, 6000, (6, 0), True)):
    ch = ModularChain(ctx, w, h, seed=w + h, rct=rct)
    os.environ["JXLH_CHAIN_FLOW"] = "0"
    ch.run_chain(); ctx.sync()
    ref = [zlib.crc32(p.tobytes()) for p in ch.result()]
    if check_oracle:
        want = helpers.modular_chain_oracle(ch, o)
        if ref != [zlib.crc32(np.ascontiguousarray(p).tobytes()) for p in want]:
            bad += 1
            print("level-by-level != oracle", w, h, rct)
    os.environ["JXLH_CHAIN_FLOW"] = "1"
    n_bad = 0
    for it in range(runs):
        for d in ch.d_out:  # poison: a level that is skipped or read early cannot pass on old contents
            if it % 10 == 0:
                d.upload(np.full(16, -1, np.int32))
        ch.run_chain()
        if it % 5 == 4 or it == 0:  # back-to-back runs in between: the next launch's memset meets the previous launch's tail
            ctx.sync()
            got = [zlib.crc32(p.tobytes()) for p in ch.result()]
            if got != ref:
                n_bad += 1
    ctx.sync()
    print(f"{w}x{h} rct={rct}: {runs} runs, {n_bad} differing", flush=True)
    bad += n_bad
    ch.free()
stop = True
if len(sys.argv) > 2 and sys.argv[2] == "chains":
    th.join(timeout=30)
    print("the other context's chains:", other_bad[0], "differing")
    bad += other_bad[0]
print(f"soak_chain_flow: {'OK' if bad == 0 else 'FAILED'} ({bad} bad) in {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
