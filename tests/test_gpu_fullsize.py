"""GPU tests at BASELINE.json's full sizes:
* VarDCT at 4096^2 / 8192^2 / 16384^2: the WHOLE frame against the oracle's whole frame, bit for bit (the C oracle
  does an 8K frame in well under a second on the box's cores), plus a band recomputed the way a rank would and
  run-to-run identity.
* 8192x8192 Modular: encode -> decode round trips (forward squeeze / forward YCoCg are independent
  numpy code), i.e. losslessness at full size, for one horizontal and one vertical full-resolution
  unsqueeze step and the RCT."""
import numpy as np
import pytest

import os

from helpers import (bit_equal, diff_report, forward_squeeze_h, forward_squeeze_v, oracle_params_from, run_gpu_frame,
                     run_oracle_frame)
import helpers

ORACLE_THREADS = max(1, min(32, len(os.sched_getaffinity(0))))


def _whole_frame_check(ctx, oracle, wl, what):
    """the GPU's whole frame == the oracle's whole frame, every pixel of every plane (and the smoothed LF)"""
    got, got_lf = run_gpu_frame(ctx, wl)
    want, want_lf = run_oracle_frame(oracle, wl, num_threads=ORACLE_THREADS)
    for c in range(3):
        assert bit_equal(got_lf[c], want_lf[c]), f"{what}: LF ch{c}"
    for c in range(3):
        assert got[c].shape == want[c].shape == (wl.ysize, wl.xsize)
        if not bit_equal(got[c], want[c]):
            bad = np.argwhere(got[c].view(np.uint32) != want[c].view(np.uint32))
            raise AssertionError(f"{what}: plane {c}: {len(bad)} of {got[c].size} differ, first {bad[:5].tolist()}")
    assert all(np.isfinite(g).all() for g in got)
    return got

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from jxl_rs_amd import Context
    c = Context(0, n_slots=1)
    yield c
    c.close()


def test_8k_vardct_whole_frame_band_and_determinism(ctx, oracle):
    from jxl_rs_amd import synth
    wl = synth.make_vardct(8192, 8192, mix=synth.MIX_D1, seed=77, unique_groups=16, epf_iters=2)
    got = _whole_frame_check(ctx, oracle, wl, "config 3 (8192^2 d1, EPF x2)")
    p = oracle_params_from(oracle, wl)
    lf = oracle.adaptive_lf_smoothing(p, oracle.dequant_lf(p, *wl.lf_q))
    # ... and one interior band the way a rank computes it (K1 on band + halo group rows, stages on the band's rows)
    for row0, row1 in ((13, 14),):
        band = oracle.vardct_band(p, wl.coeffs, wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob, lf,
                                  wl.tables, row0, row1)
        y0, y1 = row0 * 256, min(row1 * 256, wl.ysize)
        for c in range(3):
            a, b = got[c][y0:y1], band[c][y0:y1, : wl.xsize]
            assert bit_equal(a, b), f"rows {y0}:{y1} ch{c}: {diff_report(a, b)}"
    ctx.frame_run()
    ctx.sync()
    again = ctx.read_planes()
    for c in range(3):
        assert bit_equal(again[c], got[c])
    # ---- the SAME frame submitted in the slot-bucketed form, the way the bench's slot-resident and PCIe legs time it
    # (1024 groups, 17 M entries, read in place by the direct transforms; VERDICT r05: compared with the oracle at
    # <= 1024 x 768 only): packed by the C packer, two consecutive epochs so that both buffer sets are read, then with
    # one group as a dense slab and one carrying values beyond the entries' range (per-group routing).  `got` equals the
    # oracle's frame bit for bit (checked above).
    from jxl_rs_amd import lib as jl
    ng = wl.coeffs.shape[0]
    uniq = {}
    for g in range(ng):
        k = g % 16
        if k not in uniq or not np.array_equal(wl.coeffs[g], wl.coeffs[uniq[k][0]]):
            uniq[k] = (g, jl.host_pack_slots(wl.coeffs[g]))
    assert all(np.array_equal(wl.coeffs[g], wl.coeffs[uniq[g % 16][0]]) for g in range(0, ng, 37))
    parts = [uniq[g % 16][1] for g in range(ng)]
    assert all(len(q[3]) == 0 for q in parts)
    ids = np.arange(ng, dtype=np.uint32)
    ents, cnts, ns = (np.concatenate([q[0] for q in parts]), np.concatenate([q[1].reshape(-1) for q in parts]),
                      np.concatenate([q[2] for q in parts]))
    for epoch in range(2):
        ctx.submit_groups_slots(ids, ents, cnts, ns, None)
        ctx.slot_wait(0)
        ctx.frame_run()
        ctx.sync()
        sl = ctx.read_planes()
        for c in range(3):
            assert bit_equal(sl[c], got[c]), f"slot-submitted 8K frame, epoch {epoch}, plane {c}: {diff_report(sl[c], got[c])}"
    keep = [g for g in range(ng) if g not in (5, 700)]
    ctx.submit_groups_slots(np.asarray(keep, dtype=np.uint32), np.concatenate([parts[g][0] for g in keep]),
                            np.concatenate([parts[g][1].reshape(-1) for g in keep]), np.concatenate([parts[g][2] for g in keep]), None)
    ctx.submit_group(5, wl.coeffs[5])
    big = wl.coeffs[700].copy()
    big.reshape(-1)[np.flatnonzero(big.reshape(-1))[:3]] += 70000      # beyond what the packer splits: `wide`
    e7, c7, n7, w7 = jl.host_pack_slots(big, group_id=700)
    assert len(w7) == 3
    w7[:, 1] = (w7[:, 1].view(np.int32) - 70000).view(np.uint32)       # ... carrying the ORIGINAL values: same frame
    ctx.submit_groups_slots(np.uint32([700]), e7, c7.reshape(-1), n7, w7)
    ctx.slot_wait(0)
    ctx.frame_run()
    ctx.sync()
    sl = ctx.read_planes()
    for c in range(3):
        assert bit_equal(sl[c], got[c]), f"8K frame with two routed groups, plane {c}: {diff_report(sl[c], got[c])}"


def test_8k_modular_round_trips(ctx, oracle):
    rng = np.random.default_rng(8192)
    n = 8192
    img = rng.integers(0, 256, size=(n, n)).astype(np.int32)
    img[:, ::7] += rng.integers(-40, 40, size=(n, (n + 6) // 7)).astype(np.int32)
    a, r = forward_squeeze_h(img)
    assert np.array_equal(ctx.unsqueeze(True, a, r, n, n), img)
    a, r = forward_squeeze_v(img)
    assert np.array_equal(ctx.unsqueeze(False, a, r, n, n), img)
    # YCoCg-R: forward in numpy, inverse on the device
    r_, g_, b_ = img, np.roll(img, 3, axis=1), np.roll(img, 5, axis=0)
    co = r_ - b_
    tmp = b_ + (co >> 1)
    cg = g_ - tmp
    y = tmp + (cg >> 1)
    out = ctx.rct([y, co, cg], 6, 0)
    assert np.array_equal(out[0], r_) and np.array_equal(out[1], g_) and np.array_equal(out[2], b_)


def test_2k_default_squeeze_chain_round_trip(ctx):
    """Whole default-squeeze chain (squeeze.rs:39-105) at 2048x1536: forward in numpy, inverse on the GPU."""
    from jxl_rs_amd import synth
    rng = np.random.default_rng(5)
    w, h = 2048, 1536
    img = rng.integers(0, 1024, size=(h, w)).astype(np.int32)
    steps, _ = synth.default_squeeze_steps(w, h)
    cur = img
    residuals = []
    for horizontal, ow, oh in reversed(steps):  # encoder order
        assert cur.shape == (oh, ow)
        a, r = forward_squeeze_h(cur) if horizontal else forward_squeeze_v(cur)
        residuals.append(r)
        cur = a
    for (horizontal, ow, oh), r in zip(steps, reversed(residuals)):
        cur = ctx.unsqueeze(horizontal, cur, r, ow, oh)
    assert np.array_equal(cur, img)


def test_4k_config2_whole_frame_vs_oracle(ctx, oracle):
    """BASELINE configs[1]: 4096x4096 VarDCT d1 (mixed DCT8..32), IDCT + dequant + Gaborish, EPF off"""
    from jxl_rs_amd import synth
    wl = synth.make_vardct(4096, 4096, mix=synth.MIX_D1, seed=402, unique_groups=24, epf_iters=0, gab=True)
    _whole_frame_check(ctx, oracle, wl, "config 2 (4096^2, EPF off)")


def test_4k_all_types_epf0_whole_frame_vs_oracle(ctx, oracle):
    """every transform type with epf_iters = 3 (Gaborish + EPF0 + EPF1 + EPF2: the two-pass fused path) at 4096^2"""
    from jxl_rs_amd import synth
    wl = synth.make_vardct(4096, 4096, mix=synth.MIX_ALL, seed=403, unique_groups=24, epf_iters=3, gab=True)
    _whole_frame_check(ctx, oracle, wl, "4096^2 all types, epf_iters = 3")


def test_16k_config5_whole_frame_and_determinism(ctx, oracle):
    """BASELINE configs[4]: 16384x16384, every one of the 27 transform types incl. DCT256 and AFV0-3"""
    from jxl_rs_amd import synth
    wl = synth.make_vardct(16384, 16384, mix=synth.MIX_ALL, seed=1605, unique_groups=32, epf_iters=2)
    types = set(np.unique(wl.transform_map[wl.transform_map >= 128] & 127).tolist())
    # DCT256X256, the 128-pixel family, the 64-pixel family, every special 8x8 type incl. AFV0-3 (the generator seeds
    # the large types per group by area share, so one of the two 128x256 orientations may be missing for a seed)
    assert {24, 21, 18, 1, 2, 3, 12, 13, 14, 15, 16, 17} <= types and len(types) >= 25, sorted(types)
    got = _whole_frame_check(ctx, oracle, wl, "config 5 (16384^2, all 27 types)")
    ctx.frame_run()
    ctx.sync()
    again = ctx.read_planes()
    for c in range(3):
        assert bit_equal(again[c], got[c])


def test_8k_modular_chain_vs_oracle(ctx, oracle):
    """BASELINE configs[3] against the ORACLE (not only a round trip): the whole default squeeze chain of an
    8192x8192 image on three channels (residuals as in SURVEY 8(d): Laplacian(b = 3), 8-bit averages), then the
    YCoCg RCT; and a 256-colour palette expansion of an 8192x8192 index plane."""
    from jxl_rs_amd import synth
    n = 8192
    base, residuals, steps = synth.make_modular_planes(n, n, seed=84)
    cur_g = [b.copy() for b in base]
    cur_o = [b.copy() for b in base]
    for (horizontal, ow, oh), res in zip(steps, residuals):
        cur_g = [ctx.unsqueeze(horizontal, cur_g[c], res[c], ow, oh) for c in range(3)]
        if horizontal:
            cur_o = [oracle.unsqueeze_h(cur_o[c], res[c], ow) for c in range(3)]
        else:
            cur_o = [oracle.unsqueeze_v(cur_o[c], res[c], oh) for c in range(3)]
        for c in range(3):
            assert np.array_equal(cur_g[c], cur_o[c]), f"unsqueeze step {'h' if horizontal else 'v'} -> {ow}x{oh}, channel {c}"
    assert cur_g[0].shape == (n, n)
    out_g = ctx.rct(cur_g, 6, 0)
    out_o = oracle.rct(cur_o, 6, 0)
    for c in range(3):
        assert np.array_equal(out_g[c], out_o[c]), f"rct channel {c}"
    del cur_g, cur_o, out_g, out_o
    rng = np.random.default_rng(256)
    pal = rng.integers(0, 256, size=(3, 256)).astype(np.int32)
    idx = rng.integers(-3, 300, size=(n, n)).astype(np.int32)  # incl. implicit entries below 0 / beyond the palette
    got = ctx.palette(idx, pal, 256, 3, 8)
    want = oracle.palette(idx, pal, 256, 3, 8)
    assert np.array_equal(got, want)


def test_8k_modular_timed_sequence_vs_oracle(ctx, oracle):
    """BASELINE configs[3], the EXACT device-resident call bench.py times (jxl_rs_amd.modular.ModularChain.run_chain:
    one jxlh_unsqueeze_chain -- first levels in one LDS launch, streamed middle levels over three planes, last level
    fused with the YCoCg RCT) at 8192 x 8192 against the oracle's step-by-step chain + RCT, then the palette."""
    from jxl_rs_amd.modular import ModularChain
    n = 8192
    ch = ModularChain(ctx, n, n, seed=84)
    try:
        ch.run_chain()
        got = ch.result()
        want = helpers.modular_chain_oracle(ch, oracle)
        for c in range(3):
            assert np.array_equal(got[c], want[c]), f"channel {c}: first mismatches {np.argwhere(got[c] != want[c])[:5]}"
        del want
        ch.run_chain()
        again = ch.result()
        for c in range(3):
            assert np.array_equal(again[c], got[c]), "run-to-run"
    finally:
        ch.free()


@pytest.mark.parametrize("weighted", [False, True])
def test_delta_palette_many_bands_bit_exact(ctx, oracle, weighted):
    """2048 x 3072 x 3 channels: twelve 256-row bands per channel pipelined across workgroups (progress counters,
    write-through stores, the rows above a band read from another XCD) -- the whole image against the raster-order
    oracle, for the gradient predictor and for the Weighted one (whose band-edge state rows travel the same way)."""
    rng = np.random.default_rng(77 + weighted)
    h, w, nb = 3072, 2048, 3
    num_colors, num_deltas = 40, 8
    pal = rng.integers(0, 256, size=(nb, num_colors + num_deltas)).astype(np.int32)
    pal[:, :num_deltas] = rng.integers(-6, 7, size=(nb, num_deltas))
    idx = rng.integers(0, num_colors + num_deltas, size=(h, w)).astype(np.int32)
    idx[rng.random((h, w)) < 0.5] = rng.integers(0, num_deltas)
    if weighted:
        hdr = (16, 10, 7, 7, 7, 0, 0, 13, 12, 12, 12)
        got = ctx.palette_delta_wp(idx, pal, num_colors, num_deltas, 8, hdr)
        want = oracle.palette_delta_wp(idx, pal, num_colors, num_deltas, nb, 8, hdr)
    else:
        got = ctx.palette_delta(idx, pal, num_colors, num_deltas, 8, 5)
        want = oracle.palette_delta(idx, pal, num_colors, num_deltas, 8, 5)
    assert np.array_equal(got, want), np.argwhere(got != want)[:5]
