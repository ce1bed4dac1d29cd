"""Randomised differential test: frames of random (ragged) sizes, transform mixes, stage lists and
HEADER PARAMETERS (quantiser, chroma-from-luma, Gaborish weights, every EPF knob) through the C ABI
vs the CPU oracle, bit for bit.  The fixed parity cases use the reference's header defaults; this one
walks the parameter space the reference's FrameHeader / RestorationFilter / ColorCorrelationParams
can express (headers/frame_header.rs:146-233, frame/color_correlation_map.rs:21-94)."""
import os

import numpy as np
import pytest

from helpers import bit_equal, diff_report, run_gpu_frame, run_oracle_frame

pytestmark = pytest.mark.gpu

# soak runs (tools/soak_vardct.sh): JXLH_FUZZ_OFFSET shifts every seed, JXLH_FUZZ_SCALE multiplies the frame sizes
# (more groups per frame: 4-group scan workgroups with dead quarters, several group rows)
FUZZ_OFFSET = int(os.environ.get("JXLH_FUZZ_OFFSET", "0"))
FUZZ_SCALE = int(os.environ.get("JXLH_FUZZ_SCALE", "1"))


@pytest.fixture(scope="module")
def ctx():
    from jxl_rs_amd import Context
    c = Context(0, n_slots=1)
    yield c
    c.close()


def _random_case(rng):
    from jxl_rs_amd import synth
    w = int(rng.choice([rng.integers(1, 40), rng.integers(40, 300), rng.integers(300, 700)])) * FUZZ_SCALE
    h = int(rng.choice([rng.integers(1, 40), rng.integers(40, 300), rng.integers(300, 700)])) * FUZZ_SCALE
    mix = [synth.MIX_DCT8, synth.MIX_D1, synth.MIX_ALL][int(rng.integers(0, 3))]
    opts = dict(epf_iters=int(rng.integers(0, 4)), gab=bool(rng.integers(0, 2)), lf_smoothing=bool(rng.integers(0, 2)))
    over = {}
    if rng.random() < 0.8:
        over["global_scale"] = int(rng.integers(2000, 60000))
        over["quant_lf"] = int(rng.integers(1, 200))
        over["x_qm_scale"] = int(rng.integers(0, 8))
        over["b_qm_scale"] = int(rng.integers(0, 8))
        over["color_factor"] = int(rng.integers(1, 200))
        over["base_correlation_x"] = float(np.float32(rng.uniform(-1, 1)))
        over["base_correlation_b"] = float(np.float32(rng.uniform(0, 2)))
        over["epf_quant_mul"] = float(np.float32(rng.uniform(0.1, 1.5)))
        over["epf_pass0_sigma_scale"] = float(np.float32(rng.uniform(0.3, 2)))
        over["epf_pass2_sigma_scale"] = float(np.float32(rng.uniform(1, 10)))
        over["epf_border_sad_mul"] = float(np.float32(rng.uniform(0.2, 1.2)))
    arrays = {}
    if rng.random() < 0.7:
        arrays["gab_w1"] = rng.uniform(0.0, 0.3, 3).astype(np.float32)
        arrays["gab_w2"] = rng.uniform(0.0, 0.15, 3).astype(np.float32)
        arrays["epf_channel_scale"] = rng.uniform(1.0, 60.0, 3).astype(np.float32)
        arrays["epf_sharp_lut"] = np.sort(rng.uniform(0.0, 1.5, 8)).astype(np.float32)
        arrays["quant_biases"] = np.concatenate([rng.uniform(0.85, 1.0, 3), rng.uniform(0.05, 0.3, 1)]).astype(np.float32)
        arrays["lf_quant_factors"] = (1.0 / rng.uniform(100, 8000, 3)).astype(np.float32)
    return w, h, mix, opts, over, arrays


class _Setter(dict):
    """kwargs for helpers.*_params_from: scalars via setattr, arrays element-wise"""


def _apply_arrays(p, arrays):
    for name, vals in arrays.items():
        field = getattr(p, name)
        for i, v in enumerate(vals):
            field[i] = float(v)


def _submit_rotating(ctx, synth, wl, seed):
    """the three submission forms take turns: dense slabs, (position, value) pairs (device sort, the transforms read the
    bucketed pairs), slot-bucketed entries (read in place: direct dequantisation, fallback list, dense route for values
    beyond 10 bits)"""
    ng = wl.coeffs.shape[0]
    if seed % 3 == 0:
        for g in range(ng):
            ctx.submit_group(g, wl.coeffs[g])
    elif seed % 3 == 1:
        for g in range(ng):
            ctx.submit_group_sparse(g, *synth.to_sparse(wl.coeffs[g]))
    else:
        # round 6: alternately the numpy packer (values beyond 10 bits in `wide`: those groups are routed to their dense
        # slabs) and the library's C packer (values split into repeated entries: the frame stays in place); every other
        # such frame also hands one random group over as a dense slab or as plain pairs (per-group routing)
        from jxl_rs_amd import lib as jl
        rr = np.random.default_rng(seed)
        c_packer = (seed // 3) % 2 == 1
        other = int(rr.integers(0, ng)) if ng >= 2 and (seed // 6) % 2 == 1 else -1
        slotted = [g for g in range(ng) if g != other]
        parts = {g: (jl.host_pack_slots(wl.coeffs[g], group_id=g) if c_packer else synth.to_slots(wl.coeffs[g])) for g in slotted}
        wide = []
        for g, q in parts.items():
            if len(q[3]):
                e = q[3].copy()
                if not c_packer:
                    e[:, 0] += np.uint32(g * 3 * 65536)
                wide.append(e)
        ctx.submit_groups_slots(np.asarray(slotted, dtype=np.uint32), np.concatenate([parts[g][0] for g in slotted]),
                                np.concatenate([parts[g][1].reshape(-1) for g in slotted]),
                                np.concatenate([parts[g][2] for g in slotted]), np.concatenate(wide) if wide else None)
        if other >= 0:
            if rr.integers(0, 2):
                ctx.submit_group(other, wl.coeffs[other])
            else:
                ctx.submit_group_sparse(other, *synth.to_sparse(wl.coeffs[other]))


@pytest.mark.parametrize("seed", range(64))
def test_random_frames_and_header_parameters_bit_exact(ctx, oracle, kat, seed):
    from jxl_rs_amd import synth
    import helpers
    rng = np.random.default_rng(1000 + seed + FUZZ_OFFSET)
    w, h, mix, opts, over, arrays = _random_case(rng)
    wl = synth.make_vardct(w, h, mix=mix, seed=seed + FUZZ_OFFSET, **opts)
    # oracle
    po = helpers.oracle_params_from(oracle, wl, **over)
    _apply_arrays(po, arrays)
    lf = oracle.dequant_lf(po, *wl.lf_q)
    planes, lf_sm = oracle.vardct_frame(po, wl.coeffs, wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob,
                                        lf, wl.tables, num_threads=8)
    want = [pl[:h, :w] for pl in planes]
    # device (both submission forms alternate)
    pg = helpers.gpu_params_from(ctx, wl, **over)
    _apply_arrays(pg, arrays)
    ctx.frame_begin(pg)
    ctx.set_dequant_tables(wl.tables)
    ctx.set_lf_quantized(*wl.lf_q)
    ctx.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
    _submit_rotating(ctx, synth, wl, seed)
    ctx.slot_wait(0)
    ctx.frame_run()
    ctx.sync()
    got = ctx.read_planes()
    got_lf = ctx.read_lf()
    desc = f"{w}x{h} {opts} {sorted(over)} {sorted(arrays)} form {seed % 3}"
    for c in range(3):
        assert bit_equal(got_lf[c], lf_sm[c]), f"LF ch{c} {desc}: {diff_report(got_lf[c], lf_sm[c])}"
        assert bit_equal(got[c], want[c]), f"plane {c} {desc}: {diff_report(got[c], want[c])}"
    # 8-bit sRGB output with a random opsin matrix / bias / intensity target (XybParams, xyb.rs:147-163)
    k = kat["output_stage"]
    mat = (np.asarray(k["opsin_inverse_matrix"]) * rng.uniform(0.8, 1.2, 9)).astype(np.float32)
    bias = (np.full(3, k["opsin_bias"]) * rng.uniform(0.5, 1.5, 3)).astype(np.float32)
    xp = oracle.xyb_params(mat, bias, float(rng.choice([80.0, 255.0, 1000.0, 4000.0])))
    channels = 3 + seed % 2
    want8 = oracle.xyb_to_rgb8(xp, [np.ascontiguousarray(p) for p in want], w, h, channels)
    assert np.array_equal(ctx.read_rgb8(xp, channels), want8), f"rgb8 {desc}"


@pytest.mark.parametrize("seed", range(24))
def test_random_subsampled_frames_bit_exact(ctx, oracle, seed):
    """Chroma-subsampled frames: random per-channel shifts (each 0 or 1, any channel), ragged sizes, every 8x8
    transform, random stage lists and header parameters; ends in the YCbCr -> RGB8 output."""
    from jxl_rs_amd import synth
    import helpers
    rng = np.random.default_rng(5000 + seed + FUZZ_OFFSET)
    w, h, _, opts, over, arrays = _random_case(rng)
    hs = tuple(int(v) for v in rng.integers(0, 2, 3))
    vs = tuple(int(v) for v in rng.integers(0, 2, 3))
    if not (any(hs) or any(vs)):
        hs = (1, 0, 1)
    wl = synth.make_vardct(w, h, mix=synth.MIX_8X8, seed=seed, hshift=hs, vshift=vs, **opts)
    po = helpers.oracle_params_from(oracle, wl, **over)
    _apply_arrays(po, arrays)
    lf = [oracle.dequant_lf_channel(po, 0, wl.lf_q[1]), oracle.dequant_lf_channel(po, 1, wl.lf_q[0]),
          oracle.dequant_lf_channel(po, 2, wl.lf_q[2])]
    planes, lf_sm = oracle.vardct_frame(po, wl.coeffs, wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob,
                                        lf, wl.tables, num_threads=8)
    want = [pl[:h, :w] for pl in planes]
    pg = helpers.gpu_params_from(ctx, wl, **over)
    _apply_arrays(pg, arrays)
    pg.flags = seed % 3 == 2  # every third case on the one-kernel-per-stage filters
    ctx.frame_begin(pg)
    ctx.set_dequant_tables(wl.tables)
    ctx.set_lf_quantized(*wl.lf_q)
    ctx.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
    _submit_rotating(ctx, synth, wl, seed)
    ctx.slot_wait(0)
    ctx.frame_run()
    ctx.sync()
    got = ctx.read_planes()
    got_lf = ctx.read_lf()
    desc = f"{w}x{h} h{hs} v{vs} {opts} {sorted(over)} {sorted(arrays)} form {seed % 3}"
    for c in range(3):
        assert bit_equal(got_lf[c], lf_sm[c]), f"LF ch{c} {desc}: {diff_report(got_lf[c], lf_sm[c])}"
        assert bit_equal(got[c], want[c]), f"plane {c} {desc}: {diff_report(got[c], want[c])}"
    channels = 3 + seed % 2
    want8 = oracle.ycbcr_to_rgb8([np.ascontiguousarray(p) for p in want], w, h, channels)
    assert np.array_equal(ctx.read_ycbcr_rgb8(channels), want8), f"rgb8 {desc}"


@pytest.mark.parametrize("seed", range(48))
def test_random_feature_combinations_bit_exact(ctx, oracle, kat, seed):
    """Everything after the transforms, combined at random: chroma subsampling or not, 2x/4x/8x upsampling with
    default or custom weights and cropped target size, noise, whole frame or a band of group rows, and a random
    output transfer function at 8 or 16 bits"""
    from jxl_rs_amd import synth, lib
    import helpers
    rng = np.random.default_rng(9000 + seed)
    w, h = int(rng.integers(20, 420)), int(rng.integers(20, 560))
    sub = bool(rng.integers(0, 2))
    hs = vs = (0, 0, 0)
    if sub:
        hs, vs = tuple(int(v) for v in rng.integers(0, 2, 3)), tuple(int(v) for v in rng.integers(0, 2, 3))
        if not (any(hs) or any(vs)):
            vs = (1, 0, 1)
    ups = int(rng.choice([1, 1, 2, 4, 8])) if w * h < 60000 else int(rng.choice([1, 2]))
    noise = bool(rng.integers(0, 2))
    opts = dict(epf_iters=int(rng.integers(0, 4)), gab=bool(rng.integers(0, 2)), lf_smoothing=bool(rng.integers(0, 2)))
    wl = synth.make_vardct(w, h, mix=synth.MIX_8X8 if sub else synth.MIX_ALL, seed=seed, hshift=hs, vshift=vs, **opts)
    over = dict(ytox_lf=int(rng.integers(-20, 21)), ytob_lf=int(rng.integers(-20, 21)))
    base, _ = helpers.run_oracle_frame(oracle, wl, **over)
    ow, oh = w * ups, h * ups
    custom = None
    if ups > 1:
        if rng.random() < 0.5:  # any size whose ceil division by ups is the coded size
            ow, oh = int(rng.integers((w - 1) * ups + 1, w * ups + 1)), int(rng.integers((h - 1) * ups + 1, h * ups + 1))
        if rng.random() < 0.5:
            custom = rng.uniform(-0.05, 0.2, {2: 15, 4: 55, 8: 210}[ups]).astype(np.float32)
        base = [oracle.upsample(ups, np.ascontiguousarray(p), custom)[:oh, :ow] for p in base]
    lut = rng.uniform(0, 0.4, 8).astype(np.float32)
    vis, nonvis = int(rng.integers(0, 100)), int(rng.integers(0, 100))
    p = helpers.gpu_params_from(ctx, wl, upsampling=ups, xsize_upsampled=ow if ups > 1 else 0,
                                ysize_upsampled=oh if ups > 1 else 0, noise=int(noise), visible_frame_index=vis,
                                nonvisible_frame_index=nonvis, **over)
    for i in range(8):
        p.noise_lut[i] = float(lut[i])
    if noise:
        ytox = float(np.float32(p.base_correlation_x) + np.float32(over["ytox_lf"]) / np.float32(p.color_factor))
        ytob = float(np.float32(p.base_correlation_b) + np.float32(over["ytob_lf"]) / np.float32(p.color_factor))
        rnd = [oracle.noise_convolve(r) for r in oracle.noise_generate(vis, nonvis, ow, oh)]
        base = oracle.noise_add(lut, ytox, ytob, [np.ascontiguousarray(q) for q in base], rnd)
    ctx.set_upsampling_weights(**({f"w{ups}": custom} if custom is not None else {}))
    ctx.frame_begin(p)
    ctx.set_dequant_tables(wl.tables)
    ctx.set_lf_quantized(*wl.lf_q)
    ctx.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
    for g in range(wl.coeffs.shape[0]):
        if seed % 2:
            ctx.submit_group_sparse(g, *synth.to_sparse(wl.coeffs[g]))
        else:
            ctx.submit_group(g, wl.coeffs[g])
    ctx.slot_wait(0)
    ygroups = (h + 255) // 256
    band = ups == 1 and ygroups > 1 and rng.random() < 0.6
    r0, r1 = (int(rng.integers(0, ygroups)), ygroups) if band else (0, ygroups)
    if band:
        r1 = int(rng.integers(r0 + 1, ygroups + 1))
    ctx.frame_run(r0, r1)
    ctx.sync()
    got = ctx.read_planes()
    y0, y1 = (r0 * 256, min(h, r1 * 256)) if band else (0, oh)
    desc = f"{w}x{h} sub={hs}/{vs} ups={ups}->{ow}x{oh} custom={custom is not None} noise={noise} rows {y0}:{y1} {opts}"
    for c in range(3):
        assert bit_equal(got[c][y0:y1], base[c][y0:y1]), f"plane {c} {desc}: {diff_report(got[c][y0:y1], base[c][y0:y1])}"
    ctx.set_upsampling_weights()
    # output stage on the rows that were computed
    tf = str(rng.choice(["linear", "srgb", "bt709", "pq", "hlg", "gamma"]))
    param = {"pq": 4000.0, "hlg": -0.2, "gamma": 0.5}.get(tf, 0.0)
    bits, channels = int(rng.choice([8, 16])), int(rng.choice([3, 4]))
    lum = (0.2627, 0.678, 0.0593)
    k = kat["output_stage"]
    xp = oracle.xyb_params(k["opsin_inverse_matrix"], [k["opsin_bias"]] * 3, 255.0)
    want = oracle.xyb_to_rgb_tf(xp, tf, [np.ascontiguousarray(q) for q in base], ow, oh, channels, bits, param, lum)
    out = ctx.read_output(lib.COLOR_XYB, tf, xp, param, lum, bits, channels, y0, y1)
    assert np.array_equal(out, want[y0:y1]), f"output {tf}/{bits}/{channels} {desc}"


@pytest.mark.parametrize("seed", range(24))
def test_random_modular_ops_bit_exact(ctx, oracle, seed):
    """RCT (every op and permutation), Palette (explicit, implicit and delta entries), delta Palette with a random
    predictor and with the Weighted one, one unsqueeze step in each direction, the smooth unsqueeze kinds on a tile
    and the Modular -> RGB8 bridge on random shapes and values"""
    rng = np.random.default_rng(7000 + seed)
    h, w = int(rng.integers(1, 200)), int(rng.integers(1, 300))
    lim = int(rng.choice([256, 4096, 1 << 20]))
    planes = [rng.integers(-lim, lim, size=(h, w)).astype(np.int32) for _ in range(3)]
    op, perm = int(rng.integers(0, 7)), int(rng.integers(0, 6))
    got, want = ctx.rct(planes, op, perm), oracle.rct(planes, op, perm)
    for c in range(3):
        assert np.array_equal(got[c], want[c]), f"rct op={op} perm={perm} {h}x{w}"
    nb, bit_depth = int(rng.integers(1, 5)), int(rng.choice([8, 10, 16]))
    ncol = int(rng.integers(1, 300))
    pal = rng.integers(-50, 1 << bit_depth, size=(nb, ncol)).astype(np.int32)
    idx = rng.integers(-80, ncol + 200, size=(h, w)).astype(np.int32)
    assert np.array_equal(ctx.palette(idx, pal, ncol, nb, bit_depth), oracle.palette(idx, pal, ncol, nb, bit_depth)), "palette"
    nd = int(rng.integers(0, min(ncol, 9)))
    pred = int(rng.choice([0, 1, 2, 3, 4, 5, 7, 8, 9, 10, 11, 12, 13]))
    idx2 = idx.copy()
    idx2[rng.random((h, w)) < 0.4] = rng.integers(0, nd + 2)
    assert np.array_equal(ctx.palette_delta(idx2, pal, ncol - nd, nd, bit_depth, pred),
                          oracle.palette_delta(idx2, pal, ncol - nd, nd, bit_depth, pred)), f"delta palette pred={pred} nd={nd}"
    if w >= 2:
        avg = rng.integers(-lim, lim, size=(h, (w + 1) // 2)).astype(np.int32)
        res = rng.integers(-lim // 4 - 1, lim // 4 + 1, size=(h, w // 2)).astype(np.int32)
        assert np.array_equal(ctx.unsqueeze(True, avg, res, w, h), oracle.unsqueeze_h(avg, res, w)), "unsqueeze h"
    if h >= 2:
        avg = rng.integers(-lim, lim, size=((h + 1) // 2, w)).astype(np.int32)
        res = rng.integers(-lim // 4 - 1, lim // 4 + 1, size=(h // 2, w)).astype(np.int32)
        assert np.array_equal(ctx.unsqueeze(False, avg, res, w, h), oracle.unsqueeze_v(avg, res, h)), "unsqueeze v"
    # the Weighted-predictor branch of the palette step, random header
    hdr = [int(v) for v in rng.integers(0, 32, size=7)] + [int(v) for v in rng.integers(0, 16, size=4)]
    assert np.array_equal(ctx.palette_delta_wp(idx2, pal, ncol - nd, nd, bit_depth, hdr),
                          oracle.palette_delta_wp(idx2, pal, ncol - nd, nd, nb, bit_depth, hdr)), f"wp palette {hdr}"
    # smooth (progressive-preview) unsqueeze, the three kinds, on a tile of the channel
    for kind in (0, 1, 2):
        ah, aw = ((h + 1) // 2 if kind != 0 else h), ((w + 1) // 2 if kind != 1 else w)
        avg = rng.integers(-lim, lim, size=(ah, aw)).astype(np.int32)
        x0 = 2 * int(rng.integers(0, max(1, w // 4)))
        y0 = 2 * int(rng.integers(0, max(1, h // 4)))
        tw, th = w - x0, h - y0
        assert np.array_equal(ctx.smooth_unsqueeze(kind, avg, tw, th, x0, y0),
                              oracle.smooth_unsqueeze(kind, avg, tw, th, x0, y0)), f"smooth unsqueeze kind={kind}"
    bits = int(rng.choice([1, 2, 4, 8]))
    mult = 255 // ((1 << bits) - 1)
    rgb = ctx.modular_to_rgb8(planes, mult, 255, 3)
    for c in range(3):
        assert np.array_equal(rgb[..., c], oracle.i32_to_u8(planes[c], mult, 255))
