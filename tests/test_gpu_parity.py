"""GPU parity tests (run with -m gpu on an MI355X): every kernel, called through the C ABI,
against the CPU oracle on the same seeded inputs.

Bar: bit-exact vs the oracle's FMA build (which models the reference's AVX2 back-end; GPU
v_fma_f32 == fmaf) for all float stages, within the reference's own tolerance table vs the
unfused build and the f64 definitions; bit-exact for the integer Modular transforms.
"""
import numpy as np
import pytest

from helpers import (bit_equal, diff_report, forward_squeeze_h, forward_squeeze_v, gpu_params_from,
                     oracle_params_from, run_gpu_frame, run_oracle_frame, upload_frame)
import helpers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from jxl_rs_amd import Context
    c = Context(0, n_slots=2)
    yield c
    c.close()


def _sparse_coeffs(rng, n, size, scale=1.0, density=0.15):
    c = rng.normal(size=(n, size)).astype(np.float32) * scale
    mask = rng.random((n, size)) < density
    return np.where(mask, c, 0).astype(np.float32)


# ------------------------------------------------------------------ K1 cores, per type
@pytest.mark.parametrize("ttype", list(range(27)))
def test_transform_to_pixels_bit_exact(ctx, oracle, oracle_unfused, kat, ttype):
    cx, cy = oracle.covered_x[ttype], oracle.covered_y[ttype]
    size = cx * cy * 64
    n = 37 if size <= 1024 else (5 if size <= 16384 else 2)
    rng = np.random.default_rng(100 + ttype)
    coeffs = _sparse_coeffs(rng, n, size, density=0.3 if size <= 1024 else 0.05)
    coeffs[0] = rng.uniform(-1, 1, size).astype(np.float32)  # one dense block
    lf = rng.uniform(0, 1, size=(n, cx * cy)).astype(np.float32)
    got = ctx.stage_transform_to_pixels(ttype, coeffs, lf)
    want = np.stack([oracle.transform_to_pixels(ttype, lf[i], coeffs[i]) for i in range(n)])
    assert got.shape == want.shape
    assert bit_equal(got, want), f"type {ttype}: {diff_report(got, want)}"
    # vs the unfused (scalar back-end) build: within the reference's per-shape tolerance
    tol = {(r, c): t for r, c, t in kat["tolerances"]["idct2d"]}.get((cy * 8, cx * 8), 1e-5)
    wantu = np.stack([oracle_unfused.transform_to_pixels(ttype, lf[i], coeffs[i]) for i in range(n)])
    err = np.abs(got.astype(np.float64) - wantu)
    scale = max(1.0, float(np.abs(wantu).max()))
    assert err.max() <= 8 * tol * scale, f"type {ttype} vs unfused: {err.max()} (tol {tol}, scale {scale})"


# ------------------------------------------------------------------ stage hooks
@pytest.mark.parametrize("size", [(2, 2), (37, 23), (256, 64), (500, 300)])
def test_gaborish_stage_bit_exact(ctx, oracle, size):
    w, h = size
    rng = np.random.default_rng(w * 7 + h)
    img = rng.uniform(-1, 1, size=(h, w)).astype(np.float32)
    got = ctx.stage_gaborish(img, 0.115169525, 0.061248592)
    want = oracle.gaborish(img, 0.115169525, 0.061248592)
    assert bit_equal(got, want), diff_report(got, want)


def test_gaborish_checkerboard_golden(ctx, kat):
    g = kat["gaborish_checkerboard"]
    out = ctx.stage_gaborish(np.array(g["input"], dtype=np.float32), g["w1"], g["w2"])
    assert np.abs(out - np.array(g["output"])).max() < g["tol"]


@pytest.mark.parametrize("stage", [0, 1, 2])
@pytest.mark.parametrize("size", [(9, 7), (40, 24), (333, 129)])
def test_epf_stage_bit_exact(ctx, oracle, stage, size):
    w, h = size
    rng = np.random.default_rng(stage * 100 + w)
    planes = [rng.uniform(0, 1, size=(h, w)).astype(np.float32) * s for s in (0.05, 1.0, 1.0)]
    # sigma image like epf/test.rs: random, including values below MIN_SIGMA (passthrough)
    sig = -rng.uniform(0.05, 6.0, size=((h + 7) // 8, (w + 7) // 8)).astype(np.float32)
    po = oracle.default_params(w, h)
    pg = ctx.default_params(w, h)
    got = ctx.stage_epf(stage, pg, planes, sig)
    want = oracle.epf(stage, po, planes, sig)
    for c in range(3):
        assert bit_equal(got[c], want[c]), f"epf{stage} ch{c}: {diff_report(got[c], want[c])}"


@pytest.mark.parametrize("size", [(2, 2), (3, 3), (65, 40), (128, 128)])
def test_lf_smoothing_bit_exact(ctx, oracle, size):
    w, h = size
    rng = np.random.default_rng(w + h)
    lf = [rng.uniform(0, 1, size=(h, w)).astype(np.float32) * s for s in (0.02, 1.0, 1.0)]
    got = ctx.stage_lf_smooth(ctx.default_params(w * 8, h * 8), lf)
    want = oracle.adaptive_lf_smoothing(oracle.default_params(w * 8, h * 8), lf)
    for c in range(3):
        assert bit_equal(got[c], want[c]), f"ch{c}: {diff_report(got[c], want[c])}"


# ------------------------------------------------------------------ whole frames
FRAME_CASES = [
    # (w, h, mix, opts)
    (256, 256, "MIX_DCT8", dict(epf_iters=2, gab=True, lf_smoothing=True)),     # BASELINE config 1
    (520, 300, "MIX_D1", dict(epf_iters=0, gab=True, lf_smoothing=True)),       # config-2 style
    (520, 300, "MIX_D1", dict(epf_iters=2, gab=True, lf_smoothing=True)),       # config-3 style
    (777, 513, "MIX_ALL", dict(epf_iters=3, gab=True, lf_smoothing=True)),      # config-5 style, ragged
    (512, 512, "MIX_ALL", dict(epf_iters=1, gab=False, lf_smoothing=False)),
    (9, 9, "MIX_D1", dict(epf_iters=2, gab=True, lf_smoothing=True)),           # tiny: smoothing skipped path
    (1024, 768, "MIX_D1", dict(epf_iters=2, gab=True, lf_smoothing=True)),
    (333, 77, "MIX_D1", dict(epf_iters=2, gab=False, lf_smoothing=True)),      # fused variant without Gaborish
    (333, 77, "MIX_D1", dict(epf_iters=1, gab=True, lf_smoothing=True)),       # Gaborish + EPF1
    (66, 34, "MIX_D1", dict(epf_iters=2, gab=True, lf_smoothing=True)),        # frame edge inside a filter tile
    (3, 2, "MIX_DCT8", dict(epf_iters=2, gab=True, lf_smoothing=True)),        # frame smaller than every halo
    # epf_iters == 3: Gaborish + EPF0 kernel followed by the EPF1 + EPF2 kernel on the fused path
    (333, 77, "MIX_D1", dict(epf_iters=3, gab=False, lf_smoothing=True)),
    (66, 34, "MIX_D1", dict(epf_iters=3, gab=True, lf_smoothing=True)),
    (9, 9, "MIX_D1", dict(epf_iters=3, gab=True, lf_smoothing=True)),
    (1024, 768, "MIX_D1", dict(epf_iters=3, gab=True, lf_smoothing=True)),
]


@pytest.mark.parametrize("case", FRAME_CASES, ids=lambda c: f"{c[0]}x{c[1]}-{c[2]}-epf{c[3]['epf_iters']}")
def test_vardct_frame_bit_exact(ctx, oracle, case):
    from jxl_rs_amd import synth
    w, h, mix, opts = case
    wl = synth.make_vardct(w, h, mix=getattr(synth, mix), seed=w + h, **opts)
    want, want_lf = run_oracle_frame(oracle, wl)
    # production path (fused Gaborish+EPF kernel) and the one-kernel-per-stage path
    for flags in (0, 1):
        got, got_lf = run_gpu_frame(ctx, wl, flags=flags)
        for c in range(3):
            assert bit_equal(got_lf[c], want_lf[c]), f"flags={flags} LF ch{c}: {diff_report(got_lf[c], want_lf[c])}"
        for c in range(3):
            assert bit_equal(got[c], want[c]), f"flags={flags} plane {c}: {diff_report(got[c], want[c])}"


@pytest.mark.parametrize("filtered", [True, False])
@pytest.mark.parametrize("ttype", list(range(18, 27)))
def test_uniform_large_varblock_frame(ctx, oracle, ttype, filtered):
    """a frame tiled by ONE large type, every type of the family.  DCT64X32 / DCT32X64 varblocks cover 32 blocks but
    still take a whole slab unit each: a frame of nothing else needs nblocks / 32 units, the worst case the unit list is
    sized for (round 2 sized it for nblocks / 64 and wrote past the work-list allocation).  The types below 256 pixels
    take the one-launch route (k1_large_fused: pass 2 from LDS, results stored from registers), the 256-pixel ones the
    two passes; filtered = the 8x8-tiled plane layout the fused filter kernel reads, unfiltered = raster planes."""
    from jxl_rs_amd import synth
    wl = synth.make_vardct(512, 768, mix={ttype: 1.0}, seed=100 + ttype, epf_iters=1 if filtered else 0, gab=filtered)
    first = wl.transform_map[wl.transform_map >= 128] & 127
    assert (first == ttype).all() and first.size == 64 * 96 // (synth.COVERED_X[ttype] * synth.COVERED_Y[ttype])
    want, _ = run_oracle_frame(oracle, wl)
    got, _ = run_gpu_frame(ctx, wl)
    for c in range(3):
        assert bit_equal(got[c], want[c]), f"type {ttype} plane {c}: {diff_report(got[c], want[c])}"


def test_vardct_frame_prefilter_planes_and_determinism(ctx, oracle):
    """K1 alone (no filters) vs jxlo_decode_group, and run-to-run bit-identical output
    (result independent of launch geometry / scheduling)."""
    from jxl_rs_amd import synth
    wl = synth.make_vardct(600, 520, mix=synth.MIX_ALL, seed=9, epf_iters=0, gab=False, lf_smoothing=True)
    want, _ = run_oracle_frame(oracle, wl)
    a, _ = run_gpu_frame(ctx, wl)
    b, _ = run_gpu_frame(ctx, wl)
    for c in range(3):
        assert bit_equal(a[c], want[c]), f"plane {c}: {diff_report(a[c], want[c])}"
        assert bit_equal(a[c], b[c])


def test_vardct_frame_vs_unfused_oracle_within_tolerance(ctx, oracle_unfused):
    """Against the scalar-back-end behaviour of the reference (unfused mul_add)."""
    from jxl_rs_amd import synth
    wl = synth.make_vardct(520, 300, mix=synth.MIX_D1, seed=4)
    want, _ = run_oracle_frame(oracle_unfused, wl)
    got, _ = run_gpu_frame(ctx, wl)
    for c in range(3):
        err = np.abs(got[c].astype(np.float64) - want[c])
        assert err.max() < 2e-5, f"plane {c}: max abs err {err.max()}"


@pytest.mark.parametrize("epf_iters", [2, 3])
def test_band_runs_equal_whole_frame(ctx, oracle, epf_iters):
    """jxlh_frame_run(row0, row1): a band of group rows (multi-GPU sharding unit) produces exactly
    the rows the whole-frame run produces -- halo group rows are recomputed, nothing is exchanged."""
    from jxl_rs_amd import synth
    wl = synth.make_vardct(520, 700, mix=synth.MIX_D1, seed=31, epf_iters=epf_iters)
    want, _ = run_oracle_frame(oracle, wl)
    for flags in (0, 1):
        upload_frame(ctx, wl, flags=flags)
        for row0, row1 in ((0, 1), (1, 2), (2, 3), (1, 3)):
            ctx.frame_run(row0, row1)
            ctx.sync()
            got = ctx.read_planes()
            y0, y1 = row0 * 256, min(row1 * 256, wl.ysize)
            for c in range(3):
                assert bit_equal(got[c][y0:y1], want[c][y0:y1]), (flags, row0, row1, c, diff_report(got[c][y0:y1], want[c][y0:y1]))


@pytest.mark.parametrize("mix", ["MIX_D1", "MIX_ALL"])
def test_band_runs_of_a_slot_resident_frame(ctx, oracle, mix):
    """the same band runs on a frame resident in the slot-bucketed form (the transforms read the entries of the band's
    groups and its halo group rows; groups with special / large varblocks get their dense slab from the entries)"""
    from jxl_rs_amd import synth
    wl = synth.make_vardct(520, 700, mix=getattr(synth, mix), seed=33, epf_iters=2)
    want, _ = run_oracle_frame(oracle, wl)
    p = gpu_params_from(ctx, wl)
    ctx.frame_begin(p)
    ctx.set_dequant_tables(wl.tables)
    ctx.set_lf_quantized(*wl.lf_q)
    ctx.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
    ng = wl.coeffs.shape[0]
    parts = [synth.to_slots(wl.coeffs[g]) for g in range(ng)]
    assert all(len(q[3]) == 0 for q in parts)
    ctx.submit_groups_slots(np.arange(ng, dtype=np.uint32), np.concatenate([q[0] for q in parts]),
                            np.concatenate([q[1].reshape(-1) for q in parts]), np.concatenate([q[2] for q in parts]), None)
    ctx.slot_wait(0)
    for row0, row1 in ((0, 1), (1, 2), (2, 3), (1, 3), (0, 3)):
        ctx.frame_run(row0, row1)
        ctx.sync()
        got = ctx.read_planes()
        y0, y1 = row0 * 256, min(row1 * 256, wl.ysize)
        for c in range(3):
            assert bit_equal(got[c][y0:y1], want[c][y0:y1]), (row0, row1, c, diff_report(got[c][y0:y1], want[c][y0:y1]))


def test_invalid_transform_id_is_reported(ctx):
    from jxl_rs_amd import synth, JxlHipError
    from jxl_rs_amd import lib
    wl = synth.make_vardct(256, 256, mix=synth.MIX_DCT8, seed=1)
    wl.transform_map = wl.transform_map.copy()
    wl.transform_map[3, 3] = 0x80 | 40
    upload_frame(ctx, wl)
    ctx.frame_run()
    with pytest.raises(JxlHipError) as e:
        ctx.sync()
    assert e.value.status == lib.ERR_INVALID_TRANSFORM


# ---------------------------------------------------------------- chroma-subsampled (JPEG-recompression) frames
SUBSAMPLINGS = {
    "420": ((1, 0, 1), (1, 0, 1)),
    "422": ((1, 0, 1), (0, 0, 0)),
    "440": ((0, 0, 0), (1, 0, 1)),
    "mixed": ((1, 0, 0), (0, 0, 1)),   # Cb halved horizontally, Cr vertically
    "luma": ((0, 1, 0), (0, 1, 0)),    # any channel may be the sub-sampled one
}
SUB_CASES = [
    (64, 64, "420", dict(epf_iters=0, gab=False, lf_smoothing=False)),      # what a recompressed JPEG looks like
    (300, 270, "420", dict(epf_iters=0, gab=False, lf_smoothing=False)),    # two group columns, ragged
    (515, 389, "420", dict(epf_iters=2, gab=True, lf_smoothing=True)),      # filters after the upsampling (tiled K1)
    (515, 389, "422", dict(epf_iters=0, gab=False, lf_smoothing=True)),
    (333, 77, "440", dict(epf_iters=1, gab=True, lf_smoothing=False)),
    (257, 263, "mixed", dict(epf_iters=3, gab=True, lf_smoothing=True)),
    (97, 131, "luma", dict(epf_iters=2, gab=False, lf_smoothing=False)),
    (9, 7, "420", dict(epf_iters=2, gab=True, lf_smoothing=True)),          # chroma is a single block
    (2100, 40, "420", dict(epf_iters=0, gab=False, lf_smoothing=False)),    # crosses an LF group: corner-packed LF
]


@pytest.mark.parametrize("case", SUB_CASES, ids=lambda c: f"{c[0]}x{c[1]}-{c[2]}-epf{c[3]['epf_iters']}")
def test_subsampled_frame_bit_exact(ctx, oracle, case):
    """K1e (frame/group.rs:223-250, :443-504) + chroma upsampling (render/stages/chroma_upsample.rs) + the
    filter chain, dense and sparse submission, fused and per-stage filters"""
    from jxl_rs_amd import synth
    w, h, sub, opts = case
    hs, vs = SUBSAMPLINGS[sub]
    wl = synth.make_vardct(w, h, mix=synth.MIX_8X8, seed=w + 3 * h, hshift=hs, vshift=vs, **opts)
    want, want_lf = run_oracle_frame(oracle, wl)
    for flags in (0, 1):
        got, got_lf = run_gpu_frame(ctx, wl, flags=flags)
        for c in range(3):
            assert bit_equal(got_lf[c], want_lf[c]), f"flags={flags} LF ch{c}: {diff_report(got_lf[c], want_lf[c])}"
            assert bit_equal(got[c], want[c]), f"flags={flags} plane {c}: {diff_report(got[c], want[c])}"
    # sparse submission: the transforms read the pairs
    p = gpu_params_from(ctx, wl)
    ctx.frame_begin(p)
    ctx.set_dequant_tables(wl.tables)
    ctx.set_lf_quantized(*wl.lf_q)
    ctx.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
    for g in range(wl.coeffs.shape[0]):
        pairs, n, wide = synth.to_sparse(wl.coeffs[g])
        ctx.submit_group_sparse(g, pairs, n, wide)
    ctx.slot_wait(0)
    ctx.frame_run()
    ctx.sync()
    got = ctx.read_planes()
    for c in range(3):
        assert bit_equal(got[c], want[c]), f"sparse plane {c}: {diff_report(got[c], want[c])}"


def test_subsampled_frame_rejects_large_varblocks(ctx):
    """Error::InvalidBlockSizeForChromaSubsampling (frame/modular/mod.rs:1058-1060)"""
    from jxl_rs_amd import synth, lib, JxlHipError
    wl = synth.make_vardct(128, 128, mix=synth.MIX_D1, seed=2, epf_iters=0, gab=False)
    wl.opts["hshift"], wl.opts["vshift"] = (1, 0, 1), (1, 0, 1)
    upload_frame(ctx, wl)
    ctx.frame_run()
    with pytest.raises(JxlHipError) as e:
        ctx.sync()
    assert e.value.status == lib.ERR_INVALID_BLOCK_SIZE
    p = ctx.default_params(64, 64)
    p.hshift[0] = 2
    with pytest.raises(JxlHipError) as e:
        ctx.frame_begin(p)
    assert e.value.status == lib.ERR_INVALID_ARGUMENT


def test_varblock_out_of_bounds_is_reported(ctx):
    """Error::HFBlockOutOfBounds (frame/modular/mod.rs:1061-1064): a first-block flag whose varblock would cross
    the frame / group edge is reported and not reconstructed (no out-of-plane stores)."""
    from jxl_rs_amd import synth, lib, JxlHipError
    wl = synth.make_vardct(300, 260, mix=synth.MIX_DCT8, seed=5, epf_iters=0, gab=False)
    # a DCT32x32 (type 5) whose top-left block is the last block of the first group row/column: crosses the group edge
    bad = wl.transform_map.copy()
    bad[31, 31] = 128 | 5
    upload_frame(ctx, wl)
    ctx.set_hf_meta(bad, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
    ctx.frame_run()
    with pytest.raises(JxlHipError) as e:
        ctx.sync()
    assert e.value.status == lib.ERR_BLOCK_OUT_OF_BOUNDS
    # crossing the frame's right edge (300 px = 38 blocks; block column 37 is the last)
    bad = wl.transform_map.copy()
    bad[2, 37] = 128 | 6  # DCT16X8: covered_x = 1, covered_y = 2 -> fine; 7 = DCT8X16: covered_x = 2 -> out
    bad[3, 37] = 6
    upload_frame(ctx, wl)
    ctx.set_hf_meta(bad, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
    ctx.frame_run()
    ctx.sync()  # legal
    bad[2, 37] = 128 | 7
    bad[3, 37] = 128 | 0
    upload_frame(ctx, wl)
    ctx.set_hf_meta(bad, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
    ctx.frame_run()
    with pytest.raises(JxlHipError) as e:
        ctx.sync()
    assert e.value.status == lib.ERR_BLOCK_OUT_OF_BOUNDS
    # a group whose first-block flags claim more than its 1024 blocks (overlapping 32x32 varblocks everywhere)
    bad = wl.transform_map.copy()
    bad[:28, :28] = 128 | 5
    upload_frame(ctx, wl)
    ctx.set_hf_meta(bad, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
    ctx.frame_run()
    with pytest.raises(JxlHipError) as e:
        ctx.sync()
    assert e.value.status == lib.ERR_BLOCK_OUT_OF_BOUNDS
    # the context is still usable
    got, _ = run_gpu_frame(ctx, wl)
    assert np.isfinite(got[1]).all()


def test_unset_map_rects_read_as_empty(ctx, oracle):
    """jxlh_frame_begin clears the HfMetadata maps: a rect the caller never sets holds no varblocks, whatever an
    earlier (larger) frame left in the buffers."""
    from jxl_rs_amd import synth
    big = synth.make_vardct(600, 520, mix=synth.MIX_ALL, seed=9, epf_iters=0, gab=False)
    run_gpu_frame(ctx, big)
    wl = synth.make_vardct(300, 260, mix=synth.MIX_D1, seed=4, epf_iters=0, gab=False, lf_smoothing=False)
    p = gpu_params_from(ctx, wl)
    ctx.frame_begin(p)
    ctx.set_dequant_tables(wl.tables)
    ctx.set_lf_quantized(*wl.lf_q)
    # only the first group row's maps arrive
    ctx.set_hf_meta(wl.transform_map[:32], wl.raw_quant[:32], wl.epf_map[:32], wl.ytox[:4], wl.ytob[:4])
    for g in range(wl.coeffs.shape[0]):
        ctx.submit_group(g, wl.coeffs[g])
    ctx.slot_wait(0)
    ctx.frame_run()
    ctx.sync()  # no error: the second group row simply holds no work
    got = ctx.read_planes()
    want, _ = run_oracle_frame(oracle, wl)
    for c in range(3):
        assert bit_equal(got[c][:256], want[c][:256]), f"plane {c}"


@pytest.mark.parametrize("shape", [(1, 3), (3, 1), (1, 1), (37, 53), (256, 300), (5, 1024)])
def test_chroma_upsample_stage_bit_exact(ctx, oracle, kat, shape):
    """HorizontalChromaUpsample / VerticalChromaUpsample hooks vs the oracle (and the reference's own vectors)"""
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    plane = rng.standard_normal(shape).astype(np.float32)
    for horizontal in (True, False):
        got = ctx.stage_chroma_upsample(plane, horizontal)
        assert bit_equal(got, oracle.chroma_upsample(plane, horizontal)), f"horizontal={horizontal}"
    k = kat["chroma_upsample"]
    assert np.array_equal(ctx.stage_chroma_upsample(np.array([k["input"]], np.float32), True)[0], np.float32(k["expected"]))
    assert np.array_equal(ctx.stage_chroma_upsample(np.array([k["input"]], np.float32).T, False)[:, 0], np.float32(k["expected"]))


# ---------------------------------------------------------------- 2x / 4x / 8x upsampling
@pytest.mark.parametrize("n", [2, 4, 8])
@pytest.mark.parametrize("shape", [(7, 7), (1, 1), (3, 70), (130, 67)])
def test_upsample_stage_bit_exact(ctx, oracle, kat, n, shape):
    """Upsample2x/4x/8x hook vs the oracle, default and custom weights; plus the reference's own properties
    (render/stages/upsample.rs:516-852): a constant plane stays constant, an impulse paints the kernel"""
    rng = np.random.default_rng(n * 100 + shape[0])
    plane = rng.standard_normal(shape).astype(np.float32)
    ctx.set_upsampling_weights()
    assert bit_equal(ctx.stage_upsample(n, plane), oracle.upsample(n, plane))
    const = np.full(shape, 0.777, np.float32)
    assert np.max(np.abs(ctx.stage_upsample(n, const) - np.float32(0.777))) <= kat["upsampling"]["constant_tol"][str(n)]
    # custom weights (CustomTransformData), then back to the defaults
    cnt = {2: 15, 4: 55, 8: 210}[n]
    w = rng.uniform(-0.1, 0.3, cnt).astype(np.float32)
    ctx.set_upsampling_weights(**{f"w{n}": w})
    assert bit_equal(ctx.stage_upsample(n, plane), oracle.upsample(n, plane, w))
    ctx.set_upsampling_weights()
    assert bit_equal(ctx.stage_upsample(n, plane), oracle.upsample(n, plane))


@pytest.mark.parametrize("n,size,up_size", [(2, (300, 270), None), (4, (77, 33), (305, 130)), (8, (40, 40), (313, 320)),
                                            (2, (515, 389), (1029, 777))])
def test_upsampled_frame_bit_exact(ctx, oracle, kat, n, size, up_size):
    """frame_header.upsampling: the stages run after the filters on the three colour channels
    (frame/render.rs:655-671); planes and the RGB8 output come back at the upsampled size"""
    from jxl_rs_amd import synth
    w, h = size
    wl = synth.make_vardct(w, h, mix=synth.MIX_D1, seed=w + h + n, epf_iters=2)
    want, _ = run_oracle_frame(oracle, wl)
    ow, oh = up_size if up_size else (w * n, h * n)
    want_up = [oracle.upsample(n, np.ascontiguousarray(p))[:oh, :ow] for p in want]
    over = dict(upsampling=n)
    if up_size:
        over.update(xsize_upsampled=ow, ysize_upsampled=oh)
    upload_frame(ctx, wl, **over)
    ctx.frame_run()
    ctx.sync()
    got = ctx.read_planes()
    for c in range(3):
        assert got[c].shape == (oh, ow)
        assert bit_equal(got[c], want_up[c]), f"plane {c}: {diff_report(got[c], want_up[c])}"
    params = _default_xyb_params(oracle, kat)
    want8 = oracle.xyb_to_rgb8(params, [np.ascontiguousarray(p) for p in want_up], ow, oh, 3)
    assert np.array_equal(ctx.read_rgb8(params, 3), want8)


@pytest.mark.parametrize("n,size,up_size,strip", [(2, (300, 270), (599, 539), False), (1, (264, 200), None, False),
                                                  (4, (77, 33), (305, 130), False), (2, (264, 200), None, True)])
def test_extra_channels_inside_the_frame_path(ctx, oracle, n, size, up_size, strip):
    """channels 3.. of the reference's pipeline (frame/render.rs:564-567, :624-637, :655-671): handed over as decoded
    Modular channels, converted (ConvertModularToF32Stage with the channel's bit depth) and upsampled by their OWN
    factor -- equal to the frame's (the late case), larger, or none -- by the same jxlh_frame_run that renders the colour
    channels, through the two-kernel path and through the strip kernel"""
    from jxl_rs_amd import synth, JxlHipError
    w, h = size
    wl = synth.make_vardct(w, h, mix=synth.MIX_D1, seed=w + h + n, epf_iters=2, aligned=True)
    ow, oh = up_size if up_size else (w * n, h * n)
    over = dict(upsampling=n) if n > 1 else {}
    if up_size:
        over.update(xsize_upsampled=ow, ysize_upsampled=oh)
    if strip:
        over["flags"] = 4  # JXLH_FRAME_STRIP
    upload_frame(ctx, wl, **over)
    rng = np.random.default_rng(w * 3 + n)
    chans = []
    for ec, (factor, bits) in enumerate(((max(n, 1), 8), (8, 12), (1, 16))):
        cw, ch = -(-ow // factor), -(-oh // factor)
        smp = rng.integers(0, 1 << bits, size=(ch, cw)).astype(np.int32)
        ctx.set_extra_channel(ec, smp, bits, factor)
        f32 = oracle.modular_to_f32(smp, bits)
        want = oracle.upsample(factor, f32)[:oh, :ow] if factor > 1 else f32
        chans.append((ec, want))
    with pytest.raises(JxlHipError):
        ctx.read_extra_channel(0, ow, oh)          # handed over, not rendered yet
    ctx.frame_run()
    ctx.sync()
    assert ctx.frame_path()[0] == strip
    for ec, want in chans:
        got = ctx.read_extra_channel(ec, want.shape[1], want.shape[0])
        assert bit_equal(got, want), f"extra channel {ec}: {diff_report(got, want)}"
    # the colour channels are what they are without extra channels
    want_c, _ = run_oracle_frame(oracle, wl)
    got_c = ctx.read_planes()
    for c in range(3):
        ref = oracle.upsample(n, np.ascontiguousarray(want_c[c]))[:oh, :ow] if n > 1 else want_c[c]
        assert bit_equal(got_c[c], ref)
    with pytest.raises(JxlHipError):
        ctx.set_extra_channel(8, np.zeros((4, 4), np.int32), 8, 1)      # index out of range
    with pytest.raises(JxlHipError):
        ctx.set_extra_channel(0, np.zeros((4, 4), np.int32), 8, 3)      # not a factor the format has


@pytest.mark.parametrize("bits,exp_bits", [(16, 5), (32, 8), (19, 6), (12, 4), (24, 7)])
def test_float_samples_to_f32(ctx, oracle, bits, exp_bits):
    """ConvertModularToF32Stage on floating-point samples (the extra-channel path modelled integer samples
    only): a `bits`-bit float with exp_bits exponent bits stored in an integer -> binary32, int_to_float (convert.rs:
    416-486), through jxlh_modular_to_f32 and as an extra channel of a frame -- also one handed over AFTER the first
    render, which a partial re-render must pick up"""
    from jxl_rs_amd import synth, JxlHipError
    rng = np.random.default_rng(bits * 9 + exp_bits)
    n = 4096
    smp = rng.integers(0, 1 << min(bits, 31), size=n, dtype=np.int64)
    if bits == 32:
        smp = rng.integers(0, 1 << 32, size=n, dtype=np.int64)
    mant = bits - exp_bits - 1
    special = np.array([0, 1 << (bits - 1), 1, (1 << mant) - 1, ((1 << exp_bits) - 1) << mant,   # +-0, subnormals, inf
                        (((1 << exp_bits) - 1) << mant) | 1, 1 << mant, (1 << (bits - 1)) | 5], dtype=np.int64)  # NaN, min normal
    smp[:len(special)] = special
    smp = smp.astype(np.uint32).view(np.int32)
    want = oracle.modular_to_f32(smp, bits, exp_bits)
    if (bits, exp_bits) == (16, 5):
        ref = smp.astype(np.uint16).view(np.float16).astype(np.float32)   # what the reference's f16 fast path computes
        ok = np.isnan(ref) | (ref.view(np.uint32) == want.view(np.uint32))
        assert ok.all()
    if (bits, exp_bits) == (32, 8):
        assert np.array_equal(want.view(np.int32), smp)                    # the passthrough
    got = ctx.modular_to_f32(smp, bits, exp_bits)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    with pytest.raises(JxlHipError):
        ctx.modular_to_f32(smp, bits, 9)        # not a float format (more than 8 exponent bits)
    if bits > 31:
        return
    # as an extra channel, handed over after the frame has been rendered once
    wl = synth.make_vardct(256, 256, mix=synth.MIX_DCT8, seed=bits, epf_iters=1)
    upload_frame(ctx, wl)
    ctx.frame_run()
    ctx.sync()
    plane = smp.reshape(64, 64)
    ctx.set_extra_channel(1, plane, bits, 1, exp_bits=exp_bits)
    ctx.rerender_groups([0])
    ctx.sync()
    got = ctx.read_extra_channel(1, 64, 64)
    assert np.array_equal(got.view(np.uint32), want.reshape(64, 64).view(np.uint32))


def test_upsampled_frame_argument_errors(ctx):
    from jxl_rs_amd import synth, lib, JxlHipError
    wl = synth.make_vardct(600, 600, mix=synth.MIX_DCT8, seed=1, epf_iters=0, gab=False)
    for over in (dict(upsampling=3), dict(upsampling=2, xsize_upsampled=1300), dict(upsampling=4, ysize_upsampled=100)):
        with pytest.raises(JxlHipError) as e:
            ctx.frame_begin(gpu_params_from(ctx, wl, **over))
        assert e.value.status == lib.ERR_INVALID_ARGUMENT
    upload_frame(ctx, wl, upsampling=2)
    with pytest.raises(JxlHipError) as e:
        ctx.frame_run(0, 1)        # a band of an upsampled frame
    assert e.value.status == lib.ERR_UNSUPPORTED


# ---------------------------------------------------------------- noise synthesis
def _oracle_add_noise(oracle, planes, lut, ytox, ytob, visible, nonvisible):
    h, w = planes[0].shape
    rnd = [oracle.noise_convolve(r) for r in oracle.noise_generate(visible, nonvisible, w, h)]
    return oracle.noise_add(lut, ytox, ytob, [np.ascontiguousarray(p) for p in planes], rnd)


@pytest.mark.parametrize("shape", [(1, 1), (40, 17), (256, 256), (300, 700), (513, 258)])
def test_noise_generation_bit_exact(ctx, oracle, kat, shape):
    """the random planes: one xorshift128+ stream per 256x256 tile, entered in parallel through GF(2) jumps,
    must be the reference's sequential stream bit for bit (oracle pinned on xorshift128plus.rs' goldens)"""
    h, w = shape
    for visible, nonvisible in ((0, 0), (3, 1), (0xFFFFFFFF, 12345)):
        got = ctx.stage_noise_generate(visible, nonvisible, w, h)
        want = oracle.noise_generate(visible, nonvisible, w, h)
        for c in range(3):
            assert bit_equal(got[c], want[c]), f"seed ({visible},{nonvisible}) channel {c}: {diff_report(got[c], want[c])}"
    assert got[0].min() >= 1.0 and got[0].max() < 2.0


@pytest.mark.parametrize("shape", [(2, 2), (1, 1), (5, 3), (67, 130), (300, 258)])
def test_noise_convolve_and_add_bit_exact(ctx, oracle, kat, shape):
    k = kat["noise"]
    rng = np.random.default_rng(shape[0] + 31 * shape[1])
    plane = rng.uniform(1.0, 2.0, shape).astype(np.float32)
    assert bit_equal(ctx.stage_noise_convolve(plane), oracle.noise_convolve(plane))
    if shape == (2, 2):  # render/stages/noise.rs:205-217
        got = ctx.stage_noise_convolve(np.array(k["convolve_input"], np.float32).reshape(2, 2))
        assert np.max(np.abs(got.ravel() - np.float32(k["convolve_expected"]))) <= k["convolve_tol"]
    planes = [rng.uniform(-0.2, 0.9, shape).astype(np.float32) for _ in range(3)]
    rnd = [rng.standard_normal(shape).astype(np.float32) for _ in range(3)]
    p = ctx.default_params(8, 8)
    lut = rng.uniform(0.0, 1.0, 8).astype(np.float32)
    for i in range(8):
        p.noise_lut[i] = float(lut[i])
    p.ytox_lf, p.ytob_lf = 7, -3
    ytox = float(np.float32(p.base_correlation_x) + np.float32(7) / np.float32(p.color_factor))
    ytob = float(np.float32(p.base_correlation_b) + np.float32(-3) / np.float32(p.color_factor))
    got = ctx.stage_noise_add(p, planes, rnd)
    want = oracle.noise_add(lut, ytox, ytob, planes, rnd)
    for c in range(3):
        assert bit_equal(got[c], want[c]), f"add channel {c}: {diff_report(got[c], want[c])}"


def test_noise_add_libjxl_golden(ctx, kat):
    """render/stages/noise.rs:228-325: golden data generated by libjxl, through the device hook"""
    k = kat["noise"]
    a = (np.float32(k["add_input_start"]) + np.float32(k["add_input_step"]) * np.arange(64, dtype=np.float32)).reshape(8, 8)
    p = ctx.default_params(8, 8)
    for i in range(8):
        p.noise_lut[i] = k["add_lut"][i]
    got = ctx.stage_noise_add(p, [a, a, a], [a, a, a])
    for c in range(3):
        assert np.max(np.abs(got[c].ravel() - np.float32(k["add_expected"][c]))) <= k["add_tol"]


@pytest.mark.parametrize("size,ups", [((300, 270), 1), ((520, 300), 1), ((77, 33), 4), ((9, 9), 1)])
def test_noisy_frame_bit_exact(ctx, oracle, kat, size, ups):
    """has_noise frames: generation + ConvolveNoise x3 + AddNoise after the filters (and the upsampling)"""
    from jxl_rs_amd import synth
    w, h = size
    wl = synth.make_vardct(w, h, mix=synth.MIX_D1, seed=w * 3 + h, epf_iters=2)
    base, _ = run_oracle_frame(oracle, wl)
    if ups > 1:
        base = [oracle.upsample(ups, np.ascontiguousarray(p)) for p in base]
    lut = np.float32([0.02, 0.05, 0.1, 0.2, 0.15, 0.1, 0.05, 0.3])
    po = oracle_params = None
    p = gpu_params_from(ctx, wl, upsampling=ups, noise=1, visible_frame_index=2, nonvisible_frame_index=5, ytox_lf=4,
                        ytob_lf=-9)
    for i in range(8):
        p.noise_lut[i] = float(lut[i])
    ytox = float(np.float32(p.base_correlation_x) + np.float32(4) / np.float32(p.color_factor))
    ytob = float(np.float32(p.base_correlation_b) + np.float32(-9) / np.float32(p.color_factor))
    # ytox_lf / ytob_lf also enter the LF dequantisation: same override on the oracle side
    base, _ = run_oracle_frame(oracle, wl, ytox_lf=4, ytob_lf=-9)
    if ups > 1:
        base = [oracle.upsample(ups, np.ascontiguousarray(q)) for q in base]
    want = _oracle_add_noise(oracle, base, lut, ytox, ytob, 2, 5)
    ctx.frame_begin(p)
    ctx.set_dequant_tables(wl.tables)
    ctx.set_lf_quantized(*wl.lf_q)
    ctx.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
    for g in range(wl.coeffs.shape[0]):
        ctx.submit_group(g, wl.coeffs[g])
    ctx.slot_wait(0)
    ctx.frame_run()
    ctx.sync()
    got = ctx.read_planes()
    for c in range(3):
        assert bit_equal(got[c], want[c]), f"plane {c}: {diff_report(got[c], want[c])}"
    assert not bit_equal(got[1], base[1]), "the noise must have changed the image"
    if ups == 1 and h > 256:  # bands: the same rows as the whole-frame run
        ctx.frame_run(1, 2)
        ctx.sync()
        band = ctx.read_planes()
        for c in range(3):
            assert bit_equal(band[c][256:h], want[c][256:h])
    # an all-zero LUT leaves the frame alone (noise.rs:153-155)
    for i in range(8):
        p.noise_lut[i] = 0.0
    ctx.frame_begin(p)
    ctx.set_dequant_tables(wl.tables)
    ctx.set_lf_quantized(*wl.lf_q)
    ctx.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
    for g in range(wl.coeffs.shape[0]):
        ctx.submit_group(g, wl.coeffs[g])
    ctx.slot_wait(0)
    ctx.frame_run()
    ctx.sync()
    quiet = ctx.read_planes()
    for c in range(3):
        assert bit_equal(quiet[c], base[c])


@pytest.mark.parametrize("size,sub,channels", [((300, 270), "420", 3), ((97, 131), "422", 4), ((64, 64), "440", 3),
                                               ((33, 21), "mixed", 3), ((130, 9), "luma", 4), ((3, 4), "420", 3)])
def test_ycbcr_output_bit_exact(ctx, oracle, size, sub, channels):
    """a recompressed JPEG end to end: K1e, chroma upsampling, YcbcrToRgbStage, ConvertF32ToU8/U16"""
    from jxl_rs_amd import synth
    w, h = size
    hs, vs = SUBSAMPLINGS[sub]
    wl = synth.make_vardct(w, h, mix=synth.MIX_8X8, seed=w ^ h, epf_iters=0, gab=False, lf_smoothing=False, hshift=hs, vshift=vs)
    wl.lf_q[0] = wl.lf_q[0] // 3   # keep Y + 128/255 inside [0, 1] so the clamps are not the whole story
    want_planes, _ = run_oracle_frame(oracle, wl)
    want8 = oracle.ycbcr_to_rgb8(want_planes, w, h, channels)
    want16 = oracle.ycbcr_to_rgb16(want_planes, w, h, channels)
    upload_frame(ctx, wl)
    ctx.frame_run()
    ctx.sync()
    got8 = ctx.read_ycbcr_rgb8(channels)
    bad = np.argwhere(got8 != want8)
    assert bad.size == 0, f"{len(bad)} differing bytes, first at {bad[0]}"
    assert np.array_equal(ctx.read_ycbcr_rgb16(channels), want16)
    y0, y1 = h // 3, (2 * h) // 3
    assert np.array_equal(ctx.read_ycbcr_rgb8(channels, y0, y1), want8[y0:y1])
    assert len(np.unique(want8)) > 16
    # the calls above took the sub-sampled channels straight from the transforms (no stage follows them in this
    # frame, so the full-resolution chroma planes were never built); asking for the planes builds them, and the
    # output calls then go through the general kernel: same bytes either way
    got_planes = ctx.read_planes()
    for c in range(3):
        assert bit_equal(got_planes[c], want_planes[c]), f"plane {c}: {diff_report(got_planes[c], want_planes[c])}"
    assert np.array_equal(ctx.read_ycbcr_rgb8(channels), want8)
    assert np.array_equal(ctx.read_ycbcr_rgb16(channels), want16)


def test_call_order_errors(ctx):
    from jxl_rs_amd import lib, Context
    c2 = Context(0, 1)
    try:
        assert c2.L.jxlh_frame_run(c2._ctx, 0, 1) == lib.ERR_BAD_STATE
        assert c2.L.jxlh_submit_group(c2._ctx, 0, 0, c2._ctx, 1) == lib.ERR_BAD_STATE
        p = c2.default_params(64, 64)
        c2.frame_begin(p)
        assert c2.L.jxlh_frame_run(c2._ctx, 0, 1) == lib.ERR_BAD_STATE  # no dequant tables yet
        assert c2.L.jxlh_submit_group(c2._ctx, 5, 0, c2._ctx, 1) == lib.ERR_INVALID_ARGUMENT  # bad slot
        assert c2.L.jxlh_submit_group(c2._ctx, 0, 99, c2._ctx, 1) == lib.ERR_INVALID_ARGUMENT  # bad group
        # progressive passes (flags without COMPLETE) are accepted (tests/test_gpu_progressive.py); what a dense slab
        # cannot do is accumulate on the device
        assert c2.L.jxlh_submit_group(c2._ctx, 0, 0, c2._ctx, lib.GROUP_ACCUMULATE) == lib.ERR_INVALID_ARGUMENT
    finally:
        c2.close()


# ------------------------------------------------------------------ Modular
@pytest.mark.parametrize("op", range(7))
def test_rct_bit_exact(ctx, oracle, op):
    rng = np.random.default_rng(op)
    n = 10007
    planes = [rng.integers(-70000, 70000, size=n).astype(np.int32) for _ in range(3)]
    planes[0][:4] = [2**31 - 1, -2**31, 2**31 - 1, 0]  # wrapping arithmetic
    planes[1][:4] = [2**31 - 1, -2**31, 1, -1]
    for perm in range(6):
        got = ctx.rct(planes, op, perm)
        want = oracle.rct(planes, op, perm)
        for c in range(3):
            assert np.array_equal(got[c], want[c]), (op, perm, c)


@pytest.mark.parametrize("bit_depth", [8, 12, 16])
def test_palette_bit_exact(ctx, oracle, bit_depth):
    rng = np.random.default_rng(bit_depth)
    ncol = 200
    pal = rng.integers(0, 1 << bit_depth, size=(3, ncol)).astype(np.int32)
    idx = rng.integers(-150, ncol + 64 + 130, size=(61, 47)).astype(np.int32)
    got = ctx.palette(idx, pal, ncol, 3, bit_depth)
    want = oracle.palette(idx, pal, ncol, 3, bit_depth)
    assert np.array_equal(got, want)
    got1 = ctx.palette(idx, pal[:1], ncol, 1, bit_depth)
    assert np.array_equal(got1[0], want[0])


@pytest.mark.parametrize("ncol", [1, 4096, 5000])
def test_palette_sizes_lds_and_global_paths(ctx, oracle, ncol):
    """up to 12288 palette entries are staged in LDS; larger palettes are gathered from global memory"""
    rng = np.random.default_rng(ncol)
    pal = rng.integers(0, 256, size=(3, ncol)).astype(np.int32)
    idx = rng.integers(-20, ncol + 64 + 130, size=(129, 515)).astype(np.int32)
    assert np.array_equal(ctx.palette(idx, pal, ncol, 3, 8), oracle.palette(idx, pal, ncol, 3, 8))


@pytest.mark.parametrize("predictor", [0, 1, 2, 3, 4, 5, 7, 8, 9, 10, 11, 12, 13])
def test_delta_palette_bit_exact(ctx, oracle, predictor):
    """do_palette_step_general with delta entries and every neighbour predictor (palette.rs:228-251,
    modular/predict.rs:152-198): the skewed-wavefront kernel vs the raster-order oracle, incl. implicit (negative and
    beyond-the-palette) indices, images narrower than the dependency reach, and one taller than a row band"""
    rng = np.random.default_rng(100 + predictor)
    for (h, w), nb, bit_depth in (((37, 53), 3, 8), ((1, 40), 3, 8), ((50, 1), 1, 8), ((9, 2), 4, 12), ((1100, 7), 3, 8),
                                  ((130, 300), 3, 10)):
        num_colors, num_deltas = int(rng.integers(1, 20)), int(rng.integers(0, 8))
        pal = rng.integers(-30, 1 << bit_depth, size=(nb, num_colors + num_deltas)).astype(np.int32)
        pal[:, :num_deltas] = rng.integers(-12, 13, size=(nb, num_deltas))   # delta entries are small steps
        idx = rng.integers(-8, num_colors + num_deltas + 100, size=(h, w)).astype(np.int32)
        idx[rng.random((h, w)) < 0.5] = rng.integers(0, max(1, num_deltas + 2))   # plenty of predicted pixels
        got = ctx.palette_delta(idx, pal, num_colors, num_deltas, bit_depth, predictor)
        want = oracle.palette_delta(idx, pal, num_colors, num_deltas, bit_depth, predictor)
        assert np.array_equal(got, want), f"{h}x{w} nb={nb} colors={num_colors} deltas={num_deltas}"
    # without deltas and with the Zero predictor this is the plain gather
    if predictor == 0:
        pal = rng.integers(0, 256, size=(3, 16)).astype(np.int32)
        idx = rng.integers(-4, 200, size=(20, 30)).astype(np.int32)
        assert np.array_equal(ctx.palette_delta(idx, pal, 16, 0, 8, 0), ctx.palette(idx, pal, 16, 3, 8))


@pytest.mark.parametrize("shape", [(1, 1), (37, 53), (300, 258)])
def test_modular_output_bridges_bit_exact(ctx, oracle, shape):
    """ConvertI32ToU8 (interleaved), ConvertModularToF32 and ConvertModularXYBToF32 (render/stages/convert.rs)"""
    rng = np.random.default_rng(shape[0] * 7 + shape[1])
    planes = [rng.integers(-40, 300, size=shape).astype(np.int32) for _ in range(3)]
    planes[0].flat[0] = 2**31 - 1   # the multiply wraps like the reference's i32 lanes
    for bits, channels in ((8, 3), (4, 4), (2, 3), (1, 4)):
        mult, maxv = 255 // ((1 << bits) - 1), 255
        got = ctx.modular_to_rgb8(planes, mult, maxv, channels)
        for c in range(3):
            assert np.array_equal(got[..., c], oracle.i32_to_u8(planes[c], mult, maxv)), (bits, c)
        if channels == 4:
            assert (got[..., 3] == 255).all()
    for bits in (1, 8, 12, 16, 24, 32):
        assert bit_equal(ctx.modular_to_f32(planes[1], bits), oracle.modular_to_f32(planes[1], bits)), bits
    q = np.float32([1.0 / 4096, 1.0 / 512, 1.0 / 256])
    got = ctx.modular_xyb_to_f32(planes[0], planes[1], planes[2], q)
    want = oracle.modular_xyb_to_f32(planes[0], planes[1], planes[2], q)
    for c in range(3):
        assert bit_equal(got[c], want[c]), c


def test_delta_palette_weighted_predictor_needs_its_header(ctx):
    """predictor 6 through jxlh_palette_delta has no WeightedHeader to run with: the caller is pointed at
    jxlh_palette_delta_wp; header fields outside their bit widths are rejected there"""
    from jxl_rs_amd import lib, JxlHipError
    with pytest.raises(JxlHipError) as e:
        ctx.palette_delta(np.zeros((4, 4), np.int32), np.zeros((3, 4), np.int32), 2, 2, 8, 6)
    assert e.value.status == lib.ERR_UNSUPPORTED
    with pytest.raises(JxlHipError) as e:
        ctx.palette_delta_wp(np.zeros((4, 4), np.int32), np.zeros((3, 4), np.int32), 2, 2, 8,
                             (32, 10, 7, 7, 7, 0, 0, 13, 12, 12, 12))
    assert e.value.status == lib.ERR_INVALID_ARGUMENT


WP_DEFAULT = (16, 10, 7, 7, 7, 0, 0, 13, 12, 12, 12)   # WeightedHeader defaults, headers/modular.rs:16-66


@pytest.mark.parametrize("hdr", [WP_DEFAULT, (31, 0, 5, 19, 7, 3, 30, 15, 0, 1, 9), (0, 31, 31, 0, 0, 31, 1, 0, 15, 15, 0)])
def test_delta_palette_weighted_predictor_bit_exact(ctx, oracle, hdr):
    """do_palette_step_general, Predictor::Weighted (palette.rs:200-227): the wavefront kernel carries the predictor's
    error state in LDS rings / registers instead of the reference's two summed rows; bit-exact against the oracle
    (pinned on the reference's libjxl golden), incl. images narrower than the reach, one row / one column, more than
    one 256-row band (band-edge state goes through global scratch) and smooth content where the predictor locks on"""
    rng = np.random.default_rng(sum(hdr))
    for (h, w), nb, bit_depth in (((37, 53), 3, 8), ((1, 40), 3, 8), ((50, 1), 1, 8), ((9, 2), 2, 12), ((1100, 7), 3, 8),
                                  ((130, 300), 3, 10), ((530, 70), 1, 8), ((257, 3), 2, 8)):
        num_colors, num_deltas = int(rng.integers(1, 20)), int(rng.integers(0, 8))
        pal = rng.integers(-30, 1 << bit_depth, size=(nb, num_colors + num_deltas)).astype(np.int32)
        pal[:, :num_deltas] = rng.integers(-12, 13, size=(nb, num_deltas))
        idx = rng.integers(-8, num_colors + num_deltas + 100, size=(h, w)).astype(np.int32)
        idx[rng.random((h, w)) < 0.6] = rng.integers(0, max(1, num_deltas + 2))
        got = ctx.palette_delta_wp(idx, pal, num_colors, num_deltas, bit_depth, hdr)
        want = oracle.palette_delta_wp(idx, pal, num_colors, num_deltas, nb, bit_depth, hdr)
        assert np.array_equal(got, want), (f"{h}x{w} nb={nb} colors={num_colors} deltas={num_deltas}",
                                           np.argwhere(got != want)[:4])
    # a smooth ramp coded as "delta 0 everywhere after a seed": the predictor extrapolates, errors stay small
    h, w = 300, 200
    idx = np.zeros((h, w), np.int32)
    idx[0, :] = 1 + (np.arange(w) % 5)
    idx[:, 0] = 1 + (np.arange(h) % 5)
    pal = np.array([[0, 10, 12, 15, 19, 24]], np.int32)
    got = ctx.palette_delta_wp(idx, pal, 5, 1, 8, hdr)
    assert np.array_equal(got, oracle.palette_delta_wp(idx, pal, 5, 1, 1, 8, hdr))


def test_delta_palette_weighted_predictor_device_pointers(ctx, oracle):
    from helpers import DeviceArray
    rng = np.random.default_rng(9)
    h, w, nb = 300, 129, 3
    pal = rng.integers(0, 256, size=(nb, 12)).astype(np.int32)
    pal[:, :4] = rng.integers(-9, 10, size=(nb, 4))
    idx = rng.integers(0, 12, size=(h, w)).astype(np.int32)
    d_idx, d_pal, d_out = DeviceArray(idx), DeviceArray(pal), DeviceArray(nbytes=nb * h * w * 4)
    hdr = np.array(WP_DEFAULT, np.uint32)
    from jxl_rs_amd import lib
    ctx._chk(ctx.L.jxlh_palette_delta_wp(ctx._ctx, lib._addr(d_idx.ptr), w, h, lib._addr(d_pal.ptr), 8, 4, 12, nb, 8,
                                         lib._addr(hdr), lib._addr(d_out.ptr)), "palette_delta_wp")
    ctx.sync()
    got = d_out.download(np.int32, nb * h * w).reshape(nb, h, w)
    assert np.array_equal(got, oracle.palette_delta_wp(idx, pal, 8, 4, nb, 8, WP_DEFAULT))
    for d in (d_idx, d_pal, d_out):
        d.free()


@pytest.mark.parametrize("shape", [(1, 1), (1, 2), (2, 1), (5, 9), (64, 64), (67, 129), (300, 255)])
def test_unsqueeze_bit_exact_and_round_trip(ctx, oracle, shape):
    h, w = shape
    rng = np.random.default_rng(h * 31 + w)
    img = rng.integers(-5000, 5000, size=(h, w)).astype(np.int32)
    a, r = forward_squeeze_h(img)
    got = ctx.unsqueeze(True, a, r, w, h)
    assert np.array_equal(got, oracle.unsqueeze_h(a, r, w))
    assert np.array_equal(got, img)
    a, r = forward_squeeze_v(img)
    got = ctx.unsqueeze(False, a, r, w, h)
    assert np.array_equal(got, oracle.unsqueeze_v(a, r, h))
    assert np.array_equal(got, img)
    # arbitrary (non-encoder) residuals: only parity, no round trip
    avg = rng.integers(0, 256, size=(h, (w + 1) // 2)).astype(np.int32)
    res = np.round(rng.laplace(0, 30, size=(h, w // 2))).astype(np.int32)
    assert np.array_equal(ctx.unsqueeze(True, avg, res, w, h), oracle.unsqueeze_h(avg, res, w))


@pytest.mark.parametrize("shape", [(70, 256), (70, 257), (1, 300), (129, 321), (64, 322), (200, 1031), (3, 2048)])
def test_unsqueeze_long_lines_take_the_streamed_kernel(ctx, oracle, shape):
    """lines of >= 128 steps run in k6_unsqueeze_tiled (mover waves stream 32-step chunks through LDS for one chain
    wave): chunk boundaries, the scalar remainder, odd lengths, line counts that do not fill a workgroup -- in both
    directions (the shape is transposed for the vertical step) and with strided / multi-plane device buffers."""
    lines, n = shape
    rng = np.random.default_rng(lines * 7 + n)
    avg = rng.integers(-3000, 3000, size=(lines, (n + 1) // 2)).astype(np.int32)
    res = np.round(rng.laplace(0, 40, size=(lines, n // 2))).astype(np.int32)
    avg[::4, 1:] = avg[::4, :-1]  # runs of equal averages: zero tendencies
    assert np.array_equal(ctx.unsqueeze(True, avg, res, n, lines), oracle.unsqueeze_h(avg, res, n))
    at, rt = np.ascontiguousarray(avg.T), np.ascontiguousarray(res.T)
    assert np.array_equal(ctx.unsqueeze(False, at, rt, lines, n), oracle.unsqueeze_v(at, rt, n))
    # three planes in one launch, strides wider than the rows, sentinel-checked padding
    from helpers import DeviceArray
    a_stride, r_stride, o_stride = avg.shape[1] + 5, max(res.shape[1], 1) + 3, n + 9
    planes = []
    for c in range(3):
        a = np.zeros((lines, a_stride), np.int32); a[:, :avg.shape[1]] = np.roll(avg, c, axis=0)
        r = np.zeros((lines, r_stride), np.int32); r[:, :res.shape[1]] = np.roll(res, c, axis=0)
        planes.append((a, r, DeviceArray(a), DeviceArray(r), DeviceArray(np.full((lines, o_stride), -77, np.int32))))
    ctx.unsqueeze_planes(True, [p[2].ptr for p in planes], [p[3].ptr for p in planes], [p[4].ptr for p in planes],
                         n, lines, a_stride, r_stride, o_stride)
    ctx.sync()
    for a, r, da, dr, do in planes:
        got = do.download(np.int32, lines * o_stride).reshape(lines, o_stride)
        assert np.array_equal(got[:, :n], oracle.unsqueeze_h(a[:, :avg.shape[1]], r[:, :res.shape[1]], n))
        assert (got[:, n:] == -77).all()
        da.free(); dr.free(); do.free()


@pytest.mark.parametrize("shape", [(1, 1), (1, 2), (3, 7), (4, 70), (16, 64), (21, 64), (22, 65), (32, 200), (50, 129),
                                   (64, 130), (100, 257), (129, 321), (7, 1031), (4096, 70), (4128, 33)])
@pytest.mark.parametrize("op_perm", [(6, 0), (0, 0), (1, 3), (2, 1), (3, 5), (4, 2), (5, 4), (6, 5)])
@pytest.mark.parametrize("horizontal,pad", [(True, (3, 5, 2)), (False, (3, 5, 2)), (False, (4, 8, 0))])
def test_fused_unsqueeze_and_rct(ctx, oracle, shape, op_perm, horizontal, pad):
    """jxlh_unsqueeze_rct == oracle unsqueeze on three planes followed by the oracle RCT: both directions, every op,
    permutations, line counts around the 21-line / 16-column workgroups, lengths around the 32-step chunks (incl.
    everything-in-the-remainder), strided planes with sentinel padding."""
    from helpers import DeviceArray
    lines, n = shape          # lines x n samples along the squeezed axis
    op, perm = op_perm
    rng = np.random.default_rng(lines * 13 + n + op * 7 + perm)
    na, nr = (n + 1) // 2, n // 2
    host, dev = [], []   # pad (4, 8, 0): 16-byte aligned rows -> the vector movers when the column count allows
    for c in range(3):
        a = rng.integers(-3000, 3000, size=(lines, na)).astype(np.int32)
        r = np.round(rng.laplace(0, 40, size=(lines, nr))).astype(np.int32)
        if not horizontal:
            a, r = np.ascontiguousarray(a.T), np.ascontiguousarray(r.T)
        host.append((a, r))
    ow, oh = (n, lines) if horizontal else (lines, n)
    a_stride, r_stride, o_stride = host[0][0].shape[1] + pad[0], max(host[0][1].shape[1], 1) + pad[1], ow + pad[2]
    if pad[2] == 0:  # keep the strides multiples of four wherever the widths are
        a_stride, r_stride = (a_stride + 3) // 4 * 4, (r_stride + 3) // 4 * 4

    def padded(x, stride):
        out = np.zeros((max(x.shape[0], 1), stride), np.int32)
        out[:x.shape[0], :x.shape[1]] = x
        return out
    for a, r in host:
        dev.append((DeviceArray(padded(a, a_stride)), DeviceArray(padded(r, r_stride)),
                    DeviceArray(np.full((oh, o_stride), -55, np.int32))))
    ctx.unsqueeze_rct(horizontal, [d[0].ptr for d in dev], [d[1].ptr for d in dev], [d[2].ptr for d in dev], ow, oh,
                      a_stride, r_stride, o_stride, op, perm)
    ctx.sync()
    unsq = [oracle.unsqueeze_h(a, r, ow) if horizontal else oracle.unsqueeze_v(a, r, oh) for a, r in host]
    want = oracle.rct(unsq, op, perm)
    for c in range(3):
        got = dev[c][2].download(np.int32, oh * o_stride).reshape(oh, o_stride)
        exp = want[c].reshape(oh, ow)
        assert np.array_equal(got[:, :ow], exp), (c, np.argwhere(got[:, :ow] != exp)[:4])
        assert (got[:, ow:] == -55).all()
    for d in dev:
        for x in d:
            x.free()


@pytest.mark.parametrize("horizontal", [True, False])
def test_unsqueeze_rct_separate_passes_on_padded_planes(ctx, oracle, horizontal, monkeypatch):
    """the route planes of 2^31 samples and more take (unsqueeze, then the RCT as ONE strided launch: it used to be one
    launch per row), forced by JXLH_SEPARATE_RCT=1, on padded rows whose sentinel padding must survive"""
    from helpers import DeviceArray
    monkeypatch.setenv("JXLH_SEPARATE_RCT", "1")
    lines, n, op, perm = 37, 101, 6, 1
    rng = np.random.default_rng(99 + horizontal)
    na, nr = (n + 1) // 2, n // 2
    host = []
    for c in range(3):
        a = rng.integers(-3000, 3000, size=(lines, na)).astype(np.int32)
        r = np.round(rng.laplace(0, 40, size=(lines, nr))).astype(np.int32)
        if not horizontal:
            a, r = np.ascontiguousarray(a.T), np.ascontiguousarray(r.T)
        host.append((a, r))
    ow, oh = (n, lines) if horizontal else (lines, n)
    o_stride = ow + 7
    dev = [(DeviceArray(a), DeviceArray(r), DeviceArray(np.full((oh, o_stride), -55, np.int32))) for a, r in host]
    ctx.unsqueeze_rct(horizontal, [d[0].ptr for d in dev], [d[1].ptr for d in dev], [d[2].ptr for d in dev], ow, oh,
                      host[0][0].shape[1], host[0][1].shape[1], o_stride, op, perm)
    ctx.sync()
    unsq = [oracle.unsqueeze_h(a, r, ow) if horizontal else oracle.unsqueeze_v(a, r, oh) for a, r in host]
    want = oracle.rct(unsq, op, perm)
    for c in range(3):
        got = dev[c][2].download(np.int32, oh * o_stride).reshape(oh, o_stride)
        assert np.array_equal(got[:, :ow], want[c].reshape(oh, ow)), c
        assert (got[:, ow:] == -55).all()
    for d in dev:
        for x in d:
            x.free()


def test_unsqueeze_rejects_dimensions_whose_product_wraps(ctx):
    """65536 x 65536: `res_w * res_h` used to be formed in 32 bits, wrapped to 0 and skipped the null / stride /
    device-pointer checks on the residual plane"""
    import ctypes as C
    from jxl_rs_amd import lib
    one = np.zeros((1, 1), dtype=np.int32)
    p = one.ctypes.data_as(C.c_void_p)
    for hz in (0, 1):
        st = ctx.L.jxlh_unsqueeze(ctx._ctx, hz, p, 1 << 20, None, 0, 65536, 65536 * 2, p, 1 << 20)
        assert st in (lib.ERR_UNSUPPORTED, lib.ERR_INVALID_ARGUMENT), st
        pv = (C.c_void_p * 1)(p.value)
        nv = (C.c_void_p * 1)(None)
        st = ctx.L.jxlh_unsqueeze_planes(ctx._ctx, hz, 1, pv, 1 << 20, nv, 0, 65536, 65536 * 2, pv, 1 << 20)
        assert st in (lib.ERR_UNSUPPORTED, lib.ERR_INVALID_ARGUMENT), st
    # inside the dimension bound the residual plane is still checked
    st = ctx.L.jxlh_unsqueeze(ctx._ctx, 1, p, 1 << 19, None, 0, 1 << 20, 1 << 12, p, 1 << 20)
    assert st == lib.ERR_INVALID_ARGUMENT


@pytest.mark.parametrize("scale", [2**15, 2**24, 2**27, 2**28])
def test_unsqueeze_large_magnitudes(ctx, oracle, scale):
    """the device folds the reference's two parity clamps into min() operations (k_modular.hip); checked here
    on large-magnitude data inside the range where the reference's i64 scalar definition and its wrapping
    i32 SIMD form agree (tests/test_oracle_pin.py proves the min() identity on the full i32 range)"""
    rng = np.random.default_rng(scale % 1000)
    h, w = 37, 130
    avg = rng.integers(-scale, scale, size=(h, (w + 1) // 2), dtype=np.int64).astype(np.int32)
    res = rng.integers(-scale, scale, size=(h, w // 2), dtype=np.int64).astype(np.int32)
    avg[2::5, 1:] = avg[2::5, :-1]   # equal neighbours: zero differences
    got = ctx.unsqueeze(True, avg, res, w, h)
    assert np.array_equal(got, oracle.unsqueeze_h(avg, res, w))
    assert np.array_equal(got, oracle.unsqueeze_h(avg, res, w, simd_form=True))  # the reference's wrapping i32 SIMD form
    a2 = rng.integers(-scale, scale, size=((h + 1) // 2, w), dtype=np.int64).astype(np.int32)
    r2 = rng.integers(-scale, scale, size=(h // 2, w), dtype=np.int64).astype(np.int32)
    got = ctx.unsqueeze(False, a2, r2, w, h)
    assert np.array_equal(got, oracle.unsqueeze_v(a2, r2, h))
    assert np.array_equal(got, oracle.unsqueeze_v(a2, r2, h, simd_form=True))


@pytest.mark.parametrize("size", [(700, 500), (257, 129), (9, 300), (1031, 17), (16, 16), (130, 2000)])
@pytest.mark.parametrize("rct", [None, (6, 0), (3, 4)])
def test_unsqueeze_chain_one_call(ctx, oracle, size, rct):
    """jxlh_unsqueeze_chain: the whole default squeeze chain of three channels (+ the RCT after it) in one call"""
    from jxl_rs_amd.modular import ModularChain
    w, h = size
    ch = ModularChain(ctx, w, h, seed=w + 3 * h, rct=rct)
    try:
        ch.run_chain()
        got = ch.result()
        want = helpers.modular_chain_oracle(ch, oracle)
        for c in range(3):
            assert np.array_equal(got[c], want[c]), (size, rct, c, np.argwhere(got[c] != want[c])[:5])
    finally:
        ch.free()


def test_unsqueeze_chain_single_plane_strided_and_errors(ctx, oracle):
    from jxl_rs_amd import lib, synth
    from helpers import DeviceArray
    w, h = 333, 210
    base, residuals, steps = synth.make_modular_planes(w, h, seed=9, nchan=1)
    want = base[0]
    for (hz, ow, oh), res in zip(steps, residuals):
        want = oracle.unsqueeze_h(want, res[0], ow) if hz else oracle.unsqueeze_v(want, res[0], oh)
    d_base = DeviceArray(base[0])
    d_res = [DeviceArray(r[0]) if r[0].size else None for r in residuals]
    stride = w + 5
    d_out = DeviceArray(np.full((h, stride), -9, np.int32))
    levels = [(hz, ow, oh, [d.ptr if d else None], max(r[0].shape[1], 1))
              for (hz, ow, oh), r, d in zip(steps, residuals, d_res)]
    ctx.unsqueeze_chain(levels, [d_base.ptr], base[0].shape[1], base[0].shape[1], base[0].shape[0], [d_out.ptr], stride)
    ctx.sync()
    got = d_out.download(np.int32, h * stride).reshape(h, stride)
    assert np.array_equal(got[:, :w], want) and (got[:, w:] == -9).all()
    # an RCT needs three planes; level geometry must chain
    with pytest.raises(lib.JxlHipError) as e:
        ctx.unsqueeze_chain(levels, [d_base.ptr], base[0].shape[1], base[0].shape[1], base[0].shape[0], [d_out.ptr], stride,
                            rct=(6, 0))
    assert e.value.status == lib.ERR_INVALID_ARGUMENT
    bad = list(levels)
    bad[1] = (bad[1][0], bad[1][1] + 2, bad[1][2], bad[1][3], bad[1][4])
    with pytest.raises(lib.JxlHipError):
        ctx.unsqueeze_chain(bad, [d_base.ptr], base[0].shape[1], base[0].shape[1], base[0].shape[0], [d_out.ptr], stride)
    for d in [d_base, d_out] + [d for d in d_res if d]:
        d.free()


def test_modular_chain_config4_style(ctx, oracle):
    """Default squeeze chain + YCoCg RCT + palette on a mid-size image, bit-exact end to end."""
    from jxl_rs_amd import synth
    w, h = 700, 500
    base, residuals, steps = synth.make_modular_planes(w, h, seed=3)
    cur_g = [b.copy() for b in base]
    cur_o = [b.copy() for b in base]
    for (horizontal, ow, oh), res in zip(steps, residuals):
        for c in range(3):
            cur_g[c] = ctx.unsqueeze(horizontal, cur_g[c], res[c], ow, oh)
            cur_o[c] = oracle.unsqueeze_h(cur_o[c], res[c], ow) if horizontal else oracle.unsqueeze_v(cur_o[c], res[c], oh)
            assert np.array_equal(cur_g[c], cur_o[c]), (horizontal, ow, oh, c)
    assert cur_g[0].shape == (h, w)
    got = ctx.rct(cur_g, 6, 0)
    want = oracle.rct(cur_o, 6, 0)
    for c in range(3):
        assert np.array_equal(got[c], want[c])


@pytest.mark.gpu
def test_epf_fast_reciprocal_is_ieee_exact_on_weight_range():
    """1/(1 + sum w) in the EPF kernels uses rcp + FMA refinement; it must equal IEEE division for
    EVERY float the weight sum can take: [1, 5] for EPF1/2, [1, 13] for EPF0 -> checked on [1, 16)."""
    from jxl_rs_amd import Context
    c = Context(0, 1)
    try:
        assert c.selftest_recip(1.0, 16.0) == 0
    finally:
        c.close()


# ---------------------------------------------------------------- sparse coefficient transport
def _sparse_frame_equals_dense(ctx, wl, mutate=None, frame_flags=0):
    """Runs the frame twice -- dense submit vs sparse submit -- and checks identical planes."""
    from jxl_rs_amd import synth
    coeffs = wl.coeffs.copy()
    if mutate is not None:
        mutate(coeffs)
    outs = []
    for sparse in (False, True):
        params = synth.apply_opts(ctx.default_params(wl.xsize, wl.ysize), wl)
        params.flags = frame_flags
        ctx.frame_begin(params)
        ctx.set_dequant_tables(wl.tables)
        ctx.set_lf_quantized(*wl.lf_q)
        ctx.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
        if sparse:
            # half the groups one by one, the rest as one batch on the other slot
            ng = coeffs.shape[0]
            for g in range(ng // 2):
                pairs, n, wide = synth.to_sparse(coeffs[g])
                ctx.submit_group_sparse(g, pairs, n, wide, slot=0)
                ctx.slot_wait(0)
            ids, runs, ns, wides = [], [], [], []
            for g in range(ng // 2, ng):
                pairs, n, wide = synth.to_sparse(coeffs[g])
                ids.append(g); runs.append(pairs); ns.append(n)
                if len(wide):
                    wide = wide.copy()
                    wide[:, 0] += np.uint32(g * 3 * 65536)
                    wides.append(wide)
            if ids:
                ctx.submit_groups_sparse(np.array(ids, np.uint32), np.concatenate(runs), np.concatenate(ns),
                                         np.concatenate(wides) if wides else None, slot=1)
                ctx.slot_wait(1)
        else:
            for g in range(coeffs.shape[0]):
                ctx.submit_group(g, coeffs[g], slot=g % 2)
            ctx.slot_wait(0); ctx.slot_wait(1)
        ctx.frame_run()
        ctx.sync()
        outs.append(ctx.read_planes())
    for a, b in zip(outs[0], outs[1]):
        assert bit_equal(a, b), diff_report(a, b)


@pytest.mark.parametrize("frame_flags", [0, 2], ids=["k1-reads-pairs", "expand-to-dense"])
@pytest.mark.parametrize("mix,size", [("MIX_ALL", (520, 300)), ("MIX_D1", (1024, 768)), ("MIX_DCT8", (256, 256))])
def test_sparse_submit_matches_dense_submit(ctx, mix, size, frame_flags):
    """every group submitted as pairs: the transforms read the bucketed pairs directly (groups with
    special / large varblocks are expanded on the side); JXLH_FRAME_EXPAND_SPARSE forces the dense slabs"""
    from jxl_rs_amd import synth
    wl = synth.make_vardct(size[0], size[1], mix=getattr(synth, mix), seed=11, epf_iters=2)
    _sparse_frame_equals_dense(ctx, wl, frame_flags=frame_flags)


def test_sparse_k1_rerun_and_resubmit(ctx, oracle):
    """a frame submitted as pairs can be run again (bands) without resubmitting, and resubmitting one
    group densely afterwards falls back to the slabs -- both still equal the oracle"""
    from jxl_rs_amd import synth
    wl = synth.make_vardct(520, 700, mix=synth.MIX_D1, seed=41, epf_iters=2)
    want, _ = run_oracle_frame(oracle, wl)
    params = synth.apply_opts(ctx.default_params(wl.xsize, wl.ysize), wl)
    ctx.frame_begin(params)
    ctx.set_dequant_tables(wl.tables)
    ctx.set_lf_quantized(*wl.lf_q)
    ctx.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
    for g in range(wl.coeffs.shape[0]):
        pairs, n, wide = synth.to_sparse(wl.coeffs[g])
        ctx.submit_group_sparse(g, pairs, n, wide)
    ctx.slot_wait(0)
    for row0, row1 in ((0, 3), (1, 2), (0, 3)):
        ctx.frame_run(row0, row1)
        ctx.sync()
        got = ctx.read_planes()
        y0, y1 = row0 * 256, min(row1 * 256, wl.ysize)
        for c in range(3):
            assert bit_equal(got[c][y0:y1], want[c][y0:y1]), (row0, row1, c)
    ctx.submit_group(2, wl.coeffs[2])   # same content, dense: the frame is no longer pairs-only
    ctx.slot_wait(0)
    ctx.frame_run()
    ctx.sync()
    got = ctx.read_planes()
    for c in range(3):
        assert bit_equal(got[c], want[c]), diff_report(got[c], want[c])


def test_sparse_submit_wide_values_and_empty_groups(ctx):
    """values outside i16 travel in the wide list; an all-zero group is a valid (empty) submission"""
    from jxl_rs_amd import synth
    wl = synth.make_vardct(512, 512, mix=synth.MIX_D1, seed=12, epf_iters=1)

    def mutate(c):
        c[0, 1, 5] = 40000
        c[0, 0, 77] = -70000
        c[1, 2, 65535] = 32768
        c[1, 1, 0] = -32769
        c[2, :, :] = 0
    _sparse_frame_equals_dense(ctx, wl, mutate)


def test_sparse_duplicates_accumulate(ctx):
    """duplicate positions add up with wrapping i32 `+=` (multi-pass accumulation, group.rs:572)"""
    from jxl_rs_amd import synth
    wl = synth.make_vardct(256, 256, mix=synth.MIX_DCT8, seed=13, epf_iters=0, gab=False)
    params = synth.apply_opts(ctx.default_params(256, 256), wl)
    outs = []
    for split in (False, True):
        ctx.frame_begin(params)
        ctx.set_dequant_tables(wl.tables)
        ctx.set_lf_quantized(*wl.lf_q)
        ctx.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
        if not split:
            pairs, n, wide = synth.to_sparse(wl.coeffs[0])
            ctx.submit_group_sparse(0, pairs, n, wide)
        else:
            # every coefficient v sent as (v - 3) and (+3): two passes over the same positions
            a = wl.coeffs[0].copy()
            nz = a != 0
            first = np.where(nz, a - 3, 0)
            second = np.where(nz, 3, 0)
            p1, n1, _ = synth.to_sparse(first)
            p2, n2, _ = synth.to_sparse(second)
            # positions whose first part became zero vanish from p1; that is fine (0 + 3)
            runs, ns = [], []
            o1 = np.concatenate([[0], np.cumsum(n1.astype(np.int64))]); o2 = np.concatenate([[0], np.cumsum(n2.astype(np.int64))])
            for c in range(3):
                runs += [p1[o1[c]:o1[c + 1]], p2[o2[c]:o2[c + 1]]]
                ns.append(int(n1[c] + n2[c]))
            ctx.submit_group_sparse(0, np.concatenate(runs), np.array(ns, np.uint32), None)
        ctx.slot_wait(0)
        ctx.frame_run()
        ctx.sync()
        outs.append(ctx.read_planes())
    for a, b in zip(outs[0], outs[1]):
        assert bit_equal(a, b), diff_report(a, b)


@pytest.mark.parametrize("mix,size,scale", [("MIX_D1", (520, 300), 1), ("MIX_ALL", (300, 270), 40)])
def test_sparse8_submit_matches_dense_submit(ctx, oracle, mix, size, scale):
    """jxlh_submit_groups_sparse8 (u16 positions + i8 values, larger values through the wide list) gives the frame
    the dense submission gives, bit for bit; scale 40 pushes a good share of the values past 8 bits"""
    from jxl_rs_amd import synth
    w, h = size
    wl = synth.make_vardct(w, h, mix=getattr(synth, mix), seed=w + h, epf_iters=2, coeff_scale=scale)
    want, _ = run_gpu_frame(ctx, wl)
    p = gpu_params_from(ctx, wl)
    ctx.frame_begin(p)
    ctx.set_dequant_tables(wl.tables)
    ctx.set_lf_quantized(*wl.lf_q)
    ctx.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
    ng = wl.coeffs.shape[0]
    parts = [synth.to_sparse8(wl.coeffs[g]) for g in range(ng)]
    pos = np.concatenate([q[0] for q in parts])
    val = np.concatenate([q[1] for q in parts])
    n = np.concatenate([q[2] for q in parts])
    wide = []
    for g, q in enumerate(parts):   # the batched form addresses wide entries frame-wide
        if len(q[3]):
            e = q[3].copy()
            e[:, 0] += np.uint32(g * 3 * 65536)
            wide.append(e)
    wide = np.concatenate(wide) if wide else None
    if scale > 1:
        assert wide is not None and len(wide) > 100
    ctx.submit_groups_sparse8(np.arange(ng, dtype=np.uint32), pos, val, n, wide)
    ctx.slot_wait(0)
    ctx.frame_run()
    ctx.sync()
    got = ctx.read_planes()
    for c in range(3):
        assert bit_equal(got[c], want[c]), f"plane {c}: {diff_report(got[c], want[c])}"


@pytest.mark.parametrize("mix,size,scale", [("MIX_D1", (520, 300), 1), ("MIX_D1", (1024, 512), 3), ("MIX_ALL", (300, 270), 40)])
def test_sparse4_submit_matches_dense_submit(ctx, oracle, mix, size, scale):
    """jxlh_submit_groups_sparse4 (2 bytes per update: 12-bit position inside a 4096-coefficient segment + value nibble,
    per-segment counts; larger values through the 3-byte overflow arrays, the largest through the wide list) gives the
    frame the dense submission gives, bit for bit; the scales push shares of the values into the overflow forms"""
    from jxl_rs_amd import synth
    w, h = size
    wl = synth.make_vardct(w, h, mix=getattr(synth, mix), seed=w + h, epf_iters=2, coeff_scale=scale)
    want, _ = run_gpu_frame(ctx, wl)
    p = gpu_params_from(ctx, wl)
    ctx.frame_begin(p)
    ctx.set_dequant_tables(wl.tables)
    ctx.set_lf_quantized(*wl.lf_q)
    ctx.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
    ng = wl.coeffs.shape[0]
    parts = [synth.to_sparse4(wl.coeffs[g]) for g in range(ng)]
    wide = []
    for g, q in enumerate(parts):   # the batched form addresses wide entries frame-wide
        if len(q[5]):
            e = q[5].copy()
            e[:, 0] += np.uint32(g * 3 * 65536)
            wide.append(e)
    wide = np.concatenate(wide) if wide else None
    n8 = np.concatenate([q[4] for q in parts])
    if scale > 1:
        assert n8.sum() > 100
    if scale >= 40:
        assert wide is not None and len(wide) > 100
    # two batches on two slots, like two decoder threads
    half = ng // 2 if ng > 1 else ng
    for sl, (g0, g1) in enumerate(((0, half), (half, ng))):
        if g0 >= g1:
            continue
        sel = parts[g0:g1]
        wsel = None
        if wide is not None:
            m = (wide[:, 0] // (3 * 65536) >= g0) & (wide[:, 0] // (3 * 65536) < g1)
            wsel = wide[m] if m.any() else None
        ctx.submit_groups_sparse4(np.arange(g0, g1, dtype=np.uint32), np.concatenate([q[0] for q in sel]),
                                  np.concatenate([q[1].reshape(-1) for q in sel]), np.concatenate([q[2] for q in sel]),
                                  np.concatenate([q[3] for q in sel]), np.concatenate([q[4] for q in sel]), wsel, slot=0)
    ctx.slot_wait(0)
    ctx.frame_run()
    ctx.sync()
    got = ctx.read_planes()
    for c in range(3):
        assert bit_equal(got[c], want[c]), f"plane {c}: {diff_report(got[c], want[c])}"


@pytest.mark.parametrize("bits12", [False, True])
@pytest.mark.parametrize("mix,size,scale,partial", [("MIX_D1", (520, 300), 1, False), ("MIX_D1", (1024, 512), 30, False),
                                                    ("MIX_ALL", (300, 270), 200, False), ("MIX_D1", (768, 512), 1, True)])
def test_slot_bucketed_submit_matches_dense_submit(ctx, oracle, mix, size, scale, partial, bits12):
    """jxlh_submit_groups_slots (2 bytes per update, bucketed by 64-coefficient slot on the host: the frame is not sorted
    on the device) gives the frame the dense submission gives, bit for bit -- with a whole frame in this form (the
    no-sort route), with values past 10 bits in the wide list (those groups take the dense route), and with only SOME
    groups in this form and the others as plain pairs (per-group routing, tests/test_gpu_routing.py)"""
    from jxl_rs_amd import synth
    w, h = size
    wl = synth.make_vardct(w, h, mix=getattr(synth, mix), seed=w + h, epf_iters=2, coeff_scale=scale)
    want, _ = run_gpu_frame(ctx, wl)
    p = gpu_params_from(ctx, wl)
    ctx.frame_begin(p)
    ctx.set_dequant_tables(wl.tables)
    ctx.set_lf_quantized(*wl.lf_q)
    ctx.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
    ng = wl.coeffs.shape[0]
    slotted = [g for g in range(ng) if not (partial and g % 3 == 1)]
    from jxl_rs_amd import lib as jl
    parts = {g: synth.to_slots(wl.coeffs[g], bits12) for g in slotted}
    wide = []
    for g, q in parts.items():   # the batched form addresses wide entries frame-wide
        if len(q[3]):
            e = q[3].copy()
            e[:, 0] += np.uint32(g * 3 * 65536)
            wide.append(e)
    wide = np.concatenate(wide) if wide else None
    if scale >= 200:
        assert wide is not None and len(wide) > 50
    ctx.kernel_timing_reset()
    ctx.kernel_timing(True)
    ctx.submit_groups_slots(np.asarray(slotted, dtype=np.uint32), np.concatenate([parts[g][0] for g in slotted]),
                            np.concatenate([parts[g][1].reshape(-1) for g in slotted]),
                            np.concatenate([parts[g][2] for g in slotted]), wide,
                            flags=jl.GROUP_COMPLETE | (jl.GROUP_ENTRIES12 if bits12 else 0))
    for g in range(ng):
        if g not in parts:
            pr, n3, wd = synth.to_sparse(wl.coeffs[g])
            ctx.submit_group_sparse(g, pr, n3, wd)
    ctx.slot_wait(0)
    ctx.frame_run()
    ctx.sync()
    kt = ctx.kernel_times()
    ctx.kernel_timing(False)
    if not partial and wide is None:
        assert "k_sort_sparse" not in kt and "copy_bucketed_pairs" not in kt, sorted(kt)  # neither a sort nor a copy
    if partial:   # round 6: the plain-pairs groups are routed to their dense slabs, the others are still read in place
        assert "k_sort_sparse" not in kt and "k_expand_sparse" in kt, sorted(kt)
    got = ctx.read_planes()
    for c in range(3):
        assert bit_equal(got[c], want[c]), f"plane {c}: {diff_report(got[c], want[c])}"
    # the frame once more without resubmitting anything: the bucketed store persists across runs
    ctx.frame_run()
    ctx.sync()
    again = ctx.read_planes()
    for c in range(3):
        assert bit_equal(again[c], want[c])


def _merge_slot_forms(a, b):
    """two synth.to_slots results of one group -> one list per (channel, slot): the updates of two passes in ONE
    submission (duplicate positions add up on the device like `coeffs[i] += v`, frame/group.rs:572)"""
    ea, ca, na, wa = a
    eb, cb, nb_, wb = b
    assert len(wa) == 0 and len(wb) == 0
    oa = np.concatenate([[0], np.cumsum(ca.reshape(-1).astype(np.int64))])
    ob = np.concatenate([[0], np.cumsum(cb.reshape(-1).astype(np.int64))])
    out = []
    for i in range(3 * 1024):
        out.append(ea[oa[i]:oa[i + 1]])
        out.append(eb[ob[i]:ob[i + 1]])
    cnt = ca.astype(np.int64) + cb.astype(np.int64)
    assert cnt.max() <= 255
    return np.concatenate(out), cnt.astype(np.uint8), (na + nb_).astype(np.uint32)


@pytest.mark.parametrize("mix,size,dense_groups", [("MIX_D1", (768, 520), ()), ("MIX_D1", (520, 300), (1, 2)),
                                                   ("MIX_ALL", (600, 520), (0, 4)), ("MIX_DCT8", (300, 260), (0,))])
def test_entries_form_direct_path_and_its_fallbacks(ctx, oracle, mix, size, dense_groups):
    """A frame resident in the slot-bucketed form: the transforms dequantise only the positions that have an entry
    (direct path), leave varblocks with more entries than their lanes hold to the dense pass (fallback list: a few
    groups are made dense here), and JXLH_FRAME_DENSE_DEQUANT takes the dense pass everywhere -- all three give the
    bits of the dense-slab submission; so do duplicate positions (two passes' updates in one list, some cancelling to
    zero), which must add up as integers before they are dequantised."""
    from jxl_rs_amd import synth
    from jxl_rs_amd import lib as jl
    w, h = size
    wl = synth.make_vardct(w, h, mix=getattr(synth, mix), seed=w + 7 * h, epf_iters=1)
    rng = np.random.default_rng(w)
    for g in dense_groups:   # every second coefficient non-zero: far more entries than D per lane
        m = rng.random(wl.coeffs[g].shape) < 0.5
        wl.coeffs[g] = np.where(m, rng.integers(-9, 10, size=wl.coeffs[g].shape), wl.coeffs[g]).astype(np.int32)
    want, _ = run_gpu_frame(ctx, wl)
    ng = wl.coeffs.shape[0]
    # duplicates: c = a + b with overlapping supports; where b == -a' the two updates cancel
    split = rng.random(wl.coeffs.shape) < 0.3
    part_b = np.where(split & (wl.coeffs != 0), rng.integers(-3, 4, size=wl.coeffs.shape), 0).astype(np.int32)
    part_a = (wl.coeffs - part_b).astype(np.int32)
    zero_sum = (wl.coeffs == 0) & (rng.random(wl.coeffs.shape) < 0.01)      # +v and -v at a position that holds 0
    part_a = np.where(zero_sum, 5, part_a).astype(np.int32)
    part_b = np.where(zero_sum, -5, part_b).astype(np.int32)
    assert np.array_equal(part_a + part_b, wl.coeffs)
    ids = np.arange(ng, dtype=np.uint32)
    for what, flags in (("direct", 0), ("dense pass", jl.FRAME_DENSE_DEQUANT), ("duplicates", 0)):
        p = gpu_params_from(ctx, wl)
        p.flags = flags
        ctx.frame_begin(p)
        ctx.set_dequant_tables(wl.tables)
        ctx.set_lf_quantized(*wl.lf_q)
        ctx.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
        if what == "duplicates":
            parts = [_merge_slot_forms(synth.to_slots(part_a[g]), synth.to_slots(part_b[g])) for g in range(ng)]
        else:
            parts = [synth.to_slots(wl.coeffs[g]) for g in range(ng)]
            assert all(len(q[3]) == 0 for q in parts)
        ctx.submit_groups_slots(ids, np.concatenate([q[0] for q in parts]), np.concatenate([q[1].reshape(-1) for q in parts]),
                                np.concatenate([q[2] for q in parts]), None)
        ctx.slot_wait(0)
        for run in range(2):   # the resident form is read again by the second run
            ctx.frame_run()
            ctx.sync()
            got = ctx.read_planes()
            for c in range(3):
                assert bit_equal(got[c], want[c]), f"{what}, run {run}, plane {c}: {diff_report(got[c], want[c])}"


@pytest.mark.parametrize("dense_dequant", [False, True])
def test_slot_counts_that_disagree_with_the_run_length(ctx, dense_dequant):
    """a caller's mistake must stay inside its own run: a slot-count table that claims MORE entries than n says is cut at the
    run's end, entries the table does not account for are ignored -- in the direct path, in the dense pass, and when the
    group is widened into pair words (mixed epoch).  Expected result: the frame of the entries the table does cover."""
    from jxl_rs_amd import synth
    from jxl_rs_amd import lib as jl
    wl = synth.make_vardct(512, 256, mix=synth.MIX_D1, seed=77, epf_iters=0, gab=False)   # two groups
    ents, cnts, ns = [], [], []
    truth = wl.coeffs.copy()
    for g in range(2):
        e, c, n, wide = synth.to_slots(wl.coeffs[g])
        assert len(wide) == 0
        ents.append(e); cnts.append(c.copy()); ns.append(n.copy())
    # group 0, channel 1 (Y): n is 10 short of what the table says -> the last 10 entries of the run are not there
    cut = 10
    off_y = int(ns[0][0])
    e0 = np.concatenate([ents[0][:off_y + int(ns[0][1]) - cut], ents[0][off_y + int(ns[0][1]):]])
    n0 = ns[0].copy(); n0[1] -= cut
    # ... which are the last entries in slot order: zero them in the expected coefficients
    pos = np.flatnonzero(truth[0, 1])
    truth[0, 1, pos[-cut:]] = 0
    # group 1, channel 0 (X): the table covers 7 entries fewer than n (its last non-empty slots are short): they are ignored
    c1 = cnts[1].copy()
    short, s_ = 7, 1023
    posx = np.flatnonzero(truth[1, 0])
    dropped = 0
    while dropped < short:
        while c1[0, s_] == 0:
            s_ -= 1
        c1[0, s_] -= 1
        dropped += 1
    # entries of a slot are in position order (np.flatnonzero): a shorter count keeps the slot's FIRST entries; every later
    # slot then reads shifted entries -- so shorten only from the tail: the dropped ones are the run's last entries per slot.
    # Simplest exact expectation: rebuild the coefficients the device will see from (entries, counts) in numpy.
    def rebuild(e, c, n):
        out = np.zeros((3, 65536), np.int32)
        o = 0
        for ch in range(3):
            run = e[o:o + int(n[ch])]
            o += int(n[ch])
            k = 0
            for s in range(1024):
                take = run[k:k + int(c[ch, s])]      # cut at the run's end by the slice
                k += int(c[ch, s])
                v = (take.astype(np.int32) << 16 >> 22)
                np.add.at(out[ch], s * 64 + (take & 63).astype(np.int64), v)
        return out
    want_coeffs = np.stack([rebuild(e0, cnts[0], n0), rebuild(ents[1], c1, ns[1])])
    assert not np.array_equal(want_coeffs, wl.coeffs)
    wl2 = wl
    saved = wl.coeffs
    wl2.coeffs = want_coeffs
    want, _ = run_gpu_frame(ctx, wl2)
    wl.coeffs = saved
    for mixed in (False, True):
        p = gpu_params_from(ctx, wl)
        p.flags = jl.FRAME_DENSE_DEQUANT if dense_dequant else 0
        ctx.frame_begin(p)
        ctx.set_dequant_tables(wl.tables)
        ctx.set_lf_quantized(*wl.lf_q)
        ctx.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
        if mixed:   # group 1 as plain pairs of the same (already shortened) coefficients: group 0 is widened into pair words
            ctx.submit_groups_slots(np.uint32([0]), e0, cnts[0].reshape(-1), n0, None)
            ctx.submit_group_sparse(1, *synth.to_sparse(want_coeffs[1]))
        else:
            ctx.submit_groups_slots(np.uint32([0, 1]), np.concatenate([e0, ents[1]]),
                                    np.concatenate([cnts[0].reshape(-1), c1.reshape(-1)]), np.concatenate([n0, ns[1]]), None)
        ctx.slot_wait(0)
        ctx.frame_run()
        ctx.sync()
        got = ctx.read_planes()
        for c in range(3):
            assert bit_equal(got[c], want[c]), f"mixed={mixed}, plane {c}: {diff_report(got[c], want[c])}"


def test_sparse_submit_argument_errors(ctx):
    from jxl_rs_amd import synth
    from jxl_rs_amd.lib import JxlHipError
    wl = synth.make_vardct(256, 256, mix=synth.MIX_DCT8, seed=14, epf_iters=0, gab=False)
    ctx.frame_begin(synth.apply_opts(ctx.default_params(256, 256), wl))
    pairs, n, wide = synth.to_sparse(wl.coeffs[0])
    with pytest.raises(JxlHipError):
        ctx.submit_group_sparse(5, pairs, n, wide)          # group out of range
    with pytest.raises(JxlHipError):
        ctx.submit_group_sparse(0, pairs, n, np.array([[3 * 65536, 1]], np.uint32))  # wide pos out of range
    ctx.submit_group_sparse(0, pairs, n, wide, flags=0)      # a progressive pass (tests/test_gpu_progressive.py)
    with pytest.raises(JxlHipError):
        ctx.submit_group_sparse(0, pairs, n, wide)           # twice in one epoch
    ctx.slot_wait(0)


def test_sparse_expansion_matches_oracle_slab(ctx, oracle):
    """reads the device coefficient store back after the zero-fill + scatter kernel and compares it
    with the oracle's expansion (wide values, duplicates, an untouched group stays as submitted)"""
    import ctypes as C
    from jxl_rs_amd import synth
    wl = synth.make_vardct(512, 256, mix=synth.MIX_D1, seed=15, epf_iters=0, gab=False)
    ctx.frame_begin(synth.apply_opts(ctx.default_params(512, 256), wl))
    ctx.set_dequant_tables(wl.tables)
    ctx.set_lf_quantized(*wl.lf_q)
    ctx.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
    slab = wl.coeffs[1].copy()
    slab[0, 3] = 123456
    slab[2, 65535] = -40000
    pairs, n, wide = synth.to_sparse(slab)
    dup = np.array([7 | (5 << 16), 7 | (0xFFFE << 16)], dtype=np.uint32)  # pos 7: +5, -2 in channel X
    pairs = np.concatenate([dup, pairs]); n = n.copy(); n[0] += 2
    ctx.submit_group(0, wl.coeffs[0])                 # dense
    ctx.submit_group_sparse(1, pairs, n, wide)        # sparse
    ctx.slot_wait(0)
    ctx.frame_run()
    ctx.sync()
    ptr, count = ctx.coeff_buffer()
    got = np.zeros(count, dtype=np.int32)
    hip = C.CDLL("libamdhip64.so")
    assert hip.hipMemcpy(C.c_void_p(got.ctypes.data), C.c_void_p(ptr), C.c_size_t(got.nbytes), C.c_int(2)) == 0
    got = got.reshape(2, 3, 65536)
    assert np.array_equal(got[0], wl.coeffs[0])
    assert np.array_equal(got[1], oracle.expand_sparse(pairs, n, wide))
    assert got[1][0, 7] == slab[0, 7] + 3


# ---------------------------------------------------------------- 8-bit sRGB output stage
def _default_xyb_params(oracle, kat, intensity_target=255.0):
    k = kat["output_stage"]
    return oracle.xyb_params(k["opsin_inverse_matrix"], [k["opsin_bias"]] * 3, intensity_target)


@pytest.mark.parametrize("size,channels,intensity", [((520, 300), 3, 255.0), ((333, 77), 4, 255.0),
                                                     ((66, 34), 3, 400.0), ((9, 9), 4, 255.0)])
def test_rgb8_output_bit_exact(ctx, oracle, kat, size, channels, intensity):
    """XybStage + sRGB FromLinearStage + ConvertF32ToU8Stage in one device pass == the oracle's
    restatement of the three stages applied to the oracle's planes, byte for byte"""
    from jxl_rs_amd import synth
    w, h = size
    wl = synth.make_vardct(w, h, mix=synth.MIX_D1, seed=w * 7 + h, epf_iters=2)
    want_planes, _ = run_oracle_frame(oracle, wl)
    params = _default_xyb_params(oracle, kat, intensity)
    want = oracle.xyb_to_rgb8(params, want_planes, w, h, channels)
    upload_frame(ctx, wl)
    ctx.frame_run()
    ctx.sync()
    got = ctx.read_rgb8(params, channels)
    assert got.shape == want.shape
    bad = np.argwhere(got != want)
    assert bad.size == 0, f"{len(bad)} differing bytes, first at {bad[0]}: got {got[tuple(bad[0])]} want {want[tuple(bad[0])]}"
    # a band of rows equals the same rows of the whole image (dither is position dependent)
    y0, y1 = h // 3, max(h // 3 + 1, (2 * h) // 3)
    band = ctx.read_rgb8(params, channels, y0, y1)
    assert np.array_equal(band, want[y0:y1])
    assert len(np.unique(want)) > 16, "test image should exercise many code values"
    # 16-bit samples (ConvertF32ToU16Stage: no dither)
    want16 = oracle.xyb_to_rgb16(params, want_planes, w, h, channels)
    assert np.array_equal(ctx.read_rgb16(params, channels), want16)
    assert np.array_equal(ctx.read_rgb16(params, channels, y0, y1), want16[y0:y1])


@pytest.mark.parametrize("tf,param", [("linear", 0.0), ("srgb", 0.0), ("bt709", 0.0), ("pq", 10000.0), ("pq", 4000.0),
                                      ("hlg", -0.1667), ("hlg", 0.05), ("gamma", 0.45454545)])
@pytest.mark.parametrize("bits,channels", [(8, 3), (16, 4)])
def test_output_transfer_functions_bit_exact(ctx, oracle, kat, tf, param, bits, channels):
    """XybStage + FromLinearStage(tf) + integer conversion through jxlh_frame_read_output, every transfer function
    of render/stages/from_linear.rs:133-145, byte for byte vs the oracle"""
    from jxl_rs_amd import synth, lib
    w, h = 300, 140
    wl = synth.make_vardct(w, h, mix=synth.MIX_D1, seed=11, epf_iters=1)
    want_planes, _ = run_oracle_frame(oracle, wl)
    params = _default_xyb_params(oracle, kat, 255.0 if tf != "pq" else param)
    lum = (0.2627, 0.678, 0.0593)
    want = oracle.xyb_to_rgb_tf(params, tf, want_planes, w, h, channels, bits, param, lum)
    upload_frame(ctx, wl)
    ctx.frame_run()
    ctx.sync()
    got = ctx.read_output(lib.COLOR_XYB, tf, params, param, lum, bits, channels)
    bad = np.argwhere(got != want)
    assert bad.size == 0, f"{len(bad)} differing samples, first at {bad[0]}: got {got[tuple(bad[0])]} want {want[tuple(bad[0])]}"
    assert len(np.unique(want)) > 16
    if tf == "srgb":  # the dedicated entry points are this call
        ref = ctx.read_rgb8(params, channels) if bits == 8 else ctx.read_rgb16(params, channels)
        assert np.array_equal(got, ref)
    # COLOR_NONE: the planes are taken as RGB
    raw = ctx.read_output(lib.COLOR_NONE, "linear", None, 0.0, lum, 16, 3)
    want_raw = np.stack([np.rint(np.clip(p, 0, 1) * np.float32(65535)).astype(np.uint16) for p in want_planes], axis=-1)
    assert np.array_equal(raw, want_raw)


def test_rgb8_async_read_equals_blocking_read(ctx, oracle, kat):
    """jxlh_frame_read_rgb8_async: same bytes as the blocking call once the context has been synchronised,
    into pinned host memory"""
    import ctypes as C
    from jxl_rs_amd import synth
    w, h = 333, 77
    wl = synth.make_vardct(w, h, mix=synth.MIX_D1, seed=5, epf_iters=2)
    upload_frame(ctx, wl)
    ctx.frame_run()
    ctx.sync()
    params = _default_xyb_params(oracle, kat)
    want = ctx.read_rgb8(params, 3)
    pr = np.ascontiguousarray(params, dtype=np.float32)
    pinned, addr = ctx.alloc_pinned(w * h * 3)
    pinned[:] = 0
    ctx._chk(ctx.L.jxlh_frame_read_rgb8_async(ctx._ctx, pr.ctypes.data_as(C.c_void_p), 3, 0, h, C.c_void_p(addr), w * 3),
             "frame_read_rgb8_async")
    ctx.sync()
    assert np.array_equal(pinned[:w * h * 3].reshape(h, w, 3), want)
    # the general descriptor form, 16-bit PQ
    from jxl_rs_amd import lib
    want16 = ctx.read_output(lib.COLOR_XYB, "pq", params, 4000.0, bits=16, channels=3)
    d = lib.OutputDesc()
    d.color, d.transfer, d.bits, d.channels, d.tf_param = lib.COLOR_XYB, lib.TF["pq"], 16, 3, 4000.0
    for i, v in enumerate(np.asarray(params, dtype=np.float32).ravel()):
        d.xyb[i] = float(v)
    pinned16, addr16 = ctx.alloc_pinned(w * h * 3 * 2)
    ctx._chk(ctx.L.jxlh_frame_read_output_async(ctx._ctx, C.byref(d), 0, h, C.c_void_p(addr16), w * 3 * 2),
             "frame_read_output_async")
    ctx.sync()
    assert np.array_equal(pinned16[:w * h * 6].view(np.uint16).reshape(h, w, 3), want16)


def test_rgb8_output_argument_errors(ctx, oracle, kat):
    from jxl_rs_amd import synth
    from jxl_rs_amd.lib import JxlHipError
    wl = synth.make_vardct(64, 64, mix=synth.MIX_DCT8, seed=3, epf_iters=0, gab=False)
    upload_frame(ctx, wl)
    params = _default_xyb_params(oracle, kat)
    with pytest.raises(JxlHipError):
        ctx.read_rgb8(params, 3)          # no frame has been run yet
    ctx.frame_run()
    ctx.sync()
    with pytest.raises(JxlHipError):
        ctx.read_rgb8(params, 2)          # channels must be 3 or 4
    with pytest.raises(JxlHipError):
        ctx.read_rgb8(params, 3, 10, 10)  # empty row range


@pytest.mark.parametrize("form", ["dense", "sparse", "mixed", "slots", "mixed_slots"])
def test_concurrent_submission_from_host_threads(oracle, form):
    """jxlh_ctx_create's n_slots = the number of host threads that call jxlh_submit_group* concurrently (the
    JxlParallelRunner's threads, jxl/src/api/mod.rs:77-81): six threads, one slot each, submit the groups of a frame at
    the same time (ctypes releases the GIL for the duration of a call), a seventh thread runs and reads the frame.  HIP's
    current device is per-thread state: every entry point selects the context's device itself."""
    import threading

    import helpers
    from jxl_rs_amd import Context, synth
    wl = synth.make_vardct(1500, 1100, mix=synth.MIX_ALL, seed=77, epf_iters=2)
    want, want_lf = helpers.run_oracle_frame(oracle, wl)
    nthreads = 6
    ctx = Context(0, n_slots=nthreads)
    try:
        for rep in range(3):  # epochs: the bookkeeping of one frame must not leak into the next
            p = helpers.gpu_params_from(ctx, wl)
            ctx.frame_begin(p)
            ctx.set_dequant_tables(wl.tables)
            ctx.set_lf_quantized(*wl.lf_q)
            ctx.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
            ng = wl.coeffs.shape[0]
            sparse = {g: synth.to_sparse(wl.coeffs[g]) for g in range(ng)} if form in ("sparse", "mixed", "mixed_slots") else {}
            slotted = {g: synth.to_slots(wl.coeffs[g]) for g in range(ng)} if form in ("slots", "mixed_slots") else {}
            errors = []
            start = threading.Barrier(nthreads)

            def worker(t):
                try:
                    start.wait()
                    mine = list(range(t, ng, nthreads))
                    if form == "slots" or (form == "mixed_slots" and (t + rep) % 2 == 0):
                        # one slot-bucketed batch per thread (every thread's groups: the whole frame arrives this way in
                        # "slots"; half of the threads in "mixed_slots", whose frame then takes the general route)
                        qs = [slotted[g] for g in mine]
                        assert all(len(q[3]) == 0 for q in qs)
                        if mine:
                            ctx.submit_groups_slots(np.asarray(mine, dtype=np.uint32), np.concatenate([q[0] for q in qs]),
                                                    np.concatenate([q[1].reshape(-1) for q in qs]),
                                                    np.concatenate([q[2] for q in qs]), None, slot=t)
                        mine = []
                    for g in mine:
                        use_sparse = form in ("sparse", "mixed_slots") or (form == "mixed" and (g + rep) % 2 == 0)
                        if use_sparse:
                            ctx.submit_group_sparse(g, *sparse[g], slot=t)
                        else:
                            ctx.submit_group(g, wl.coeffs[g], slot=t)
                    ctx.slot_wait(t)
                except Exception as e:  # noqa: BLE001 - reported by the main thread
                    errors.append(f"thread {t}: {type(e).__name__}: {e}")

            ths = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
            for th in ths:
                th.start()
            for th in ths:
                th.join()
            assert not errors, errors
            out = {}

            def runner():
                try:
                    ctx.frame_run()
                    ctx.sync()
                    out["planes"] = ctx.read_planes()
                    out["lf"] = ctx.read_lf()
                except Exception as e:  # noqa: BLE001
                    errors.append(f"runner: {type(e).__name__}: {e}")

            th = threading.Thread(target=runner)
            th.start()
            th.join()
            assert not errors, errors
            for c in range(3):
                assert helpers.bit_equal(out["lf"][c], want_lf[c]), f"{form} rep {rep} LF {c}"
                assert helpers.bit_equal(out["planes"][c], want[c]), \
                    f"{form} rep {rep} plane {c}: {helpers.diff_report(out['planes'][c], want[c])}"
    finally:
        ctx.close()


@pytest.mark.parametrize("size", [(2048, 1536), (4099, 1030), (3000, 8), (8, 2500), (1200, 4097), (640, 641)])
@pytest.mark.parametrize("rct", [None, (6, 0)])
def test_unsqueeze_chain_dataflow_launch(ctx, oracle, size, rct, monkeypatch):
    """The streamed levels of a chain as ONE dataflow launch (k6_unsqueeze_flow: a level starts on the rows / columns the
    level before it has finished, progress words between workgroups) against the level-by-level launches
    (JXLH_CHAIN_FLOW=0) and the oracle.  Sizes: square-ish (alternating directions), ragged (odd lines, a last partial
    group), flat and tall images (several steps of ONE direction in a row: the consumer follows the producer's own
    lines), with and without the fused last level.  Run twice: the second call meets the first one's planes and
    progress words."""
    from jxl_rs_amd.modular import ModularChain
    w, h = size
    ch = ModularChain(ctx, w, h, seed=5 * w + h, rct=rct)
    try:
        want = helpers.modular_chain_oracle(ch, oracle)
        for rep in range(2):
            for d in ch.d_out:
                d.upload(np.full(w * h, -77, np.int32))
            ch.run_chain()
            ctx.sync()
            got = ch.result()
            for c in range(3):
                assert np.array_equal(got[c], want[c]), ("flow", size, rct, c, rep, np.argwhere(got[c] != want[c])[:5])
        monkeypatch.setenv("JXLH_CHAIN_FLOW", "0")
        ch.run_chain()
        ctx.sync()
        got = ch.result()
        for c in range(3):
            assert np.array_equal(got[c], want[c]), ("levels", size, rct, c)
    finally:
        ch.free()


def test_flow_profile_reports_the_dataflow_launch(ctx, oracle):
    """jxlh_flow_profile (jxl_hip_dev.h): the per-level timeline of the dataflow launch -- every level of the launch is
    there, starts before it ends, the levels end in order, and a later level starts before the one before it has ended
    (that overlap is what the launch is for)."""
    from jxl_rs_amd.modular import ModularChain
    ch = ModularChain(ctx, 2048, 2048, seed=3, rct=(6, 0))
    try:
        assert ctx.flow_profile(True) == []           # nothing recorded yet; profiling on from here
        ch.run_chain()
        ctx.sync()
        rows = ctx.flow_profile(False)
        # 2048^2: LDS levels up to 128^2, then H 256x128 ... H 2048x1024 in the dataflow launch, V 2048^2 fused with the RCT
        assert len(rows) == 7, rows
        for r in rows:
            assert r["end_us"] > r["start_us"] >= 0.0 and r["lifetime_us_sum"] > 0.0
        ends = [r["end_us"] for r in rows]
        assert ends == sorted(ends)
        assert rows[-1]["start_us"] < rows[-2]["end_us"]
        got = ch.result()
        want = helpers.modular_chain_oracle(ch, oracle)
        for c in range(3):
            assert np.array_equal(got[c], want[c])
        ch.run_chain()                                  # profiling off again: nothing new is recorded
        ctx.sync()
        assert ctx.flow_profile(False) == []
    finally:
        ch.free()


@pytest.mark.parametrize("size", [(2051, 1030), (1030, 2051), (4097, 258)])
@pytest.mark.parametrize("rct", [None, (6, 0)])
def test_unsqueeze_chain_padded_planes_of_ragged_sizes(ctx, oracle, size, rct):
    """Ragged image sizes on planes laid out as the header advises (16-byte aligned, strides a multiple of 4 samples): the
    streamed levels then take the 16-byte movers for their complete 64-line groups and chunks, the 4-byte movers for the
    ragged last group and a line's last chunks -- both inside one launch, next to the library's own padded intermediate
    planes.  Padding is poisoned and must come back untouched."""
    from jxl_rs_amd import synth
    from helpers import DeviceArray
    w, h = size
    base, residuals, steps = synth.make_modular_planes(w, h, seed=w ^ h)
    want = [b.copy() for b in base]
    for (hz, ow, oh), res in zip(steps, residuals):
        want = [oracle.unsqueeze_h(want[c], res[c], ow) if hz else oracle.unsqueeze_v(want[c], res[c], oh) for c in range(3)]
    if rct is not None:
        want = oracle.rct(want, *rct)
    pad4 = lambda n: (n + 3) & ~3
    bufs = []

    def padded(a):
        s = max(pad4(a.shape[1]), 4)
        m = np.full((max(a.shape[0], 1), s), -12345, np.int32)
        m[:a.shape[0], :a.shape[1]] = a
        d = DeviceArray(m)
        bufs.append(d)
        return d, s

    d_base = [padded(b) for b in base]
    levels = []
    for (hz, ow, oh), res in zip(steps, residuals):
        pr = [padded(r) for r in res]
        levels.append((hz, ow, oh, [d.ptr if res[0].size else None for d, _ in pr], pr[0][1]))
    stride = pad4(w)
    d_out = [DeviceArray(np.full((h, stride), -777, np.int32)) for _ in range(3)]
    bufs += d_out
    try:
        ctx.unsqueeze_chain(levels, [d.ptr for d, _ in d_base], d_base[0][1], base[0].shape[1], base[0].shape[0],
                            [d.ptr for d in d_out], stride, rct=rct)
        ctx.sync()
        for c in range(3):
            got = d_out[c].download(np.int32, h * stride).reshape(h, stride)
            assert np.array_equal(got[:, :w], want[c]), (size, rct, c, np.argwhere(got[:, :w] != want[c])[:5])
            assert (got[:, w:] == -777).all()
    finally:
        for d in bufs:
            d.free()
