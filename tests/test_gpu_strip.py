"""The strip kernel (k123_strip, JXLH_FRAME_STRIP: dequantisation + IDCT + Gaborish / EPF in ONE persistent launch, no
intermediate planes in HBM) against the CPU oracle and against the two-kernel path, bit for bit:
every stage subset, frame sizes that are not multiples of the 64-pixel strips / tiles / bands, tiles the strip kernel
transforms itself next to tiles it only loads (varblocks that leave their 32x32 quadrant, special and large
transforms), run-to-run identity, a progressive re-render behind a strip run."""
import numpy as np
import pytest

from helpers import bit_equal, diff_report, run_gpu_frame, run_oracle_frame, upload_frame

pytestmark = pytest.mark.gpu
STRIP = 4  # JXLH_FRAME_STRIP


@pytest.fixture(scope="module")
def ctx():
    from jxl_rs_amd import Context
    c = Context(0, n_slots=1)
    yield c
    c.close()


def _check(ctx, oracle, wl, what, expect_class_tiles=None):
    got, got_lf = run_gpu_frame(ctx, wl, flags=STRIP)
    ran, tiles, by_class = ctx.frame_path()
    assert ran, f"{what}: the strip kernel did not run"
    assert tiles == ((wl.xblocks + 7) // 8) * ((wl.yblocks + 7) // 8)
    if expect_class_tiles == 0:
        assert by_class == 0, f"{what}: {by_class} of {tiles} tiles were left to the class kernels"
    elif expect_class_tiles:
        assert by_class > 0, f"{what}: no tile took the class-kernel route"
    want, want_lf = run_oracle_frame(oracle, wl, num_threads=16)
    for c in range(3):
        assert bit_equal(got_lf[c], want_lf[c]), f"{what}: LF ch{c}"
        assert bit_equal(got[c], want[c]), f"{what}: plane {c}: {diff_report(got[c], want[c])}"
    return got


@pytest.mark.parametrize("w,h,mixname,epf,gab", [
    (64, 64, "dct8", 2, True),          # one tile, one band, one strip (no exchange at all)
    (128, 128, "dct8", 2, True),
    (128, 128, "dct8", 0, True),        # Gaborish only (BASELINE configs[1]'s stage list)
    (128, 128, "dct8", 1, False),       # EPF1 only
    (200, 136, "d1", 2, True),          # partial last strip and tile, frame edge inside a block
    (333, 77, "d1", 2, True),
    (520, 328, "d1", 1, True),          # Gaborish + EPF1
    (520, 328, "d1", 2, False),         # EPF1 + EPF2
    (1024, 768, "d1", 2, True),
    (2048, 520, "d1", 2, True),         # several bands: seeds / extra steps between them
])
def test_strip_aligned_frames_vs_oracle(ctx, oracle, w, h, mixname, epf, gab):
    from jxl_rs_amd import synth
    mix = {"dct8": synth.MIX_DCT8, "d1": synth.MIX_D1}[mixname]
    wl = synth.make_vardct(w, h, mix=mix, seed=w * 7 + h, epf_iters=epf, gab=gab, aligned=True)
    _check(ctx, oracle, wl, f"{w}x{h} {mixname} epf{epf} gab{int(gab)}", expect_class_tiles=0)


def test_strip_mixed_tiles_vs_oracle(ctx, oracle):
    """unaligned varblocks and every transform type: tiles transformed by the strip kernel next to tiles it only loads"""
    from jxl_rs_amd import synth
    wl = synth.make_vardct(512, 512, mix=synth.MIX_D1, seed=5, epf_iters=2, gab=True, aligned=False)
    _check(ctx, oracle, wl, "d1 unaligned", expect_class_tiles=1)
    wl = synth.make_vardct(1024, 1024, mix=synth.MIX_ALL, seed=6, epf_iters=2, gab=True, aligned=True)
    _check(ctx, oracle, wl, "all types", expect_class_tiles=1)
    # a frame whose left half is aligned (strip tiles) and right half is not (class-kernel tiles): both kinds exchange
    # edge columns with each other
    a = synth.make_vardct(512, 256, mix=synth.MIX_D1, seed=7, epf_iters=2, gab=True, aligned=True)
    b = synth.make_vardct(512, 256, mix=synth.MIX_D1, seed=7, epf_iters=2, gab=True, aligned=False)
    a.transform_map[:, 32:] = b.transform_map[:, 32:]
    a.raw_quant[:, 32:] = b.raw_quant[:, 32:]
    a.coeffs[1] = b.coeffs[1]
    got = _check(ctx, oracle, a, "half aligned", expect_class_tiles=1)
    _, tiles, by_class = ctx.frame_path()
    assert 0 < by_class < tiles
    assert all(np.isfinite(g).all() for g in got)


def test_strip_equals_two_kernel_path_and_is_deterministic(ctx):
    from jxl_rs_amd import synth
    wl = synth.make_vardct(4096, 4096, mix=synth.MIX_D1, seed=11, unique_groups=24, epf_iters=2, gab=True, aligned=True)
    a, _ = run_gpu_frame(ctx, wl, flags=STRIP)
    assert ctx.frame_path()[0]
    ctx.frame_run()
    ctx.sync()
    again = ctx.read_planes()
    b, _ = run_gpu_frame(ctx, wl)
    assert not ctx.frame_path()[0]
    for c in range(3):
        assert bit_equal(a[c], again[c]), "run-to-run"
        assert bit_equal(a[c], b[c]), f"strip vs two-kernel path, plane {c}: {diff_report(a[c], b[c])}"


def test_strip_4k_whole_frame_vs_oracle(ctx, oracle):
    from jxl_rs_amd import synth
    wl = synth.make_vardct(4096, 4096, mix=synth.MIX_D1, seed=12, unique_groups=24, epf_iters=2, gab=True, aligned=True)
    _check(ctx, oracle, wl, "4096^2 d1 aligned", expect_class_tiles=0)


def test_rerender_behind_a_strip_run(ctx, oracle):
    """a strip run leaves no unfiltered planes behind: re-rendering groups renders the frame again, same bits"""
    from jxl_rs_amd import synth
    wl = synth.make_vardct(768, 512, mix=synth.MIX_D1, seed=13, epf_iters=2, gab=True, aligned=True)
    upload_frame(ctx, wl, flags=STRIP)
    ctx.frame_run()
    ctx.sync()
    first = ctx.read_planes()
    ctx.submit_group(1, wl.coeffs[1])
    ctx.slot_wait(0)
    ctx.rerender_groups([1])
    ctx.sync()
    second = ctx.read_planes()
    want, _ = run_oracle_frame(oracle, wl, num_threads=8)
    for c in range(3):
        assert bit_equal(first[c], want[c]) and bit_equal(second[c], want[c])
