"""Smooth unsqueeze through the C ABI (jxlh_smooth_unsqueeze): the step a squeeze runs while its residual channel is
still all-zero -- smooth_h / smooth_v / smooth_2d_unsqueeze of the reference (modular/transforms/squeeze.rs:908-1225,
dispatched from transforms/step.rs:841-851).  Bit-exact against the oracle (FMA build, truncating convert)."""
import numpy as np
import pytest

from helpers import DeviceArray

pytestmark = pytest.mark.gpu

H, V, D2 = 0, 1, 2


@pytest.fixture(scope="module")
def ctx():
    from jxl_rs_amd import Context
    c = Context(0)
    yield c
    c.close()


def avg_shape(kind, w, h):
    return ((h + 1) // 2 if kind != H else h, (w + 1) // 2 if kind != V else w)


SHAPES = [(1, 1), (2, 1), (1, 2), (2, 2), (3, 3), (5, 4), (8, 8), (9, 7), (64, 4), (65, 5), (129, 9), (130, 17),
          (257, 63), (31, 300), (512, 512), (1000, 37)]


@pytest.mark.parametrize("kind", [H, V, D2])
@pytest.mark.parametrize("shape", SHAPES)
def test_whole_channel_bit_exact(ctx, oracle, kind, shape):
    w, h = shape
    rng = np.random.default_rng(w * 1000 + h + kind)
    avg = rng.integers(-4000, 4000, size=avg_shape(kind, w, h)).astype(np.int32)
    got = ctx.smooth_unsqueeze(kind, avg, w, h)
    want = oracle.smooth_unsqueeze(kind, avg, w, h)
    assert np.array_equal(got, want), np.argwhere(got != want)[:5]


@pytest.mark.parametrize("kind", [H, V, D2])
def test_degenerate_rectangles_are_left_untouched(ctx, oracle, kind):
    # no complete pair on a doubled axis: the reference returns before writing (squeeze.rs:921-923)
    w, h = (1, 5) if kind != V else (5, 1)
    avg = np.full(avg_shape(kind, w, h), 7, dtype=np.int32)
    out = np.full((h, w), -123, dtype=np.int32)
    ctx.smooth_unsqueeze(kind, avg, w, h, out=out)
    assert (out == -123).all()
    ref = np.full((h, w), -123, dtype=np.int32)
    oracle.smooth_unsqueeze(kind, avg, w, h, out=ref)
    assert (ref == -123).all()


@pytest.mark.parametrize("kind", [H, V, D2])
def test_grid_tiles_equal_the_whole_channel(ctx, oracle, kind):
    """A group-sized Rect of the output reads across tile edges of the average channel and clamps / mirrors at the
    CHANNEL's borders only (TiledChannelView, step.rs:372-470): tiles assemble to the whole-channel result."""
    W, Hh, g = 333, 270, 128
    rng = np.random.default_rng(77 + kind)
    avg = rng.integers(-70000, 70000, size=avg_shape(kind, W, Hh)).astype(np.int32)
    whole = oracle.smooth_unsqueeze(kind, avg, W, Hh)
    assert np.array_equal(ctx.smooth_unsqueeze(kind, avg, W, Hh), whole)
    for y0 in range(0, Hh, g):
        for x0 in range(0, W, g):
            w, h = min(g, W - x0), min(g, Hh - y0)
            tile = ctx.smooth_unsqueeze(kind, avg, w, h, x0, y0)
            assert np.array_equal(tile, whole[y0:y0 + h, x0:x0 + w]), (x0, y0)


@pytest.mark.parametrize("kind", [H, V, D2])
def test_device_pointers_and_strides(ctx, oracle, kind):
    w, h = 777, 201
    ah, aw = avg_shape(kind, w, h)
    a_stride, o_stride = aw + 13, w + 7
    rng = np.random.default_rng(5 + kind)
    avg = rng.integers(-1 << 20, 1 << 20, size=(ah, a_stride)).astype(np.int32)
    d_avg = DeviceArray(avg)
    sentinel = np.full((h, o_stride), -999, dtype=np.int32)
    d_out = DeviceArray(sentinel)
    ctx.smooth_unsqueeze_dev(kind, d_avg.ptr, a_stride, aw, ah, d_out.ptr, o_stride, w, h)
    ctx.sync()
    got = d_out.download(np.int32, h * o_stride).reshape(h, o_stride)
    assert np.array_equal(got[:, :w], oracle.smooth_unsqueeze(kind, avg[:, :aw], w, h))
    assert (got[:, w:] == -999).all()  # the stride padding is not written
    # the host-pointer path with a strided destination leaves the padding alone as well
    host = sentinel.copy()
    ctx.smooth_unsqueeze(kind, np.ascontiguousarray(avg[:, :aw]), w, h, out=host)
    assert np.array_equal(host[:, :w], got[:, :w]) and (host[:, w:] == -999).all()
    d_avg.free()
    d_out.free()


def test_constant_and_impulse_known_answers(ctx):
    """The reference's own unit tests (squeeze.rs:1243-1319), run through the kernel: a constant channel stays that
    constant; an off-centre impulse mirrors with the window."""
    for val in (-1000, -1, 0, 1, 42, 255, 10000):
        avg = np.full((6, 6), val, dtype=np.int32)
        assert (ctx.smooth_unsqueeze(D2, avg, 12, 12) == val).all()
        assert (ctx.smooth_unsqueeze(H, avg, 12, 6) == val).all()
        assert (ctx.smooth_unsqueeze(V, avg, 6, 12) == val).all()
    a = np.zeros((9, 9), dtype=np.int32)
    a[4, 4] = 10000
    up = ctx.smooth_unsqueeze(D2, a, 18, 18)
    assert np.array_equal(up, up[:, ::-1]) and np.array_equal(up, up[::-1, :])  # centre impulse: symmetric response
    assert up[8, 8] == up[9, 9] == up[8, 9] == up[9, 8] and up[8, 8] > 6000


@pytest.mark.parametrize("kind", [H, V, D2])
def test_rounding_mode_of_the_reference_build(ctx, oracle, kind):
    """`as_i32` is truncation on the reference's scalar / NEON / wasm back-ends and cvtps2dq (round to nearest even) on
    its x86 ones (jxl_simd/src/x86_64/avx.rs:580 vs scalar.rs:178): the caller picks the build it stands in for with
    JXLH_SMOOTH_CVT_NEAREST_EVEN.  Both modes against the oracle's two modes, whole channels and an offset tile, host
    and device pointers; and the two modes really differ (by at most one)."""
    w, h = 173, 97
    ih, iw = avg_shape(kind, w, h)
    avg = np.random.default_rng(21 + kind).integers(-4000, 4000, size=(ih, iw)).astype(np.int32)
    trunc = ctx.smooth_unsqueeze(kind, avg, w, h)
    rne = ctx.smooth_unsqueeze(kind, avg, w, h, cvt_rne=True)
    assert np.array_equal(trunc, oracle.smooth_unsqueeze(kind, avg, w, h, cvt_rne=False))
    assert np.array_equal(rne, oracle.smooth_unsqueeze(kind, avg, w, h, cvt_rne=True))
    d = np.abs(rne.astype(np.int64) - trunc)
    assert d.max() == 1 and 0.2 < (d != 0).mean() < 0.8
    # a grid tile at an offset, device-resident, x86 mode
    x0, y0, tw, th = 64, 32, 60, 50
    want = oracle.smooth_unsqueeze(kind, avg, tw, th, x0=x0, y0=y0, cvt_rne=True)
    d_avg, d_out = DeviceArray(avg), DeviceArray(nbytes=tw * th * 4)
    ctx.smooth_unsqueeze_dev(kind | ctx.SMOOTH_CVT_NEAREST_EVEN, d_avg.ptr, iw, iw, ih, d_out.ptr, tw, tw, th, x0=x0, y0=y0)
    ctx.sync()
    assert np.array_equal(d_out.download(np.int32, tw * th).reshape(th, tw), want)
    assert np.array_equal(want, rne[y0:y0 + th, x0:x0 + tw])
    for buf in (d_avg, d_out):
        buf.free()
    # a stray flag bit is still an invalid kind
    from jxl_rs_amd import lib, JxlHipError
    with pytest.raises(JxlHipError) as e:
        ctx.smooth_unsqueeze(kind | 0x200, avg, w, h)
    assert e.value.status == lib.ERR_INVALID_ARGUMENT


def test_progressive_preview_of_a_squeezed_channel(ctx, oracle):
    """The use this exists for: a channel squeezed h then v whose two finest residuals have not arrived.  The
    reference upsamples the quarter-size average with the 2-D kernel for the final (horizontal) step
    (SqueezeStepKind::Upsample2D, step.rs:140-145); once the vertical residual arrives the last step alone is smooth
    (Upsample1D).  Both previews stay close to the true image for smooth content."""
    from helpers import forward_squeeze_h as squeeze_h, forward_squeeze_v as squeeze_v
    n = 256
    yy, xx = np.mgrid[0:n, 0:n]
    img = (2000 * np.sin(xx / 37.0) * np.cos(yy / 23.0) + 3 * xx).astype(np.int32)
    a1, r1 = squeeze_h(img)          # a1: n x n/2
    a2, r2 = squeeze_v(a1)           # a2: n/2 x n/2
    prev2d = ctx.smooth_unsqueeze(D2, a2, n, n)
    assert np.array_equal(prev2d, oracle.smooth_unsqueeze(D2, a2, n, n))
    assert np.abs(prev2d - img).mean() < 8
    a1_back = ctx.unsqueeze(False, a2, r2, n // 2, n)
    assert np.array_equal(a1_back, a1)
    prev1d = ctx.smooth_unsqueeze(H, a1_back, n, n)
    assert np.array_equal(prev1d, oracle.smooth_unsqueeze(H, a1_back, n, n))
    assert np.abs(prev1d - img).mean() < np.abs(prev2d - img).mean()
    assert np.array_equal(ctx.unsqueeze(True, a1_back, r1, n, n), img)


# ---------------------------------------------------------------- several squeeze levels in one call
@pytest.mark.parametrize("size", [(9, 13), (16, 8), (100, 77), (128, 128), (64, 128), (127, 65), (300, 200), (1000, 130), (1, 40),
                                  (33, 1)])
@pytest.mark.parametrize("nchan", [1, 3])
def test_unsqueeze_levels_equals_the_steps_one_by_one(ctx, oracle, size, nchan):
    """jxlh_unsqueeze_levels on the default squeeze chain of a w x h image (squeeze.rs:71-105): planes up to 128 x 128
    run as one launch with the planes in LDS, larger chains level by level through scratch -- either way the result
    is the oracle's step-by-step one.  Residual planes have padded strides."""
    from jxl_rs_amd import synth
    w, h = size
    base, residuals, steps = synth.make_modular_planes(w, h, seed=w * 131 + h, nchan=nchan)
    bh, bw = base[0].shape
    want = [b.copy() for b in base]
    for (horizontal, ow, oh), res in zip(steps, residuals):
        want = [oracle.unsqueeze_h(want[c], res[c], ow) if horizontal else oracle.unsqueeze_v(want[c], res[c], oh)
                for c in range(nchan)]
    if not steps:
        pytest.skip("no squeeze level at this size")
    dev_base = [DeviceArray(b) for b in base]
    levels, keep = [], []
    for (horizontal, ow, oh), res in zip(steps, residuals):
        rstride = max(res[0].shape[1], 1) + 3
        planes = []
        for c in range(nchan):
            padded = np.zeros((max(res[c].shape[0], 1), rstride), np.int32)
            padded[:res[c].shape[0], :res[c].shape[1]] = res[c]
            d = DeviceArray(padded)
            keep.append(d)
            planes.append(d.ptr)
        levels.append((horizontal, ow, oh, planes + [None] * (3 - nchan), rstride))
    o_stride = w + 5
    dev_out = [DeviceArray(np.full((h, o_stride), -9, np.int32)) for _ in range(nchan)]
    ctx.unsqueeze_levels(levels, [d.ptr for d in dev_base], bw, bw, bh, [d.ptr for d in dev_out], o_stride)
    ctx.sync()
    for c in range(nchan):
        got = dev_out[c].download(np.int32, h * o_stride).reshape(h, o_stride)
        assert np.array_equal(got[:, :w], want[c]), (c, np.argwhere(got[:, :w] != want[c])[:4])
        assert (got[:, w:] == -9).all()
    for d in dev_base + dev_out + keep:
        d.free()


def test_unsqueeze_levels_rejects_inconsistent_geometry(ctx):
    from jxl_rs_amd import lib, JxlHipError
    base = DeviceArray(np.zeros((8, 8), np.int32))
    res = DeviceArray(np.zeros((8, 8), np.int32))
    out = DeviceArray(np.zeros((8, 32), np.int32))
    with pytest.raises(JxlHipError) as e:   # 8 x 8 averages cannot produce a 32-wide level
        ctx.unsqueeze_levels([(True, 32, 8, [res.ptr, None, None], 8)], [base.ptr], 8, 8, 8, [out.ptr], 32)
    assert e.value.status == lib.ERR_INVALID_ARGUMENT
    for d in (base, res, out):
        d.free()
