"""Progressive passes through the C ABI: submissions without JXLH_GROUP_COMPLETE render with what has arrived, later
passes replace (dense) or add to (sparse + JXLH_GROUP_ACCUMULATE) a group's coefficients, and
jxlh_frame_rerender_groups brings exactly the affected pixels up to date -- set_buffer_for_group(.., complete = false)
+ mark_group_to_rerender of the reference (render/mod.rs:128-146, frame/decode.rs:703-711).  Expected images come
from the oracle run on the coefficients each group holds at that point."""
import numpy as np
import pytest

from helpers import bit_equal, diff_report, gpu_params_from, oracle_params_from

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from jxl_rs_amd import Context
    c = Context(0, n_slots=2)
    yield c
    c.close()


def split_passes(coeffs, seed):
    """pass 1 = a pseudo-random subset of the coefficients, pass 2 = the rest (c1 + c2 == coeffs)"""
    rng = np.random.default_rng(seed)
    keep = rng.random(coeffs.shape) < 0.5
    c1 = np.where(keep, coeffs, 0).astype(np.int32)
    return c1, (coeffs - c1).astype(np.int32)


def oracle_frame(o, wl, coeffs, **over):
    p = oracle_params_from(o, wl, **over)
    lf = o.dequant_lf(p, *wl.lf_q)
    planes, _ = o.vardct_frame(p, coeffs, wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob, lf, wl.tables,
                               num_threads=8)
    return [pl[:wl.ysize, :wl.xsize].copy() for pl in planes]


def begin(ctx, wl, **over):
    ctx.frame_begin(gpu_params_from(ctx, wl, **over))
    ctx.set_dequant_tables(wl.tables)
    ctx.set_lf_quantized(*wl.lf_q)
    ctx.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)


def check(ctx, want, what):
    ctx.sync()
    got = ctx.read_planes()
    for c in range(3):
        assert bit_equal(got[c], want[c]), f"{what}, plane {c}: {diff_report(got[c], want[c])}"


CASES = [dict(epf_iters=2, gab=True), dict(epf_iters=0, gab=False), dict(epf_iters=3, gab=True),
         dict(epf_iters=1, gab=True, flags=1), dict(epf_iters=2, gab=True, flags=1), dict(epf_iters=2, gab=True, noise=True),
         dict(epf_iters=0, gab=False, noise=True)]  # noise is added in place to the planes K1 writes


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()))
def test_dense_passes_and_rerender(ctx, oracle, case):
    from jxl_rs_amd import synth, lib
    case = dict(case)
    flags = case.pop("flags", 0)
    noise = case.pop("noise", False)
    wl = synth.make_vardct(600, 700, mix=synth.MIX_ALL, seed=77, **case)  # 3 x 3 groups
    c1, c2 = split_passes(wl.coeffs, 5)
    over = dict(flags=flags)
    if noise:  # an all-zero LUT would skip the stage
        over.update(noise=1, visible_frame_index=3)
    p = gpu_params_from(ctx, wl, **over)
    if noise:
        for i in range(8):
            p.noise_lut[i] = 0.05 + 0.01 * i
    ctx.frame_begin(p)
    ctx.set_dequant_tables(wl.tables)
    ctx.set_lf_quantized(*wl.lf_q)
    ctx.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
    ng = wl.coeffs.shape[0]
    for g in range(ng):
        ctx.submit_group(g, c1[g], flags=0)  # not complete: a first progressive pass
    ctx.slot_wait(0)
    ctx.frame_run()

    def expect(coeffs):
        want = oracle_frame(oracle, wl, coeffs)
        if noise:
            lut = np.float32([0.05 + 0.01 * i for i in range(8)])
            rnd = [oracle.noise_convolve(r) for r in oracle.noise_generate(3, 0, wl.xsize, wl.ysize)]
            want = oracle.noise_add(lut, 0.0, 1.0, want, rnd)
        return want

    check(ctx, expect(c1), "first pass")
    # the second pass reaches the centre group and a corner group: they are complete now
    later = [4, 8]
    mixed = c1.copy()
    for g in later:
        mixed[g] = wl.coeffs[g]
        ctx.submit_group(g, wl.coeffs[g], flags=lib.GROUP_COMPLETE)
    ctx.slot_wait(0)
    ctx.rerender_groups(later)
    check(ctx, expect(mixed), "after re-rendering groups 4 and 8")
    # ... and everything else
    rest = [g for g in range(ng) if g not in later]
    for g in rest:
        ctx.submit_group(g, wl.coeffs[g], flags=lib.GROUP_COMPLETE)
    ctx.slot_wait(0)
    ctx.rerender_groups(rest + rest[:2])  # duplicates are fine
    check(ctx, expect(wl.coeffs), "all passes")


@pytest.mark.parametrize("epf_iters", [2, 0])
def test_sparse_passes_accumulate_on_the_device(ctx, oracle, epf_iters):
    from jxl_rs_amd import synth, lib, JxlHipError
    wl = synth.make_vardct(520, 600, mix=synth.MIX_D1, seed=41, epf_iters=epf_iters)  # 3 x 3 groups
    c1, c2 = split_passes(wl.coeffs, 9)
    begin(ctx, wl)
    ng = wl.coeffs.shape[0]
    for g in range(ng):
        ctx.submit_group_sparse(g, *synth.to_sparse(c1[g]), flags=0)
    ctx.slot_wait(0)
    ctx.frame_run()  # every group arrived as pairs: the transforms read the bucketed pairs
    check(ctx, oracle_frame(oracle, wl, c1), "first pass")
    later = [0, 4, 5]
    mixed = c1.copy()
    for g in later:
        mixed[g] = wl.coeffs[g]
        ctx.submit_group_sparse(g, *synth.to_sparse(c2[g]), flags=lib.GROUP_COMPLETE | lib.GROUP_ACCUMULATE)
    ctx.slot_wait(0)
    ctx.rerender_groups(later)
    check(ctx, oracle_frame(oracle, wl, mixed), "second pass added on the device")
    # a pass WITHOUT the accumulate flag replaces the group's coefficients
    ctx.submit_group_sparse(1, *synth.to_sparse(c2[1]), flags=lib.GROUP_COMPLETE)
    ctx.slot_wait(0)
    mixed[1] = c2[1]
    ctx.rerender_groups([1])
    check(ctx, oracle_frame(oracle, wl, mixed), "replacing pass")
    # dense slabs cannot be added on the device
    with pytest.raises(JxlHipError) as e:
        ctx.submit_group(2, c2[2], flags=lib.GROUP_ACCUMULATE)
    assert e.value.status == lib.ERR_INVALID_ARGUMENT


def test_rerender_before_any_render_renders_the_frame(ctx, oracle):
    from jxl_rs_amd import synth
    wl = synth.make_vardct(300, 300, mix=synth.MIX_D1, seed=2, epf_iters=2)
    begin(ctx, wl)
    for g in range(wl.coeffs.shape[0]):
        ctx.submit_group(g, wl.coeffs[g])
    ctx.slot_wait(0)
    ctx.rerender_groups([0])
    check(ctx, oracle_frame(oracle, wl, wl.coeffs), "rerender as the first render")


@pytest.mark.parametrize("epf_iters,gab", [(2, True), (1, False), (3, True), (0, True), (0, False)])
def test_modular_frame_filters_constant_sigma(ctx, oracle, epf_iters, gab):
    """Gaborish / EPF on a Modular frame: SigmaSource::Constant(INV_SIGMA_NUM / epf_sigma_for_modular)
    (features/epf.rs:81-84) for every pixel, on caller-held device planes"""
    import ctypes as C
    from helpers import DeviceArray
    w, h, stride = 301, 217, 304
    rng = np.random.default_rng(5 + epf_iters)
    planes = [np.zeros((h, stride), np.float32) for _ in range(3)]
    for c in range(3):
        planes[c][:, :w] = (rng.random((h, w)) * (0.1 if c == 0 else 1.0)).astype(np.float32)
        planes[c][:, :w] += np.float32(0.2) * (np.arange(w) // 16 % 2)[None, :]  # edges for the filter to see
    p = ctx.default_params(w, h)
    p.epf_iters, p.gab = epf_iters, 1 if gab else 0
    p.epf_sigma_for_modular = 0.7
    po = oracle.default_params(w, h)
    po.epf_iters, po.gab = epf_iters, 1 if gab else 0
    sigma = np.full(((h + 7) // 8, (w + 7) // 8), np.float32(-1.1715728752538099024) / np.float32(0.7), dtype=np.float32)
    cur = [pl[:, :w].copy() for pl in planes]
    if gab:
        cur = [oracle.gaborish(cur[c], po.gab_w1[c], po.gab_w2[c]) for c in range(3)]
    for stage, need in ((0, 3), (1, 1), (2, 2)):
        if epf_iters >= need:
            cur = oracle.epf(stage, po, cur, sigma)
    tin = [DeviceArray(pl) for pl in planes]
    tout = [DeviceArray(nbytes=h * stride * 4) for _ in range(3)]
    vin = (C.c_void_p * 3)(*[t.ptr for t in tin])
    vout = (C.c_void_p * 3)(*[t.ptr for t in tout])
    ctx._chk(ctx.L.jxlh_modular_frame_filters(ctx._ctx, C.byref(p), vin, vout, w, h, stride), "modular_frame_filters")
    ctx.sync()
    for c in range(3):
        got = tout[c].download(np.float32, h * stride).reshape(h, stride)[:, :w]
        assert bit_equal(got, cur[c]), f"channel {c}: {diff_report(got, cur[c])}"
    for t in tin + tout:
        t.free()
