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


def _slots_batch(synth, coeffs, groups, bits12=False):
    parts = [synth.to_slots(coeffs[g], bits12) for g in groups]
    for q in parts:
        assert len(q[3]) == 0
    return (np.asarray(groups, dtype=np.uint32), np.concatenate([q[0] for q in parts]),
            np.concatenate([q[1].reshape(-1) for q in parts]), np.concatenate([q[2] for q in parts]))


@pytest.mark.parametrize("first", ["slots", "pairs"])
@pytest.mark.parametrize("epf_iters", [2, 0])
def test_slot_bucketed_pass_added_to_a_resident_bucketed_frame(ctx, oracle, epf_iters, first):
    """pass 1 in a sparse form that leaves the frame resident in its bucketed form, run; pass 2 of
    SOME groups through jxlh_submit_groups_slots with JXLH_GROUP_ACCUMULATE, run.  The second submission must not
    disturb what the first frame is read through (round 4 overwrote the live slot tables at submission time: the
    dense slabs of the accumulating groups were then rebuilt from garbage); then a replacing slot-bucketed pass, a
    whole-frame resubmission, and a mixed epoch (slots + dense slab + plain pairs)."""
    from jxl_rs_amd import synth, lib
    wl = synth.make_vardct(520, 600, mix=synth.MIX_D1, seed=43, epf_iters=epf_iters)  # 3 x 3 groups
    c1, c2 = split_passes(wl.coeffs, 19)
    begin(ctx, wl)
    ng = wl.coeffs.shape[0]
    if first == "slots":
        ctx.submit_groups_slots(*_slots_batch(synth, c1, list(range(ng))), None, flags=0)
    else:
        for g in range(ng):
            ctx.submit_group_sparse(g, *synth.to_sparse(c1[g]), flags=0)
    ctx.slot_wait(0)
    ctx.frame_run()
    check(ctx, oracle_frame(oracle, wl, c1), "first pass")
    ctx.frame_run()  # the resident form is read again
    check(ctx, oracle_frame(oracle, wl, c1), "first pass, second run")
    later = [0, 4, 5]
    mixed = c1.copy()
    for g in later:
        mixed[g] = wl.coeffs[g]
    ctx.submit_groups_slots(*_slots_batch(synth, c2, later), None, flags=lib.GROUP_COMPLETE | lib.GROUP_ACCUMULATE)
    ctx.slot_wait(0)
    ctx.rerender_groups(later)
    check(ctx, oracle_frame(oracle, wl, mixed), "second pass added through the slot-bucketed form")
    # a slot-bucketed pass WITHOUT the flag replaces the group's coefficients
    ctx.submit_groups_slots(*_slots_batch(synth, c2, [1]), None, flags=lib.GROUP_COMPLETE)
    ctx.slot_wait(0)
    mixed[1] = c2[1]
    ctx.rerender_groups([1])
    check(ctx, oracle_frame(oracle, wl, mixed), "replacing pass")
    # the whole frame again in the slot-bucketed form, twice in a row (the two sets trade places every time)
    for rep, cc in enumerate((wl.coeffs, c2)):
        ctx.submit_groups_slots(*_slots_batch(synth, cc, list(range(ng)), bits12=bool(rep)), None,
                                flags=lib.GROUP_COMPLETE | (lib.GROUP_ENTRIES12 if rep else 0))
        ctx.slot_wait(0)
        ctx.frame_run()
        check(ctx, oracle_frame(oracle, wl, cc), f"whole frame resubmitted, round {rep}")
    # a mixed epoch on top of a resident slot-bucketed frame: group 2 as a dense slab, group 3 as plain pairs, group 6
    # slot-bucketed, the others keep what they hold
    mixed = c2.copy()
    for g in (2, 3, 6):
        mixed[g] = c1[g]
    ctx.submit_group(2, c1[2])
    ctx.submit_group_sparse(3, *synth.to_sparse(c1[3]))
    ctx.submit_groups_slots(*_slots_batch(synth, c1, [6]), None)
    ctx.slot_wait(0)
    ctx.frame_run()
    check(ctx, oracle_frame(oracle, wl, mixed), "mixed epoch")


def test_group_resubmitted_in_another_form_inside_one_epoch(ctx, oracle):
    """slots, then a dense slab, then ... of the same group between two runs: the last submission wins
    and the frame is not mistaken for an all-bucketed one"""
    from jxl_rs_amd import synth, lib
    wl = synth.make_vardct(300, 520, mix=synth.MIX_D1, seed=47, epf_iters=1)  # 2 x 3 groups
    c1, c2 = split_passes(wl.coeffs, 23)
    begin(ctx, wl)
    ng = wl.coeffs.shape[0]
    ctx.submit_groups_slots(*_slots_batch(synth, c1, list(range(ng))), None)
    ctx.submit_group(1, c2[1])                       # replaces the slot-bucketed submission of group 1
    ctx.slot_wait(0)
    ctx.frame_run()
    want = c1.copy()
    want[1] = c2[1]
    check(ctx, oracle_frame(oracle, wl, want), "slots then dense")


def test_slots_argument_errors_leave_the_epoch_intact(ctx, oracle):
    """a rejected jxlh_submit_groups_slots call reserves nothing -- the group can be submitted again"""
    from jxl_rs_amd import synth, lib, JxlHipError
    wl = synth.make_vardct(256, 256, mix=synth.MIX_D1, seed=5, epf_iters=0, gab=False)
    begin(ctx, wl)
    ids, ent, cnt, n = _slots_batch(synth, wl.coeffs, [0])
    odd = n.copy()
    if odd[0] % 2 == 0:
        odd[0] += 1
    with pytest.raises(JxlHipError) as e:  # 12-bit runs must be even
        ctx.submit_groups_slots(ids, ent, cnt, odd, None, flags=lib.GROUP_COMPLETE | lib.GROUP_ENTRIES12)
    assert e.value.status == lib.ERR_INVALID_ARGUMENT
    ctx.submit_groups_slots(ids, ent, cnt, n, None)   # not JXLH_ERR_BAD_STATE: nothing was reserved
    ctx.slot_wait(0)
    ctx.frame_run()
    check(ctx, oracle_frame(oracle, wl, wl.coeffs), "after a rejected call")


def test_frames_streamed_through_one_context_behind_marks(ctx, oracle):
    """jxlh_ctx_mark / jxlh_ctx_wait_mark: consecutive frames through ONE context -- frame i + 1 is submitted (slot-bucketed:
    its upload does not wait for frame i) and enqueued before the host waits for frame i's mark; every frame's planes,
    copied out asynchronously behind its kernels, must be that frame's"""
    import ctypes as C
    from jxl_rs_amd import synth, lib, JxlHipError
    from jxl_rs_amd.lib import Plane
    wl = synth.make_vardct(520, 300, mix=synth.MIX_D1, seed=71, epf_iters=2)
    begin(ctx, wl)
    ng = wl.coeffs.shape[0]
    rng = np.random.default_rng(5)
    frames = []
    for i in range(5):   # same maps, different coefficients per frame
        keep = rng.random(wl.coeffs.shape) < 0.7
        frames.append(np.where(keep, wl.coeffs, 0).astype(np.int32))
    outs = [[np.zeros((wl.ysize, wl.xsize), np.float32) for _ in range(3)] for _ in frames]
    marks = []
    for i, cf in enumerate(frames):
        ctx.submit_groups_slots(*_slots_batch(synth, cf, list(range(ng))), None, slot=i % 2)
        ctx.frame_run()
        planes = (Plane * 3)(*[Plane(o.ctypes.data, wl.xsize * 4, wl.ysize, wl.xsize * 4) for o in outs[i]])
        ctx._chk(ctx.L.jxlh_frame_read_planes_rect_async(ctx._ctx, 0, 0, wl.xsize, wl.ysize, planes), "read_rect_async")
        marks.append(ctx.mark())
        if i >= 1:
            ctx.wait_mark(marks[i - 1])
            want = oracle_frame(oracle, wl, frames[i - 1])
            for c in range(3):
                assert bit_equal(outs[i - 1][c], want[c]), f"frame {i - 1}, plane {c}: {diff_report(outs[i - 1][c], want[c])}"
    ctx.wait_mark(marks[-1])
    want = oracle_frame(oracle, wl, frames[-1])
    for c in range(3):
        assert bit_equal(outs[-1][c], want[c])
    assert marks == sorted(marks) and len(set(marks)) == len(marks)
    ctx.wait_mark(marks[0])          # long reached: returns at once
    # jxlh_slot_after: device-side ordering of uploads across contexts (here: against itself and a second context)
    from jxl_rs_amd import Context
    other = Context(0, n_slots=1)
    try:
        ctx.slot_after(0, other, 0)      # nothing submitted there yet: a no-op
        begin(other, wl)
        other.submit_groups_slots(*_slots_batch(synth, frames[0], list(range(ng))), None)
        ctx.slot_after(1, other, 0)      # slot 1 of ctx now starts behind other's upload
        ctx.submit_groups_slots(*_slots_batch(synth, frames[1], list(range(ng))), None, slot=1)
        other.frame_run(); ctx.frame_run()
        other.sync(); ctx.sync()
        for c_, f_ in ((other, frames[0]), (ctx, frames[1])):
            got = c_.read_planes()
            want = oracle_frame(oracle, wl, f_)
            for ch in range(3):
                assert bit_equal(got[ch], want[ch])
        with pytest.raises(JxlHipError):
            ctx.slot_after(9, other, 0)  # no such slot
    finally:
        other.close()
    with pytest.raises(JxlHipError):
        ctx.wait_mark(marks[-1] + 100)  # never handed out
    ctx.sync()


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
