"""Round 6: the slot-bucketed form on content that is not the synthetic d1 frame.

  - per-group routing: groups that arrive in another form (dense slab, plain pairs), carry a value in `wide`, or add a
    pass to earlier content are read from their dense slabs while the rest of the frame is still read in place
    (csrc/abi_frame.hip:run_prologue, k1_scan's dense-route lists) -- the reference's `current_coeffs[idx] += coeff` on
    arbitrary i32 (frame/group.rs:568-572) must not cost the whole frame its fast path;
  - wide values split into repeated in-range entries by the C packer (jxlh_host_pack_slots) stay in place;
  - varblocks with more than 1023 entries per channel (legal: repeated positions, u8 counts per slot);
  - coefficient densities far from d1: every batch beyond the direct path's depth takes the dense dequantisation pass.

Every case is compared bit for bit with the dense-slab submission of the same coefficients (which the other test files
compare with the oracle), and the first one with the oracle itself."""
import numpy as np
import pytest

from helpers import bit_equal, diff_report, gpu_params_from, run_gpu_frame, run_oracle_frame

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import jxl_rs_amd
    c = jxl_rs_amd.Context(0, 2)
    yield c
    c.close()


@pytest.fixture(scope="module")
def oracle():
    from oracle.oracle import Oracle
    return Oracle(fused=True)


def _begin(ctx, wl, flags=0):
    p = gpu_params_from(ctx, wl)
    p.flags = flags
    ctx.frame_begin(p)
    ctx.set_dequant_tables(wl.tables)
    ctx.set_lf_quantized(*wl.lf_q)
    ctx.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)


def _submit_slots(ctx, groups, parts, flags=None, slot=0):
    from jxl_rs_amd import lib as jl
    wide = [parts[g][3] for g in groups if len(parts[g][3])]
    ctx.submit_groups_slots(np.asarray(groups, dtype=np.uint32), np.concatenate([parts[g][0] for g in groups]),
                            np.concatenate([parts[g][1].reshape(-1) for g in groups]),
                            np.concatenate([parts[g][2] for g in groups]), np.concatenate(wide) if wide else None,
                            slot=slot, flags=jl.GROUP_COMPLETE if flags is None else flags)


def _check(ctx, want, what):
    ctx.frame_run()
    ctx.sync()
    got = ctx.read_planes()
    for c in range(3):
        assert bit_equal(got[c], want[c]), f"{what}, plane {c}: {diff_report(got[c], want[c])}"


@pytest.mark.parametrize("mix,size,seed", [("MIX_D1", (1280, 1024), 3), ("MIX_ALL", (1024, 768), 4), ("MIX_D1", (600, 300), 5)])
def test_per_group_routing(ctx, oracle, mix, size, seed):
    from jxl_rs_amd import lib as jl
    from jxl_rs_amd import synth
    w, h = size
    wl = synth.make_vardct(w, h, mix=getattr(synth, mix), seed=seed, epf_iters=2)
    ng = wl.coeffs.shape[0]
    rng = np.random.default_rng(seed)
    # one group gets values no entry holds and the C packer cannot split (beyond kMaxSplit entries): `wide`
    g_wide = ng - 1
    nz = np.flatnonzero(wl.coeffs[g_wide].reshape(-1))
    big = wl.coeffs.copy()
    sel = rng.choice(nz, size=6, replace=False)
    big[g_wide].reshape(-1)[sel] = rng.integers(60000, 900000, size=6) * rng.choice([-1, 1], size=6)
    wl.coeffs = big
    want, _ = run_gpu_frame(ctx, wl)
    oref, _ = run_oracle_frame(oracle, wl)
    for c in range(3):
        assert bit_equal(want[c], oref[c]), f"dense submission vs oracle, plane {c}: {diff_report(want[c], oref[c])}"
    parts = {g: jl.host_pack_slots(wl.coeffs[g], group_id=g) for g in range(ng)}
    assert len(parts[g_wide][3]) == 6 and all(len(parts[g][3]) == 0 for g in range(ng - 1))
    g_dense, g_pairs = 0, 1 if ng > 2 else None
    slotted = [g for g in range(ng) if g not in (g_dense, g_pairs)]
    ctx.kernel_timing_reset()
    ctx.kernel_timing(True)
    _begin(ctx, wl)
    _submit_slots(ctx, slotted, parts)
    ctx.submit_group(g_dense, wl.coeffs[g_dense])
    if g_pairs is not None:
        ctx.submit_group_sparse(g_pairs, *synth.to_sparse(wl.coeffs[g_pairs]))
    ctx.slot_wait(0)
    _check(ctx, want, "one dense slab + one plain-pairs group + one group with wide values, the rest in place")
    kt = ctx.kernel_times()
    ctx.kernel_timing(False)
    if 2 * (ng - 3) >= ng:
        assert "k_sort_sparse" not in kt, sorted(kt)       # the frame stayed in the in-place form ...
        assert "k_expand_sparse" in kt, sorted(kt)         # ... and only the routed groups were expanded
    _check(ctx, want, "the same frame run again without resubmission")
    # next epoch: every group slot-bucketed and self-contained except the wide one -> still routed per group;
    # then a clean frame (wide values clipped): back to the all-in-place form
    _submit_slots(ctx, list(range(ng)), parts)
    ctx.slot_wait(0)
    _check(ctx, want, "resubmitted: all slots, one group with wide values")
    clean = wl.coeffs.copy()
    clean[g_wide].reshape(-1)[sel] = 7
    wl.coeffs = clean
    want2, _ = run_gpu_frame(ctx, wl)
    _begin(ctx, wl)
    _submit_slots(ctx, list(range(ng)), {g: jl.host_pack_slots(wl.coeffs[g], group_id=g) for g in range(ng)})
    ctx.slot_wait(0)
    _check(ctx, want2, "clean frame after a routed one")


def test_added_pass_routes_only_its_groups(ctx):
    """JXLH_GROUP_ACCUMULATE on some groups of a frame that is resident in place: those groups' earlier content is
    expanded from the live set, the new pass added on top, and they are read from their slabs; the other groups
    (resubmitted whole) stay in place.  A third epoch then leaves the bucketed form altogether (dense slabs for a few
    groups only): the routed groups' slabs are the truth, the others are expanded from their entries."""
    from jxl_rs_amd import lib as jl
    from jxl_rs_amd import synth
    wl = synth.make_vardct(1024, 768, mix=synth.MIX_D1, seed=21, epf_iters=1)
    ng = wl.coeffs.shape[0]
    rng = np.random.default_rng(2)
    full = wl.coeffs.copy()
    split = (rng.random(full.shape) < 0.4) & (full != 0)
    part_b = np.where(split, rng.integers(-3, 4, size=full.shape), 0).astype(np.int32)
    part_a = (full - part_b).astype(np.int32)
    want_full, _ = run_gpu_frame(ctx, wl)
    wl.coeffs = part_a
    want_a, _ = run_gpu_frame(ctx, wl)
    _begin(ctx, wl)
    ids = list(range(ng))
    _submit_slots(ctx, ids, {g: jl.host_pack_slots(part_a[g], group_id=g) for g in ids})
    ctx.slot_wait(0)
    _check(ctx, want_a, "first pass, all in place")
    added = [g for g in ids if g % 4 == 1]
    whole = [g for g in ids if g % 4 != 1]
    _submit_slots(ctx, whole, {g: jl.host_pack_slots(full[g], group_id=g) for g in whole})
    _submit_slots(ctx, added, {g: jl.host_pack_slots(part_b[g], group_id=g) for g in added},
                  flags=jl.GROUP_COMPLETE | jl.GROUP_ACCUMULATE)
    ctx.slot_wait(0)
    ctx.kernel_timing_reset()
    ctx.kernel_timing(True)
    _check(ctx, want_full, "second pass added to a quarter of the groups")
    kt = ctx.kernel_times()
    ctx.kernel_timing(False)
    assert "k_sort_sparse" not in kt and "k_entries_to_pairs" in kt, sorted(kt)
    # third epoch: two groups replaced by dense slabs of the first pass, nothing else resubmitted
    for g in (0, 1):
        ctx.submit_group(g, part_a[g])
    ctx.slot_wait(0)
    mixed = full.copy()
    mixed[0], mixed[1] = part_a[0], part_a[1]
    wl.coeffs = mixed
    ctx2_want, _ = None, None
    ctx.frame_run()
    ctx.sync()
    got = ctx.read_planes()
    want_mixed, _ = run_gpu_frame(ctx, wl)
    for c in range(3):
        assert bit_equal(got[c], want_mixed[c]), f"leaving the routed form, plane {c}: {diff_report(got[c], want_mixed[c])}"


@pytest.mark.parametrize("bits12", [False, True])
def test_split_wide_values_stay_in_place(ctx, bits12):
    """values far outside the entries' range, split by jxlh_host_pack_slots into repeated in-range entries: no `wide`
    list, no expansion, the frame is read in place (the varblocks that hold them take the dense dequantisation pass)"""
    from jxl_rs_amd import lib as jl
    from jxl_rs_amd import synth
    wl = synth.make_vardct(1024, 512, mix=synth.MIX_D1, seed=9, epf_iters=2)
    ng = wl.coeffs.shape[0]
    rng = np.random.default_rng(9)
    nz = np.flatnonzero(wl.coeffs.reshape(-1))
    sel = rng.choice(nz, size=len(nz) // 500, replace=False)   # 2e-3 of the entries
    hi = 1800 if bits12 else 30000
    wl.coeffs.reshape(-1)[sel] = rng.integers(hi // 15, hi, size=len(sel)) * rng.choice([-1, 1], size=len(sel))
    want, _ = run_gpu_frame(ctx, wl)
    parts = {g: jl.host_pack_slots(wl.coeffs[g], group_id=g, bits12=bits12) for g in range(ng)}
    assert all(len(parts[g][3]) == 0 for g in range(ng))
    assert sum(int(parts[g][2].sum()) for g in range(ng)) > len(nz) + 5 * len(sel)
    ctx.kernel_timing_reset()
    ctx.kernel_timing(True)
    _begin(ctx, wl)
    _submit_slots(ctx, list(range(ng)), parts, flags=jl.GROUP_COMPLETE | (jl.GROUP_ENTRIES12 if bits12 else 0))
    ctx.slot_wait(0)
    _check(ctx, want, "split values")
    kt = ctx.kernel_times()
    ctx.kernel_timing(False)
    assert not any(k in kt for k in ("k_sort_sparse", "k_expand_sparse", "k_entries_to_pairs")), sorted(kt)


def _merge(a, b):
    """two slot forms of one group -> one list per (channel, slot) (two passes' updates in ONE submission)"""
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
    return np.concatenate(out), cnt.astype(np.uint8), (na + nb_).astype(np.uint32), wa


@pytest.mark.parametrize("dense_dequant", [False, True])
def test_more_than_1023_entries_in_a_varblock(ctx, dense_dequant):
    """16x32 / 32x16 / 32x32 varblocks may legally hold 8-16 x 255 entries per channel (repeated positions).
    Dense 32-point varblocks submitted as TWO passes in one list: up to 2048 entries per channel in a 32x32."""
    from jxl_rs_amd import lib as jl
    from jxl_rs_amd import synth
    mix = {5: 0.4, 10: 0.2, 11: 0.2, 0: 0.2}
    wl = synth.make_vardct(512, 512, mix=mix, seed=31, epf_iters=0, gab=False)
    ng = wl.coeffs.shape[0]
    rng = np.random.default_rng(31)
    a = rng.integers(-200, 201, size=wl.coeffs.shape).astype(np.int32)       # every position non-zero, nearly
    b = rng.integers(-200, 201, size=wl.coeffs.shape).astype(np.int32)
    b = np.where(rng.random(b.shape) < 0.02, -a, b).astype(np.int32)          # some positions cancel to zero
    # (a run holds at most 65536 entries: both passes dense in the first 30000 coefficients of a channel, sparse behind)
    tail = np.arange(65536) >= 30000
    a[:, :, tail] = np.where(rng.random(a[:, :, tail].shape) < 0.05, a[:, :, tail], 0)
    b[:, :, tail] = 0
    wl.coeffs = (a + b).astype(np.int32)
    want, _ = run_gpu_frame(ctx, wl)
    parts = {g: _merge(synth.to_slots(a[g]), synth.to_slots(b[g])) for g in range(ng)}
    assert max(int(parts[g][1].astype(np.int64).reshape(3, 64, 16).sum(axis=2).max()) for g in range(ng)) > 1023
    _begin(ctx, wl, flags=jl.FRAME_DENSE_DEQUANT if dense_dequant else 0)
    _submit_slots(ctx, list(range(ng)), parts)
    ctx.slot_wait(0)
    _check(ctx, want, "two dense passes in one list")


@pytest.mark.parametrize("mix", ["d1", "all"])
@pytest.mark.parametrize("density", [0.02, 0.11, 0.15, 0.25, 0.35, 0.6])
def test_coefficient_density_far_from_d1(ctx, density, mix):
    """the direct path holds 3-6 entries per lane and channel; a frame of another density falls back batch by batch
    (the fallback launch; from 0.125 entries per coefficient on the 16..32-point classes take the dense pass outright,
    from 0.25 on the 8x8 class runs its over-depth batches inline) and gives the dense submission's bits; `all`: with
    the special and large transform types, which read their dense slabs next to it"""
    from jxl_rs_amd import lib as jl
    from jxl_rs_amd import synth
    wl = synth.make_vardct(1024, 768, mix=synth.MIX_D1 if mix == "d1" else synth.MIX_ALL, seed=17, epf_iters=1)
    ng = wl.coeffs.shape[0]
    rng = np.random.default_rng(int(density * 100))
    m = rng.random(wl.coeffs.shape) < density
    wl.coeffs = np.where(m, rng.integers(-30, 31, size=wl.coeffs.shape), 0).astype(np.int32)
    want, _ = run_gpu_frame(ctx, wl)
    _begin(ctx, wl)
    _submit_slots(ctx, list(range(ng)), {g: jl.host_pack_slots(wl.coeffs[g], group_id=g) for g in range(ng)})
    ctx.slot_wait(0)
    _check(ctx, want, f"density {density}")
    _check(ctx, want, f"density {density}, second run")
