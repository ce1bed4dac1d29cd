"""Host side of the coefficient transports (no GPU): the packed forms synth.py builds for jxlh_submit_groups_sparse /
_sparse8 / _sparse4 / _slots, expanded again in numpy exactly as the device kernels of csrc/k_coeffs.hip expand them,
give back the dense slab -- every update is an addition into the group's slab, so nibble / overflow / wide splits and
any order inside a segment or slot change nothing."""
import numpy as np
import pytest

from jxl_rs_amd import synth


def _slab(seed, scale):
    rng = np.random.default_rng(seed)
    g = np.zeros((3, 65536), np.int32)
    for c in range(3):
        n = int(rng.integers(3000, 9000))
        pos = rng.choice(65536, size=n, replace=False)
        val = (rng.geometric(0.5, size=n) * rng.choice([-1, 1], size=n) * scale).astype(np.int64)
        g[c, pos] = np.clip(val, -(1 << 20), 1 << 20).astype(np.int32)
    g[0, 65535] = 5  # last position of the last slot / segment
    g[1, 0] = -8
    return g


def _add_wide(out, wide):
    for p, v in np.asarray(wide, dtype=np.uint32).reshape(-1, 2):
        out[int(p) // 65536, int(p) % 65536] += np.int32(np.uint32(v).view(np.int32))


@pytest.mark.parametrize("scale", [1, 5, 40, 3000])
def test_sparse4_form_round_trips(scale):
    g = _slab(7 + scale, scale)
    ent, counts, pos8, val8, n8, wide = synth.to_sparse4(g)
    assert counts.shape == (3, 16) and ent.dtype == np.uint16
    out = np.zeros_like(g)
    e0 = o0 = 0
    for c in range(3):
        for s in range(16):
            k = int(counts[c, s])
            e = ent[e0:e0 + k].astype(np.int32)
            e0 += k
            v = ((e << 16).astype(np.int32) >> 28)          # sign-extended nibble (k_pack_pairs4)
            np.add.at(out[c], s * 4096 + (e & 0xFFF), v)
        k = int(n8[c])
        np.add.at(out[c], pos8[o0:o0 + k].astype(np.int64), val8[o0:o0 + k].astype(np.int32))
        o0 += k
    assert e0 == len(ent) and o0 == len(pos8)
    _add_wide(out, wide)
    assert np.array_equal(out, g)
    if scale == 40:
        assert n8.sum() > 0
    if scale >= 3000:
        assert len(wide) > 0


@pytest.mark.parametrize("bits12", [False, True])
@pytest.mark.parametrize("scale", [1, 9, 400])
def test_slot_form_round_trips(scale, bits12):
    g = _slab(11 + scale, scale)
    ent, counts, n, wide = synth.to_slots(g, bits12)
    assert counts.shape == (3, 1024) and counts.dtype == np.uint8
    out = np.zeros_like(g)
    if bits12:
        b = ent.reshape(-1, 3).astype(np.uint32)
        e_all = np.empty(2 * len(b), np.uint32)
        e_all[0::2] = b[:, 0] | ((b[:, 1] & 15) << 8)       # k_unpack_entries12
        e_all[1::2] = (b[:, 1] >> 4) | (b[:, 2] << 4)
        vbits, shift = 6, 26
    else:
        e_all = ent.astype(np.uint32)
        vbits, shift = 10, 22
    e0 = 0
    for c in range(3):
        assert int(counts[c].astype(np.int64).sum()) == int(n[c])
        if bits12:
            assert n[c] % 2 == 0
        for s in range(1024):
            k = int(counts[c, s])
            e = e_all[e0:e0 + k]
            e0 += k
            v = ((e << np.uint32(32 - 6 - vbits)).astype(np.uint32).view(np.int32) >> shift)
            np.add.at(out[c], s * 64 + (e & 63).astype(np.int64), v)
    assert e0 == len(e_all)
    _add_wide(out, wide)
    assert np.array_equal(out, g)
    if scale >= 400:
        assert len(wide) > 0


def test_pair_forms_round_trip():
    g = _slab(3, 700)
    pairs, n, wide = synth.to_sparse(g)
    out = np.zeros_like(g)
    o = 0
    for c in range(3):
        p = pairs[o:o + int(n[c])]
        o += int(n[c])
        np.add.at(out[c], (p & 0xFFFF).astype(np.int64), (p >> 16).astype(np.uint16).view(np.int16).astype(np.int32))
    _add_wide(out, wide)
    assert np.array_equal(out, g)
    pos, val, n, wide = synth.to_sparse8(g)
    out = np.zeros_like(g)
    o = 0
    for c in range(3):
        np.add.at(out[c], pos[o:o + int(n[c])].astype(np.int64), val[o:o + int(n[c])].astype(np.int32))
        o += int(n[c])
    _add_wide(out, wide)
    assert np.array_equal(out, g)


def test_aligned_tilings_are_closed_under_32x32_quadrants():
    """make_vardct(aligned=True): every varblock starts on a multiple of its own size, hence lies inside one 32x32
    quadrant of a 64x64 tile -- the condition under which the strip kernel transforms a tile itself -- and the type mix
    keeps its area shares"""
    rng = np.random.default_rng(5)
    for mix in (synth.MIX_D1, synth.MIX_ALL):
        tmap, blocks = synth.random_group_tiling(rng, 32, 32, mix, aligned=True)
        area = 0
        for bx, by, t in blocks:
            cx, cy = synth.COVERED_X[t], synth.COVERED_Y[t]
            assert bx % cx == 0 and by % cy == 0
            if cx <= 4 and cy <= 4:
                assert (bx & 3) + cx <= 4 and (by & 3) + cy <= 4
            area += cx * cy
        assert area == 1024
    tmap, blocks = synth.random_group_tiling(np.random.default_rng(6), 32, 32, synth.MIX_D1, aligned=True)
    share8 = sum(1 for _, _, t in blocks if t == 0) / 1024.0
    assert 0.42 < share8 < 0.58
