"""The host side of the slot-bucketed form (csrc/host_pack.hip: jxlh_host_pack_slots, jxlh_slot_writer_*) -- plain CPU
code in the product library, so it runs without a GPU: against the numpy packer (synth.to_slots, the form every GPU test
submits), as a decode round trip, and the writer fed the way the entropy loop produces coefficients
(frame/group.rs:557-575: varblock by varblock, channels Y, X, B, coefficient order)."""
import numpy as np
import pytest

from jxl_rs_amd import lib as jl
from jxl_rs_amd import synth


def _group(seed, density=0.1, lo=-40, hi=41):
    rng = np.random.default_rng(seed)
    g = np.zeros((3, 65536), np.int32)
    m = rng.random(g.shape) < density
    g[m] = rng.integers(lo, hi, int(m.sum()))
    return g, rng


def _decode(entries, counts, n, wide, bits12=False):
    """what the device makes of a submission: integer sums per position"""
    out = np.zeros((3, 65536), np.int64)
    if bits12:
        b = entries.astype(np.uint32).reshape(-1, 3)
        e0 = b[:, 0] | ((b[:, 1] & 15) << 8)
        e1 = (b[:, 1] >> 4) | (b[:, 2] << 4)
        ee = np.stack([e0, e1], axis=1).reshape(-1)
        val = (ee >> 6) & 63
        val = np.where(val >= 32, val.astype(np.int64) - 64, val)
    else:
        ee = entries.astype(np.uint32)
        val = (ee >> 6) & 1023
        val = np.where(val >= 512, val.astype(np.int64) - 1024, val)
    o = 0
    for c in range(3):
        slots = np.repeat(np.arange(1024), counts[c])
        assert len(slots) == n[c]
        np.add.at(out[c], slots * 64 + (ee[o:o + n[c]] & 63), val[o:o + n[c]])
        o += int(n[c])
    for p, v in wide:
        out.reshape(-1)[int(p)] += int(np.uint32(v).view(np.int32))
    return out


@pytest.mark.parametrize("bits12", [False, True])
def test_packer_equals_numpy_packer_and_round_trips(bits12):
    g, rng = _group(1)
    g[0, 5], g[1, 100], g[2, 65535], g[1, 64], g[0, 7], g[0, 8], g[2, 3] = 2000, -30000, -513, 512, -512, 511, 100000
    a = synth.to_slots(g, bits12=bits12, split=True)
    b = jl.host_pack_slots(g, bits12=bits12)
    for x, y, what in zip(a, b, ("entries", "slot counts", "n", "wide")):
        assert np.array_equal(x, y), what
    assert len(b[3]) == (2 if bits12 else 1)       # only what cannot be split into <= 96 entries
    assert np.array_equal(_decode(*b, bits12=bits12), g)
    # without values beyond the range the split form IS the plain form
    g2, _ = _group(2, lo=-30, hi=31)
    for x, y in zip(synth.to_slots(g2, bits12=bits12), jl.host_pack_slots(g2, bits12=bits12)):
        assert np.array_equal(x, y)


def test_packer_wide_positions_carry_the_group_and_capacities_are_checked():
    g, _ = _group(3)
    g[2, 77] = 1 << 20
    e, c, n, w = jl.host_pack_slots(g, group_id=9)
    assert w.tolist() == [[(9 * 3 + 2) * 65536 + 77, 1 << 20]]
    with pytest.raises(jl.JxlHipError):
        jl.host_pack_slots(g, group_id=9, wide_capacity=0)
    with pytest.raises(jl.JxlHipError):
        jl.host_pack_slots(g, entries=np.empty(100, np.uint16))


def test_a_full_slot_sends_the_rest_to_wide():
    """a slot's count is a u8: 64 positions x 4 entries (|v| = 2000) = 256 > 255 -> the last value goes to `wide`"""
    g = np.zeros((3, 65536), np.int32)
    g[1, 128:192] = 2000
    e, c, n, w = jl.host_pack_slots(g)
    assert c[1, 2] == 252 and n.tolist() == [0, 252, 0] and len(w) == 1 and int(w[0, 1]) == 2000
    assert np.array_equal(_decode(e, c, n, w), g)


@pytest.mark.parametrize("bits12", [False, True])
def test_writer_fed_like_the_entropy_loop(bits12):
    g, rng = _group(4, density=0.15)
    g[1, 4097], g[0, 9000] = 25000, -1700
    w = jl.SlotWriter()
    w.begin_group(0, bits12=bits12)
    s = 0
    while s < 1024:     # a random tiling of the group into varblocks of 1 .. 16 slots, decoded in offset order
        ns = min(int(rng.choice([1, 1, 1, 2, 4, 8, 16])), 1024 - s)
        w.begin_varblock(s, ns)
        for c in (1, 0, 2):
            seg = g[c, s * 64:(s + ns) * 64]
            pos = np.flatnonzero(seg)
            perm = rng.permutation(len(pos))          # coefficient order is not position order
            half = len(perm) // 2
            w.add_many(c, pos[perm[:half]], seg[pos[perm[:half]]])
            for k in perm[half:]:
                w.add(c, int(pos[k]), int(seg[pos[k]]))
        s += ns
    e, cnt, n, wd = w.end_group()
    ref = synth.to_slots(g, bits12=bits12, split=True)
    assert np.array_equal(cnt, ref[1]) and np.array_equal(n, ref[2]) and np.array_equal(wd, ref[3])
    assert np.array_equal(_decode(e, cnt, n, wd, bits12=bits12), g)
    w.close()


def test_writer_call_order_errors():
    w = jl.SlotWriter()
    with pytest.raises(jl.JxlHipError):
        w.begin_varblock(0, 1)            # no group
    w.begin_group(0)
    with pytest.raises(jl.JxlHipError):
        w.add(0, 0, 1)                    # no varblock
    w.begin_varblock(4, 2)
    with pytest.raises(jl.JxlHipError):
        w.add(0, 128, 1)                  # beyond the varblock
    with pytest.raises(jl.JxlHipError):
        w.begin_varblock(5, 1)            # overlaps the one before
    with pytest.raises(jl.JxlHipError):
        w.begin_varblock(1020, 8)         # leaves the group
    w.add(3 - 1, 127, -4)
    e, cnt, n, wd = w.end_group()
    assert n.tolist() == [0, 0, 1] and cnt[2, 5] == 1
    w.close()


@pytest.mark.parametrize("bits12", [False, True])
def test_batched_packer_is_the_groups_one_after_the_other(bits12):
    """jxlh_host_pack_slots_many writes the arrays of one jxlh_submit_groups_slots call: group after group"""
    groups = np.stack([_group(20 + i, density=d)[0] for i, d in enumerate((0.1, 0.0, 0.3, 0.05))])
    groups[0, 1, 77] = 25000       # split
    groups[2, 0, 5] = 100000       # to `wide`
    ids = np.array([7, 3, 9, 0], np.uint32)
    ent, cnt, n, wide = jl.host_pack_slots_many(groups, ids, bits12=bits12)
    singles = [jl.host_pack_slots(groups[i], group_id=int(ids[i]), bits12=bits12) for i in range(4)]
    assert np.array_equal(ent, np.concatenate([s[0] for s in singles]))
    assert np.array_equal(cnt, np.stack([s[1] for s in singles]))
    assert np.array_equal(n, np.stack([s[2] for s in singles]))
    assert np.array_equal(wide, np.concatenate([s[3] for s in singles]))
    assert len(wide) >= 1 and int(wide[-1][0]) // (3 * 65536) == 9
    # capacities are checked before anything is written past them
    with pytest.raises(jl.JxlHipError):
        jl.host_pack_slots_many(groups, ids, bits12=bits12, entries=np.empty(100, np.uint8 if bits12 else np.uint16))
    with pytest.raises(jl.JxlHipError):
        jl.host_pack_slots_many(groups, ids, bits12=bits12, wide_capacity=0)


def test_baseline_scan_equals_the_avx2_scan():
    """the packer picks its zero scan at run time; JXLH_HOST_PACK_NO_AVX2=1 forces the SSE2 one"""
    import os
    import subprocess
    import sys
    code = ("import numpy as np, hashlib, sys; sys.path.insert(0, %r); from jxl_rs_amd import lib as jl\n"
            "rng = np.random.default_rng(5); g = np.zeros((3, 65536), np.int32); m = rng.random(g.shape) < 0.2\n"
            "g[m] = rng.integers(-600, 601, int(m.sum()))\n"
            "e, c, n, w = jl.host_pack_slots(g)\n"
            "print(hashlib.sha256(e.tobytes() + c.tobytes() + n.tobytes() + w.tobytes()).hexdigest())\n"
            % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    outs = []
    for v in ("0", "1"):
        env = dict(os.environ, JXLH_HOST_PACK_NO_AVX2=v)
        outs.append(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.strip())
    assert outs[0] == outs[1] and len(outs[0]) == 64
