"""Consumes the stage vectors the REFERENCE ITSELF produces (oracle/ref_dump/run.sh -> tests/golden/ref_stage_vectors):
EPF0/1/2 and Gaborish on a ragged 3-channel image with a variable sigma map, every RCT op, one horizontal and one
vertical inverse squeeze step.  With the vectors present the oracle (and, on the GPU box, the device path) is held
to the reference's own numbers; without them -- no Rust toolchain has run the recipe yet -- the tests are reported
as skipped: those rows stay "parity unpinned" (DESIGN.md section 4)."""
import glob
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VEC_DIR = os.environ.get("JXL_REF_VEC_DIR") or os.path.join(ROOT, "tests", "golden", "ref_stage_vectors")


def read_vec(name):
    path = os.path.join(VEC_DIR, name + ".vec")
    with open(path, "rb") as f:
        head = f.readline().decode().split()
        assert head[0] == "JXLVEC1", path
        dtype = {"f32": "<f4", "i32": "<i4"}[head[1]]
        dims = [int(x) for x in head[3:3 + int(head[2])]]
        data = np.frombuffer(f.read(), dtype=dtype)
    return data.reshape(dims).copy()


have = bool(glob.glob(os.path.join(VEC_DIR, "*.vec")))
needs_vectors = pytest.mark.skipif(
    not have, reason="parity unpinned: tests/golden/ref_stage_vectors is empty -- run oracle/ref_dump/run.sh where cargo exists")


def test_recipe_is_complete():
    """the recipe itself is part of the repository: every dump module names the reference file it extends, and that
    file exists in the reference tree when the tree is present"""
    d = os.path.join(ROOT, "oracle", "ref_dump")
    mods = sorted(glob.glob(os.path.join(d, "*_dump.rs")))
    assert len(mods) >= 3 and os.path.exists(os.path.join(d, "run.sh")) and os.path.exists(os.path.join(d, "vec_io.rs"))
    for m in mods:
        first = open(m).readline()
        assert first.startswith("// APPEND-TO: jxl/src/"), m
        target = first.split("APPEND-TO:")[1].strip()
        if os.path.isdir("/root/reference"):
            assert os.path.exists(os.path.join("/root/reference", target)), target


def _stage_inputs():
    planes = read_vec("stages_input")
    sigma = read_vec("stages_inv_sigma")
    return [planes[c] for c in range(3)], sigma


def _equal_to_either_build(got_fused, got_unfused, want, what):
    if np.array_equal(got_fused.view(np.uint32), want.view(np.uint32)):
        return
    if np.array_equal(got_unfused.view(np.uint32), want.view(np.uint32)):
        return
    err = min(np.abs(got_fused - want).max(), np.abs(got_unfused - want).max())
    raise AssertionError(f"{what}: equals neither contraction build of the oracle (max abs {err})")


@needs_vectors
@pytest.mark.parametrize("stage", [0, 1, 2])
def test_epf_matches_reference(oracle, oracle_unfused, stage):
    planes, sigma = _stage_inputs()
    want = read_vec(f"stages_epf{stage}")
    h, w = planes[0].shape
    outs = []
    for o in (oracle, oracle_unfused):
        p = o.default_params(w, h)
        outs.append(o.epf(stage, p, planes, np.ascontiguousarray(sigma[:, :(w + 7) // 8])))
    for c in range(3):
        _equal_to_either_build(outs[0][c], outs[1][c], want[c], f"EPF{stage} channel {c}")


@needs_vectors
def test_gaborish_matches_reference(oracle, oracle_unfused):
    planes, _ = _stage_inputs()
    want = read_vec("stages_gaborish")[0]
    a = oracle.gaborish(planes[0], 0.115169525, 0.061248592)
    b = oracle_unfused.gaborish(planes[0], 0.115169525, 0.061248592)
    _equal_to_either_build(a, b, want, "Gaborish")


@needs_vectors
def test_rct_matches_reference(oracle):
    base = read_vec("rct_input")
    for op in range(7):
        want = read_vec(f"rct_op{op}")
        got = oracle.rct([base[0], base[1], base[2]], op, 0)
        for c in range(3):
            assert np.array_equal(got[c], want[c]), (op, c)


@needs_vectors
def test_unsqueeze_matches_reference(oracle):
    a, r, want = read_vec("unsqueeze_h_avg"), read_vec("unsqueeze_h_res"), read_vec("unsqueeze_h_out")
    assert np.array_equal(oracle.unsqueeze_h(a, r, want.shape[1]), want)
    a, r, want = read_vec("unsqueeze_v_avg"), read_vec("unsqueeze_v_res"), read_vec("unsqueeze_v_out")
    assert np.array_equal(oracle.unsqueeze_v(a, r, want.shape[0]), want)


@needs_vectors
@pytest.mark.gpu
def test_device_path_matches_reference_vectors():
    """the same vectors through the C ABI (the stage hooks): the device must equal the reference bit for bit where the
    fused oracle does"""
    import jxl_rs_amd
    ctx = jxl_rs_amd.Context(0, 1)
    try:
        planes, sigma = _stage_inputs()
        h, w = planes[0].shape
        p = ctx.default_params(w, h)
        for stage in range(3):
            want = read_vec(f"stages_epf{stage}")
            got = ctx.stage_epf(stage, p, planes, np.ascontiguousarray(sigma[:, :(w + 7) // 8]))
            for c in range(3):
                assert np.abs(got[c] - want[c]).max() < 2e-6, (stage, c)
        base = read_vec("rct_input")
        for op in range(7):
            got = ctx.rct([base[0], base[1], base[2]], op, 0)
            for c in range(3):
                assert np.array_equal(got[c], read_vec(f"rct_op{op}")[c])
        a, r, want = read_vec("unsqueeze_h_avg"), read_vec("unsqueeze_h_res"), read_vec("unsqueeze_h_out")
        assert np.array_equal(ctx.unsqueeze(True, a, r, want.shape[1], want.shape[0]), want)
        a, r, want = read_vec("unsqueeze_v_avg"), read_vec("unsqueeze_v_res"), read_vec("unsqueeze_v_out")
        assert np.array_equal(ctx.unsqueeze(False, a, r, want.shape[1], want.shape[0]), want)
    finally:
        ctx.close()
