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
    # every stage VERDICT r02 listed is covered: unit-test dumps (EPF / Gaborish / RCT / squeeze / palette) and the
    # instrumented functions (LF smoothing, dequant_lf, sigma map, one whole group through decode_vardct_group)
    names = {os.path.basename(m) for m in mods}
    assert {"stages_dump.rs", "rct_dump.rs", "squeeze_dump.rs", "palette_dump.rs", "frame_dump.rs"} <= names
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_dump_instrument", os.path.join(d, "instrument.py"))
    ins = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ins)
    files = {pth for pth, _, _, _ in ins.PATCHES}
    assert {"jxl/src/frame/adaptive_lf_smoothing.rs", "jxl/src/frame/modular/mod.rs", "jxl/src/features/epf.rs",
            "jxl/src/frame/group.rs"} <= files
    if os.path.isdir("/root/reference"):  # the anchors still match the reference tree (nothing is modified)
        assert ins.apply("/root/reference", check_only=True) == []


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
def test_device_path_matches_reference_vectors(oracle):
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
            sig = np.ascontiguousarray(sigma[:, :(w + 7) // 8])
            got = ctx.stage_epf(stage, p, planes, sig)
            for c in range(3):
                # the device is the FMA build: wherever the fused oracle equals the reference bit for bit (x86 AVX2 /
                # NEON back-ends) so must the device; a reference run on a back-end without FMA equals the unfused
                # oracle instead, and the device then differs from the vectors exactly as the fused oracle does
                ref_fused = oracle.epf(stage, oracle.default_params(w, h), planes, sig)[c]
                assert np.array_equal(got[c].view(np.uint32), ref_fused.view(np.uint32)), (stage, c, "device != fused oracle")
                if np.array_equal(ref_fused.view(np.uint32), want[c].view(np.uint32)):
                    assert np.array_equal(got[c].view(np.uint32), want[c].view(np.uint32)), (stage, c)
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


# ------------------------------------------------------------------------------------------------------------------
# Vectors of the INSTRUMENTED reference (oracle/ref_dump/instrument.py + frame_dump.rs): functions that only run inside a
# decoded frame.  Prefix = the decoded file (vardct444: green_queen_vardct_e3.jxl, jpeg420: multiple_lf_420.jxl).
def _have(name):
    return os.path.exists(os.path.join(VEC_DIR, name + ".vec"))


def _params_from_header(o, prefix, w, h):
    """oracle frame parameters rebuilt from the dumped header fields"""
    hi = read_vec(prefix + "_header_ints")
    hf = read_vec(prefix + "_header_floats")
    p = o.default_params(w, h)
    p.global_scale, p.quant_lf, p.x_qm_scale, p.b_qm_scale = int(hi[0]), int(hi[1]), int(hi[2]), int(hi[3])
    p.color_factor, p.ytox_lf, p.ytob_lf, p.epf_iters, p.gab = int(hi[4]), int(hi[5]), int(hi[6]), int(hi[7]), int(hi[8])
    for c in range(3):
        p.lf_quant_factors[c] = float(hf[c])
        p.gab_w1[c], p.gab_w2[c] = float(hf[3 + 2 * c]), float(hf[4 + 2 * c])
        p.epf_channel_scale[c] = float(hf[17 + c])
    for i in range(8):
        p.epf_sharp_lut[i] = float(hf[9 + i])
    p.epf_quant_mul, p.epf_pass0_sigma_scale, p.epf_pass2_sigma_scale, p.epf_border_sad_mul = [float(v) for v in hf[20:24]]
    gp = read_vec(prefix + "_group_params")
    for i in range(4):
        p.quant_biases[i] = float(gp[i])
    p.base_correlation_x, p.base_correlation_b = float(gp[7]), float(gp[8])
    return p


@needs_vectors
@pytest.mark.parametrize("prefix", ["vardct444"])
def test_lf_smoothing_matches_reference(oracle, prefix):
    if not _have(prefix + "_lfs_input"):
        pytest.skip("this file does not smooth its LF")
    lf_in, want = read_vec(prefix + "_lfs_input"), read_vec(prefix + "_lfs_output")
    h, w = lf_in.shape[1:]
    p = _params_from_header(oracle, prefix, w * 8, h * 8)
    # finalize_lf's factors (frame/mod.rs:360-369) as the oracle derives them == what the reference passed in
    fac = read_vec(prefix + "_lfs_factors")
    inv_quant_lf = np.float32(65536.0) / (np.float32(p.global_scale) * np.float32(p.quant_lf))
    assert np.array_equal(np.float32([np.float32(p.lf_quant_factors[c]) * inv_quant_lf for c in range(3)]), fac)
    got = oracle.adaptive_lf_smoothing(p, [lf_in[c] for c in range(3)])
    for c in range(3):
        assert np.array_equal(got[c].view(np.uint32), want[c].view(np.uint32)), c


@needs_vectors
def test_dequant_lf_matches_reference(oracle):
    q, want = read_vec("vardct444_dqlf_input_yxb"), read_vec("vardct444_dqlf_output_xyb")
    h, w = q.shape[1:]
    p = _params_from_header(oracle, "vardct444", w * 8, h * 8)
    par = read_vec("vardct444_dqlf_params")
    mul = 1.0  # extra_precision scales fac_*: recover it from the dumped factor
    inv_quant_lf = np.float32(65536.0) / (np.float32(p.global_scale) * np.float32(p.quant_lf))
    base = np.float32(p.lf_quant_factors[1]) * inv_quant_lf
    for e in range(4):
        if np.float32(base * np.float32(1.0 / (1 << e))) == par[1]:
            mul = 1.0 / (1 << e)
    got = oracle.dequant_lf(p, q[0], q[1], q[2], mul=mul)
    for c in range(3):
        assert np.array_equal(got[c].view(np.uint32), want[c].view(np.uint32)), c
    for c in range(3):  # the sub-sampled branch (no chroma-from-luma), channel by channel
        name = f"jpeg420_dqlfsub{c}"
        if _have(name + "_input"):
            qc, wc = read_vec(name + "_input"), read_vec(name + "_output")
            fac = read_vec(name + "_fac")[0]
            assert np.array_equal((qc.astype(np.float32) * fac).view(np.uint32), wc.view(np.uint32)), c


@needs_vectors
def test_sigma_map_matches_reference(oracle):
    rq, em = read_vec("vardct444_sigma_raw_quant"), read_vec("vardct444_sigma_epf_map")
    want = read_vec("vardct444_sigma_inv_sigma")
    h, w = rq.shape
    p = _params_from_header(oracle, "vardct444", w * 8, h * 8)
    got = oracle.sigma_map(p, rq, em.astype(np.uint8))
    tm = read_vec("vardct444_sigma_transform_map")
    assert (tm > 0).any()
    assert np.array_equal(got.view(np.uint32), want[:, :w].view(np.uint32))


@needs_vectors
def test_whole_group_matches_reference(oracle, oracle_unfused):
    """group 0 of a real VarDCT frame through decode_vardct_group's dequant / CfL / LLF / IDCT branch
    (frame/group.rs:579-611): the reference's own coefficients, maps, LF and dequant tables in, its pixels out"""
    pre = "vardct444"
    co = read_vec(pre + "_group_coeffs_xyb")
    tm, rq = read_vec(pre + "_group_transform_map"), read_vec(pre + "_group_raw_quant")
    yx, yb = read_vec(pre + "_group_ytox"), read_vec(pre + "_group_ytob")
    lf = read_vec(pre + "_group_lf_xyb")
    bh, bw = tm.shape
    from jxl_rs_amd import synth
    tables = []
    for t in range(17):  # table index -> some transform type that uses it
        ty = synth.TABLE_FOR_TYPE.index(t)
        tables.append(read_vec(f"{pre}_group_table_type{ty}")[: 3 * 64 * synth.REQ_X[t] * synth.REQ_Y[t]])
    want = [read_vec(f"{pre}_group_pixels_c{c}") for c in range(3)]
    outs = []
    for o in (oracle, oracle_unfused):
        p = _params_from_header(o, pre, bw * 8, bh * 8)
        planes = [np.zeros((bh * 8, bw * 8), np.float32) for _ in range(3)]
        o.decode_group(p, 0, co, tm.astype(np.uint8), rq, yx.astype(np.int8), yb.astype(np.int8), [lf[c] for c in range(3)],
                       tables, planes)
        outs.append(planes)
    for c in range(3):
        _equal_to_either_build(outs[0][c], outs[1][c], np.ascontiguousarray(want[c][: bh * 8, : bw * 8]), f"group 0 channel {c}")


@needs_vectors
def test_palette_matches_reference(oracle):
    idx, pal = read_vec("palette_index"), read_vec("palette_table")
    nc, nd, depth = [int(v) for v in read_vec("palette_meta")]
    want = read_vec("palette_plain")
    assert np.array_equal(oracle.palette(idx, pal, nc + nd, pal.shape[0], depth), want)
    for pred in range(14):
        want = read_vec(f"palette_delta_pred{pred}")
        if pred == 6:
            got = oracle.palette_delta_wp(idx, pal, nc, nd, pal.shape[0], depth, (16, 10, 7, 7, 7, 0, 0, 13, 12, 12, 12))
        else:
            got = oracle.palette_delta(idx, pal, nc, nd, depth, pred)
        assert np.array_equal(got, want), pred
