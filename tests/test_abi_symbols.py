"""CPU-side checks of the drop-in boundary: the C-ABI library loads (no GPU needed for that)
and exports every symbol include/jxl_hip.h declares; argument validation paths that do not
touch the device behave as documented."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols(name="jxl_hip.h"):
    src = open(os.path.join(ROOT, "include", name)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(jxlh_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from jxl_rs_amd import lib
    L = lib.load()
    declared = header_symbols()
    assert declared, "header parse failed"
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/jxl_hip.h but not exported"
    assert sorted(lib.ABI_SYMBOLS) == declared


def test_dev_header_is_separate_from_the_product_abi():
    """timers / probes / self-tests live in include/jxl_hip_dev.h: exported by the library, declared nowhere in the
    product header (so the generated Rust -sys crate does not bind them)"""
    from jxl_rs_amd import lib
    L = lib.load()
    dev = [n for n in header_symbols("jxl_hip_dev.h")]
    assert sorted(lib.DEV_SYMBOLS) == dev and dev
    for name in dev:
        assert hasattr(L, name), name
    assert not set(dev) & set(header_symbols())


def test_abi_version_and_tables():
    from jxl_rs_amd import lib, synth
    L = lib.load()
    assert L.jxlh_abi_version() == 6
    for t in range(27):
        assert L.jxlh_covered_blocks_x(t) == synth.COVERED_X[t]
        assert L.jxlh_covered_blocks_y(t) == synth.COVERED_Y[t]
        assert L.jxlh_quant_table_for_type(t) == synth.TABLE_FOR_TYPE[t]
    assert L.jxlh_covered_blocks_x(27) == -1
    assert sum(L.jxlh_quant_table_size(q) for q in range(17)) == 2056 * 64  # quant_weights.rs:1135


def test_default_params_match_reference_header_defaults():
    from jxl_rs_amd import lib
    L = lib.load()
    p = lib.FrameParams()
    assert L.jxlh_default_frame_params(C.byref(p), 1000, 600) == 0
    assert (p.xsize, p.ysize) == (1000, 600)
    assert p.gab == 1 and p.epf_iters == 2  # frame_header.rs:150-183
    assert abs(p.gab_w1[0] - 0.115169525) < 1e-8 and abs(p.gab_w2[2] - 0.061248592) < 1e-8
    assert list(p.epf_channel_scale) == [40.0, 5.0, 3.5]
    assert p.x_qm_scale == 3 and p.b_qm_scale == 2 and p.color_factor == 84
    assert L.jxlh_default_frame_params(None, 1, 1) == lib.ERR_INVALID_ARGUMENT


def test_null_and_bad_arguments_do_not_crash():
    from jxl_rs_amd import lib
    L = lib.load()
    assert L.jxlh_ctx_create(0, 0, None) == lib.ERR_INVALID_ARGUMENT
    assert L.jxlh_frame_run(None, 0, 1) == lib.ERR_INVALID_ARGUMENT
    assert L.jxlh_ctx_sync(None) == lib.ERR_INVALID_ARGUMENT
    assert L.jxlh_ctx_tune_placement(None, 4, None, 0, None, None) == lib.ERR_INVALID_ARGUMENT
    assert L.jxlh_probe_placement(None, None, None) == lib.ERR_INVALID_ARGUMENT
    assert L.jxlh_host_pack_slots_many(None, None, 0, 0, None, 0, None, None, None, 0, None, None) == lib.ERR_INVALID_ARGUMENT
    assert L.jxlh_status_string(lib.ERR_INVALID_TRANSFORM).decode() == "invalid VarDCT transform id"
    L.jxlh_ctx_destroy(None)


def test_product_package_does_not_import_the_oracle():
    # the oracle is test infrastructure: nothing under jxl_rs_amd/ may reference it
    pkg = os.path.join(ROOT, "jxl_rs_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp", ".inc")) or fn == "Makefile":
                txt = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "import oracle" not in txt and "from oracle" not in txt and "libjxlo" not in txt, fn
                assert "jxlo_" not in txt, fn
                # ... nor take an oracle object and call into it (round 3: ModularChain's oracle comparisons moved to
                # tests/helpers.py)
                assert "oracle." not in txt and "(oracle" not in txt and ", oracle" not in txt, fn


def test_only_tests_smoke_and_the_cpu_baseline_touch_the_oracle():
    """tools/, include/ and bindings/ never import, link or run anything under oracle/; bench.py and
    __graft_entry__.py do so only in their declared places (the cpu_baseline legs, smoke())."""
    import re
    for sub in ("tools", "include", "bindings"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, sub)):
            for fn in files:
                txt = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert not re.search(r"import oracle|from oracle|libjxlo|jxlo_", txt), os.path.join(sub, fn)
    bench = open(os.path.join(ROOT, "bench.py")).read()
    uses = [m.start() for m in re.finditer(r"from oracle\.oracle import Oracle", bench)]
    assert len(uses) == 2, "bench.py: one import per CPU-baseline leg (VarDCT, Modular)"
    for pos in uses:
        assert "cpu" in bench[max(0, pos - 1500):pos].lower(), "oracle import outside a cpu_baseline leg"
    entry = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert entry.count("from oracle.oracle import Oracle") == 1 and entry.index("from oracle.oracle import Oracle") > entry.index("def smoke")


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """include/jxl_hip.h is a C ABI: it compiles as strict C99 and a C program links against
    libjxl_hip.so using nothing but the header (no GPU needed for the calls made here)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    src = tmp_path / "abi_check.c"
    src.write_text(
        '#include <stdio.h>\n#include "jxl_hip.h"\n'
        "int main(void) {\n"
        "  jxlh_frame_params p;\n"
        "  if (jxlh_default_frame_params(&p, 300, 200) != JXLH_OK) return 1;\n"
        "  if (p.xsize != 300 || p.ysize != 200 || p.epf_iters != 2) return 2;\n"
        "  if (jxlh_covered_blocks_x(5) != 4 || jxlh_covered_blocks_y(26) != 16) return 3;\n"
        "  if (jxlh_frame_run(NULL, 0, 1) != JXLH_ERR_INVALID_ARGUMENT) return 4;\n"
        "  if (sizeof(jxlh_coeff16) != 4 || sizeof(jxlh_coeff32) != 8 || sizeof(jxlh_xyb_params) != 64) return 5;\n"
        '  printf("abi %u\\n", jxlh_abi_version());\n'
        "  return 0;\n}\n")
    exe = tmp_path / "abi_check"
    libdir = os.path.join(ROOT, "jxl_rs_amd")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                    str(src), "-o", str(exe), "-L", libdir, "-ljxl_hip", "-Wl,-rpath," + libdir], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    assert out.startswith("abi ")


def test_frame_begin_validates_before_touching_the_device():
    """Chroma-subsampled frames (hshift / vshift in {0, 1}) are part of the device path (K1e); what frame_begin
    rejects without a context is the call itself.  The shift-range check needs a context: tests/test_gpu_parity.py
    ::test_subsampled_frame_rejects_large_varblocks."""
    from jxl_rs_amd import lib
    L = lib.load()
    p = lib.FrameParams()
    L.jxlh_default_frame_params(C.byref(p), 64, 64)
    assert list(p.hshift) == [0, 0, 0] and list(p.vshift) == [0, 0, 0]
    p.hshift[0] = 1
    p.vshift[2] = 1
    assert L.jxlh_frame_begin(None, C.byref(p)) == lib.ERR_INVALID_ARGUMENT


def test_generated_constant_tables_have_one_content():
    """oracle/tools/extract_reference_constants.py writes every .inc twice (oracle/ for the checker, csrc/ for the
    device library): the two copies must never drift apart, or parity tests compare different constants."""
    a, b = os.path.join(ROOT, "oracle"), os.path.join(ROOT, "jxl_rs_amd", "csrc")
    names = sorted(f for f in os.listdir(a) if f.endswith(".inc"))
    assert names, "no generated tables found"
    for f in names:
        assert open(os.path.join(a, f), "rb").read() == open(os.path.join(b, f), "rb").read(), f
    # and the device library's Makefile rebuilds when one of them changes
    mk = open(os.path.join(b, "Makefile")).read()
    for f in sorted(x for x in os.listdir(b) if x.endswith(".inc")):
        assert f in mk, f"{f} missing from HDRS in csrc/Makefile"
