"""The Rust side of the boundary (bindings/rust/): the raw binding is generated from include/jxl_hip.h, so it can
never again lag behind the header (round 1 shipped a pasted #[repr(C)] block that stopped at ABI v2's last field).
Checked here without a Rust toolchain: the committed file is what the generator produces now; every struct has the
header's fields in the header's order; their sizes and offsets, computed with Rust's #[repr(C)] rules, equal what
the C compiler lays out; every exported symbol has a binding."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

RUST_SIZE = {"u8": 1, "i8": 1, "u16": 2, "i16": 2, "u32": 4, "i32": 4, "u64": 8, "i64": 8, "usize": 8, "f32": 4, "f64": 8}


def test_committed_binding_is_current():
    import gen_rust_binding as g
    assert open(g.OUT).read() == g.generate(), "run tools/gen_rust_binding.py"


def rust_structs(text):
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^)]*\)\]\s*)?pub struct (\w+) \{(.*?)\n\}", text, flags=re.S):
        fields = re.findall(r"pub (\w+): ([^,\n]+),", m.group(2))
        out[m.group(1)] = fields
    return out


def layout(fields, structs):
    """#[repr(C)] layout: (size, align, [(name, offset)])"""
    off, align, offs = 0, 1, []
    for name, ty in fields:
        m = re.match(r"\[(.+); (\d+)\]", ty)
        n = int(m.group(2)) if m else 1
        base = m.group(1) if m else ty
        if base.startswith("*"):
            sz = al = 8
        elif base in RUST_SIZE:
            sz = al = RUST_SIZE[base]
        else:
            sz, al, _ = layout(structs[base], structs)
        off = (off + al - 1) // al * al
        offs.append((name, off))
        off += sz * n
        align = max(align, al)
    return (off + align - 1) // align * align, align, offs


def test_struct_layouts_match_the_c_compiler(tmp_path):
    import gen_rust_binding as g
    structs = rust_structs(open(g.OUT).read())
    names = [n for n in structs if structs[n] and structs[n][0][0] != "_private"]
    assert {"jxlh_frame_params", "jxlh_output_desc", "jxlh_xyb_params", "jxlh_plane", "jxlh_coeff16", "jxlh_coeff32"} <= set(names)
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "jxl_hip.h"', "int main(void) {"]
    for n in names:
        lines.append(f'  printf("{n} size %zu\\n", sizeof({n}));')
        for f, _ in structs[n]:
            lines.append(f'  printf("{n} {f} %zu\\n", offsetof({n}, {f}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = {}
    for ln in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines():
        n, f, v = ln.split()
        got[(n, f)] = int(v)
    for n in names:
        size, _, offs = layout(structs[n], structs)
        assert got[(n, "size")] == size, n
        for f, o in offs:
            assert got[(n, f)] == o, (n, f)
    # the struct the round-1 document truncated: its last fields are there, in order
    fp = [f for f, _ in structs["jxlh_frame_params"]]
    assert fp[-10:] == ["hshift", "vshift", "epf_sigma_for_modular", "upsampling", "xsize_upsampled", "ysize_upsampled",
                        "noise", "noise_lut", "visible_frame_index", "nonvisible_frame_index"]


def test_every_exported_symbol_is_bound_and_the_python_struct_agrees():
    import ctypes as C
    import gen_rust_binding as g
    from jxl_rs_amd import lib
    text = open(g.OUT).read()
    bound = set(re.findall(r"pub fn (jxlh_\w+)\(", text))
    assert bound == set(lib.ABI_SYMBOLS)
    structs = rust_structs(text)
    assert [f for f, _ in structs["jxlh_frame_params"]] == [f for f, _ in lib.FrameParams._fields_]
    assert layout(structs["jxlh_frame_params"], structs)[0] == C.sizeof(lib.FrameParams)


def test_integration_doc_points_at_the_generated_binding():
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "bindings/rust/jxl_hip_sys/src/lib.rs" in doc and "tools/gen_rust_binding.py" in doc
    assert "todo!()" not in doc
    # no hand-maintained copy of the parameter struct in the document any more
    assert "pub struct jxlh_frame_params" not in doc
