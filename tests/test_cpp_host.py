"""The drop-in boundary from compiled code: tests/cpp/frame_parity.cc (C++17, g++) drives the C++ host side
include/jxl_hip.hpp over the C ABI and checks the frame against the oracle library, no Python in the data path;
tests/cpp/pipeline_builder.cc does the same through the mirror of the reference's stage traits and
RenderPipelineBuilder (include/jxl_hip_pipeline.hpp), whose lowering / rejection logic is host-only and runs
without a GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


def _build(tmp_path, name="frame_parity"):
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s"], check=True)
    exe = os.path.join(str(tmp_path), name)
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include",
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "oracle"),
           "-I", CPP, os.path.join(CPP, name + ".cc"),
           "-o", exe, "-L", os.path.join(ROOT, "jxl_rs_amd"), "-ljxl_hip", "-L", os.path.join(ROOT, "oracle"),
           "-l:libjxlo_fused.so", "-Wl,-rpath," + os.path.join(ROOT, "jxl_rs_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
           "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-lpthread", "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def test_cpp_host_header_compiles_and_links(tmp_path):
    """no GPU needed: the header and the test program build against the two shared libraries"""
    assert os.path.exists(_build(tmp_path))


@pytest.mark.gpu
@pytest.mark.parametrize("args", [("300", "270", "2"), ("64", "64", "0"), ("515", "133", "3")])
def test_cpp_host_frame_parity(tmp_path, args):
    exe = _build(tmp_path)
    r = subprocess.run([exe, *args], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    assert "0 differing rows" in r.stdout and "error path ok" in r.stdout


def test_pipeline_builder_lowering_and_rejections(tmp_path):
    """host logic of the RenderPipelineBuilder mirror: the reference's stage lists lower onto the frame parameters the
    device path takes (accumulated border 4 for Gaborish + EPF1 + EPF2, render/mod.rs:28-36), lists outside the path
    come back as JXLH_ERR_UNSUPPORTED naming the stage.  No GPU involved."""
    exe = _build(tmp_path, "pipeline_builder")
    r = subprocess.run([exe, "host"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "host checks: ok" in r.stdout, (r.stdout + r.stderr)[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("args", [("300", "270", "2"), ("515", "133", "3"), ("260", "520", "0")])
def test_pipeline_builder_frame_two_passes(tmp_path, args):
    """a frame assembled as Frame::build_render_pipeline assembles it, decoded through GpuRenderPipeline in two passes
    (set_buffer_for_group with complete = false, then complete + mark_group_to_rerender), bit-exact against the oracle"""
    exe = _build(tmp_path, "pipeline_builder")
    r = subprocess.run([exe, "gpu", *args], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    assert "two passes: 0 differing rows" in r.stdout and "RGBA8 tail: 0 differing rows" in r.stdout \
        and "error path ok" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("args", [("256", "200"), ("131", "77")])
def test_pipeline_builder_modular_lists(tmp_path, args):
    """Modular stage lists through the builder mirror: the I32 -> U8 special case (render/builder.rs:152-170) byte-exact
    against the oracle, the f32 conversion bit-exact, the f32 + Gaborish + EPF1 route equal to the ABI calls made by hand"""
    exe = _build(tmp_path, "pipeline_builder")
    r = subprocess.run([exe, "modular", *args], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    assert "I32->U8 ok" in r.stdout and "to f32 ok" in r.stdout and "f32 + filters ok" in r.stdout
