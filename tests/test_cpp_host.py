"""The drop-in boundary from compiled code: tests/cpp/frame_parity.cc (C++17, g++) drives the C++ host side
include/jxl_hip.hpp over the C ABI and checks the frame against the oracle library, no Python in the data path."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "frame_parity.cc")


def _build(tmp_path):
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s"], check=True)
    exe = os.path.join(str(tmp_path), "frame_parity")
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "oracle"), SRC,
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
