import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (they are git-ignored): build what is missing, exactly like
    __graft_entry__.build() -- hipcc cross-compiles gfx950 without a GPU."""
    import subprocess
    if not os.path.exists(os.path.join(ROOT, "jxl_rs_amd", "libjxl_hip.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "jxl_rs_amd", "csrc"), "-j8", "-s"], check=True)
    if not all(os.path.exists(os.path.join(ROOT, "oracle", n)) for n in ("libjxlo_fused.so", "libjxlo_unfused.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s"], check=True)


@pytest.fixture(scope="session")
def kat():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_kat.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session", params=["fused", "unfused"])
def oracle_any(request):
    from oracle.oracle import Oracle
    return Oracle(fused=(request.param == "fused"))


@pytest.fixture(scope="session")
def oracle():
    """The fused build (models the reference's AVX2 back-end == GPU v_fma_f32 behaviour)."""
    from oracle.oracle import Oracle
    return Oracle(fused=True)


@pytest.fixture(scope="session")
def oracle_unfused():
    from oracle.oracle import Oracle
    return Oracle(fused=False)
