"""The stream-ordering contract of device-pointer arguments (include/jxl_hip.h "STREAM ORDERING OF DEVICE POINTERS";
VERDICT r05 'boundary'): the context's streams are non-blocking, so a caller that fills or clears a device plane on the
NULL stream must order that in front of the call -- jxlh_ctx_wait_stream / _wait_event.  Round 5's soak hit the race as a
0.2 % flake (an output plane zeroed AFTER the kernel had written it) and fixed it in the Python harness only; this is the
regression test through the ABI: the NULL stream is kept busy for milliseconds, the output planes are cleared behind
that, and the library is called right away."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _chain(ctx, seed):
    from jxl_rs_amd.modular import ModularChain
    return ModularChain(ctx, 2048, 1536, seed=seed)


@pytest.mark.parametrize("how", ["wait_stream", "wait_event"])
def test_null_stream_memset_then_device_pointer_call(how):
    import jxl_rs_amd
    from jxl_rs_amd.lib import DeviceArray
    from helpers import modular_chain_oracle
    from oracle.oracle import Oracle
    ctx = jxl_rs_amd.Context(0, 1)
    hip = DeviceArray.hip()
    hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
    hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
    hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
    ch = _chain(ctx, 5)
    want = modular_chain_oracle(ch, Oracle(fused=True))
    busy = DeviceArray(nbytes=1 << 30, device=0)
    ev = C.c_void_p()
    assert hip.hipEventCreate(C.byref(ev)) == 0
    for rep in range(8):
        # a long queue on the NULL stream (8 x 1 GiB of fills: several milliseconds), the output planes' clear behind it
        for _ in range(8):
            assert hip.hipMemsetAsync(busy.ptr, rep, busy.nbytes, None) == 0
        for d in ch.d_out:
            assert hip.hipMemsetAsync(d.ptr, 0xEE, d.nbytes, None) == 0
        if how == "wait_stream":
            ctx.wait_stream(None)
        else:
            assert hip.hipEventRecord(ev, None) == 0
            ctx.wait_event(ev)
        ch.run_chain()           # jxlh_unsqueeze_chain on the context's own (non-blocking) stream, device pointers only
        ctx.sync()
        got = ch.result()
        for c in range(3):
            assert np.array_equal(got[c], want[c]), f"rep {rep}, plane {c}: the clear landed behind the library's stores"
    busy.free()
    ch.free()
    ctx.close()


def test_chain_with_separate_rct_applies_the_rct(monkeypatch):
    """with JXLH_SEPARATE_RCT=1 (the two-pass route planes of 2^31 samples need) the dataflow run used to
    swallow the last level and the chain returned without its RCT"""
    import jxl_rs_amd
    from helpers import modular_chain_oracle
    from oracle.oracle import Oracle
    ctx = jxl_rs_amd.Context(0, 1)
    ch = _chain(ctx, 9)
    assert ch.rct is not None
    want = modular_chain_oracle(ch, Oracle(fused=True))
    for flow in ("1", "0"):
        monkeypatch.setenv("JXLH_SEPARATE_RCT", "1")
        monkeypatch.setenv("JXLH_CHAIN_FLOW", flow)
        ch.run_chain()
        ctx.sync()
        got = ch.result()
        for c in range(3):
            assert np.array_equal(got[c], want[c]), f"JXLH_CHAIN_FLOW={flow}, plane {c}"
    ch.free()
    ctx.close()
