"""Committed golden fixtures (tests/golden/frame_digests.json, written by tests/gen_frame_digests.py): SHA-256 digests
of the oracle's output for seeded whole frames and Modular chains.  The CPU test holds the oracle to them, the GPU
tests hold the device path to the SAME digests through the C ABI -- neither side can drift, alone or together."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen_frame_digests as gfd  # noqa: E402

with open(gfd.OUT) as f:
    GOLDEN = json.load(f)


def test_fixture_covers_the_generator_case_list():
    assert sorted(GOLDEN["vardct"]) == sorted(c[0] for c in gfd.VARDCT_CASES)
    assert sorted(GOLDEN["modular"]) == sorted(c[0] for c in gfd.MODULAR_CASES)


@pytest.mark.parametrize("case", gfd.VARDCT_CASES, ids=lambda c: c[0])
def test_oracle_reproduces_golden_frame(oracle, case):
    got = gfd.vardct_case(oracle, *case)[3]
    want = GOLDEN["vardct"][case[0]]
    assert got["inputs"] == want["inputs"], "the synthetic workload generator changed: regenerate the fixture deliberately"
    assert got == want


@pytest.mark.parametrize("case", gfd.MODULAR_CASES, ids=lambda c: c[0])
def test_oracle_reproduces_golden_chain(oracle, case):
    got = gfd.modular_case(oracle, *case)[2]
    assert got == GOLDEN["modular"][case[0]]


@pytest.mark.gpu
@pytest.mark.parametrize("case", gfd.VARDCT_CASES, ids=lambda c: c[0])
@pytest.mark.parametrize("sparse", [False, True], ids=["dense", "sparse"])
def test_device_frame_matches_golden_digest(case, sparse):
    import helpers
    from jxl_rs_amd import Context, synth
    name, w, h, mix, seed, opts = case
    want = GOLDEN["vardct"][name]
    wl = synth.make_vardct(w, h, mix=getattr(synth, mix), seed=seed, **opts)
    ctx = Context(0, n_slots=1)
    try:
        p = helpers.gpu_params_from(ctx, wl)
        ctx.frame_begin(p)
        ctx.set_dequant_tables(wl.tables)
        ctx.set_lf_quantized(*wl.lf_q)
        ctx.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
        for g in range(wl.coeffs.shape[0]):
            if sparse:
                ctx.submit_group_sparse(g, *synth.to_sparse(wl.coeffs[g]))
            else:
                ctx.submit_group(g, wl.coeffs[g])
        ctx.slot_wait(0)
        ctx.frame_run()
        ctx.sync()
        planes = [np.ascontiguousarray(pl, dtype=np.float32) for pl in ctx.read_planes()]
        assert [gfd.sha(pl) for pl in planes] == want["planes"]
        if not helpers.is_subsampled(wl):
            lf = [np.ascontiguousarray(l, dtype=np.float32) for l in ctx.read_lf()]
            assert [gfd.sha(l) for l in lf] == want["lf"]
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", gfd.MODULAR_CASES, ids=lambda c: c[0])
def test_device_chain_matches_golden_digest(case):
    from jxl_rs_amd import Context
    from jxl_rs_amd.modular import ModularChain
    name, w, h, seed, rct = case
    ctx = Context(0, n_slots=1)
    try:
        ch = ModularChain(ctx, w, h, seed=seed, rct=rct)
        ch.run_chain()
        got = [np.ascontiguousarray(pl) for pl in ch.result()]
        ch.free()
        assert [gfd.sha(pl) for pl in got] == GOLDEN["modular"][name]["planes"]
    finally:
        ctx.close()
