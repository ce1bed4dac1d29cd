"""jxlh_ctx_tune_placement (include/jxl_hip.h "PLACEMENT OF A CONTEXT'S BUFFERS"): the first allocation of a context's
large buffers as a pick among candidate sets rated on the device.  It must not change a single bit of the result, must
report what it saw, and must leave a context that already holds its buffers alone."""
import numpy as np
import pytest

from helpers import bit_equal, run_gpu_frame

pytestmark = pytest.mark.gpu


def test_tuned_context_gives_the_same_bits_and_reports_its_pick():
    import jxl_rs_amd
    from jxl_rs_amd import synth
    wl = synth.make_vardct(1024, 768, mix=synth.MIX_ALL, seed=31, epf_iters=2)
    plain = jxl_rs_amd.Context(0, 1)
    assert plain.tune_placement() == ([], -1)          # nothing picked, default is the plain allocation
    want, want_lf = run_gpu_frame(plain, wl)
    tuned = jxl_rs_amd.Context(0, 1)
    tuned.tune_placement(4)
    assert tuned.tune_placement() == ([], -1)          # the pick happens inside the first frame_begin
    got, got_lf = run_gpu_frame(tuned, wl)
    for c in range(3):
        assert bit_equal(got[c], want[c]) and bit_equal(got_lf[c], want_lf[c])
    ratings, pick = tuned.tune_placement()
    assert 4 <= len(ratings) <= 8 and 0 <= pick < len(ratings)   # (up to twice the trials while none stands out)
    assert all(a > 0 and b > 0 for a, b in ratings)
    assert 4 * ratings[pick][0] + ratings[pick][1] <= min(4 * a + b for a, b in ratings) * 1.000001
    # a second frame on the same context keeps its buffers: no new pick, same bits
    got2, _ = run_gpu_frame(tuned, wl)
    assert tuned.tune_placement()[1] == pick
    for c in range(3):
        assert bit_equal(got2[c], want[c])
    # the slot-bucketed form on a tuned context (its coefficient buffer is only touched by routed groups)
    from jxl_rs_amd import lib as jl
    ng = wl.coeffs.shape[0]
    t2 = jxl_rs_amd.Context(0, 1)
    t2.tune_placement(3)
    t2.frame_begin(synth.apply_opts(t2.default_params(1024, 768), wl))
    t2.set_dequant_tables(wl.tables); t2.set_lf_quantized(*wl.lf_q)
    t2.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
    parts = [jl.host_pack_slots(wl.coeffs[g], group_id=g) for g in range(ng)]
    t2.submit_groups_slots(np.arange(ng, dtype=np.uint32), np.concatenate([p[0] for p in parts]),
                           np.concatenate([p[1].reshape(-1) for p in parts]), np.concatenate([p[2] for p in parts]), None)
    t2.slot_wait(0)
    t2.frame_run(); t2.sync()
    got3 = t2.read_planes()
    for c in range(3):
        assert bit_equal(got3[c], want[c])


def test_placement_probe_needs_a_frame_and_argument_checks():
    import jxl_rs_amd
    from jxl_rs_amd import lib as jl
    c = jxl_rs_amd.Context(0, 1)
    with pytest.raises(jl.JxlHipError):
        c.probe_placement()                             # no frame yet
    with pytest.raises(jl.JxlHipError):
        c.tune_placement(65)
    with pytest.raises(jl.JxlHipError):
        c.tune_placement(-1)
