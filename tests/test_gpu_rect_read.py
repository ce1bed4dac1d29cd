"""jxlh_frame_read_planes_rect: the finished planes group by group -- the unit the reference's pipeline moves
(RenderPipeline::get_buffer / set_buffer_for_group, render/mod.rs:124-137; group buffers rounded up to 16 pixels,
render/internal.rs:144-167).  Every group of a ragged frame is read through the C ABI and the groups are put together
again: the result must be the whole-frame read (and the oracle's frame) bit for bit."""
import ctypes as C

import numpy as np
import pytest

from helpers import bit_equal, diff_report, run_gpu_frame, run_oracle_frame

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from jxl_rs_amd import Context
    c = Context(0, n_slots=1)
    yield c
    c.close()


@pytest.mark.parametrize("size,over", [((600, 520), {}), ((777, 300), {}), ((250, 130), {}),
                                       ((300, 270), dict(upsampling=2))])
def test_groups_read_one_by_one_reassemble_the_frame(ctx, oracle, size, over):
    from jxl_rs_amd import synth
    w, h = size
    wl = synth.make_vardct(w, h, mix=synth.MIX_D1, seed=w * 3 + h, epf_iters=2)
    whole, _ = run_gpu_frame(ctx, wl, **over)
    W, H = ctx.out_size
    if not over:
        want, _ = run_oracle_frame(oracle, wl)
        for c in range(3):
            assert bit_equal(whole[c], want[c]), f"plane {c}: {diff_report(whole[c], want[c])}"
    xg, yg = (W + 255) // 256, (H + 255) // 256
    got = [np.full((H, W), np.nan, dtype=np.float32) for _ in range(3)]
    for g in range(xg * yg):
        bufs, (gw, gh) = ctx.read_group_planes(g)
        bw, bh = bufs[0].shape[1], bufs[0].shape[0]
        assert bw % 16 == 0 and bh % 16 == 0 and bw >= gw and bh >= gh
        x0, y0 = (g % xg) * 256, (g // xg) * 256
        for c in range(3):
            got[c][y0:y0 + gh, x0:x0 + gw] = bufs[c][:gh, :gw]
            # what lies beyond the frame's edge is not written
            assert not bufs[c][gh:, :].any() and not bufs[c][:, gw:].any()
    for c in range(3):
        assert bit_equal(got[c], whole[c]), f"plane {c}: {diff_report(got[c], whole[c])}"


def test_rect_read_strided_destination_and_errors(ctx):
    from jxl_rs_amd import synth
    from jxl_rs_amd.lib import JxlHipError, Plane, ERR_INVALID_ARGUMENT
    wl = synth.make_vardct(300, 280, mix=synth.MIX_D1, seed=9, epf_iters=1)
    whole, _ = run_gpu_frame(ctx, wl)
    # an odd rect into the middle of wider host buffers (pitch != width)
    out = [np.full((50, 96), -7.0, dtype=np.float32) for _ in range(3)]
    views = [o[3:, 8:] for o in out]
    ctx.read_planes_rect(131, 77, 40, 33, views)
    for c in range(3):
        assert bit_equal(out[c][3:36, 8:48], whole[c][77:110, 131:171])
        ref = np.full((50, 96), -7.0, dtype=np.float32)
        ref[3:36, 8:48] = whole[c][77:110, 131:171]
        assert bit_equal(out[c], ref)          # nothing outside the rect was touched
    # a rect that starts outside the result, an empty one, a destination that is too small
    for args in ((300, 0, 16, 16), (0, 280, 16, 16), (0, 0, 0, 16)):
        with pytest.raises(JxlHipError) as e:
            ctx.read_planes_rect(*args)
        assert e.value.status == ERR_INVALID_ARGUMENT
    small = [np.zeros((8, 8), dtype=np.float32) for _ in range(3)]
    with pytest.raises(JxlHipError) as e:
        ctx.read_planes_rect(0, 0, 16, 16, small)
    assert e.value.status == ERR_INVALID_ARGUMENT
    # the asynchronous form: valid after the next sync
    dst = [np.zeros((16, 16), dtype=np.float32) for _ in range(3)]
    planes = (Plane * 3)(*[Plane(o.ctypes.data, 64, 16, 64) for o in dst])
    ctx._chk(ctx.L.jxlh_frame_read_planes_rect_async(ctx._ctx, 32, 48, 16, 16, planes), "read_planes_rect_async")
    ctx.sync()
    for c in range(3):
        assert bit_equal(dst[c], whole[c][48:64, 32:48])
