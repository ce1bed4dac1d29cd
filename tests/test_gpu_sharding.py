"""Multi-GPU band sharding through the C ABI on ONE GPU: the in-process transport (jxlh_comm_init_local) runs the
same band logic as the RCCL transport -- transforms on the own band only, halo EXCHANGE of the edge block rows,
filters, all-gather -- with several contexts standing in for the ranks.  Every rank's gathered frame must equal the
oracle's whole frame bit for bit.  The RCCL transport itself is exercised with a one-rank communicator (what a
single-GPU box can run); its N > 1 form runs in bench.py --gpus N."""
import numpy as np
import pytest

from helpers import bit_equal, diff_report, gpu_params_from, run_oracle_frame
import helpers

pytestmark = pytest.mark.gpu


def upload_band(ctx, wl, groups, **over):
    """replicated small inputs + the coefficient groups in `groups` only"""
    p = gpu_params_from(ctx, wl, **over)
    ctx.frame_begin(p)
    ctx.set_dequant_tables(wl.tables)
    ctx.set_lf_quantized(*wl.lf_q)
    ctx.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
    poison = np.full(3 * 65536, 0x3FFF, dtype=np.int32)
    for g in range(wl.coeffs.shape[0]):
        ctx.submit_group(g, wl.coeffs[g] if g in groups else poison)
    ctx.slot_wait(0)


def band_groups(wl, rank, n, extra=0):
    from jxl_rs_amd.shard import band_for_rank
    r0, r1, _ = band_for_rank(wl.ygroups, rank, n)
    if r0 >= r1:
        return set()
    r0, r1 = max(0, r0 - extra), min(wl.ygroups, r1 + extra)
    return {gy * wl.xgroups + gx for gy in range(r0, r1) for gx in range(wl.xgroups)}


@pytest.mark.parametrize("n", [2, 3, 6])
@pytest.mark.parametrize("epf_iters,gab,flags", [(2, True, 0), (0, True, 0), (3, True, 0), (1, False, 0), (2, True, 1),
                                                 (3, True, 1), (0, False, 0)])
def test_local_sharded_frame_equals_whole(oracle, n, epf_iters, gab, flags):
    import jxl_rs_amd
    from jxl_rs_amd import lib, synth
    wl = synth.make_vardct(520, 1000, mix=synth.MIX_ALL, seed=31 + n, epf_iters=epf_iters, gab=gab)
    assert wl.ygroups == 4
    want, _ = run_oracle_frame(oracle, wl)
    ctxs = [jxl_rs_amd.Context(0, 1) for _ in range(n)]
    try:
        lib.comm_init_local(ctxs)
        for r, c in enumerate(ctxs):
            # the rank holds ONLY its band's coefficients (the other groups are poisoned): any recomputed halo
            # group row would show
            upload_band(c, wl, band_groups(wl, r, n), flags=flags)
            assert c.comm_band()[:2] == (r, n)
        for rep in range(2):  # a second run on the same contexts: events and buffers are reusable
            lib.frames_run_sharded_local(ctxs)
            lib.frames_allgather_local(ctxs)
            for r, c in enumerate(ctxs):
                c.sync()
                got = c.read_planes()
                for ch in range(3):
                    assert bit_equal(got[ch], want[ch]), f"rank {r} plane {ch} rep {rep}: {diff_report(got[ch], want[ch])}"
    finally:
        for c in ctxs:
            c.close()


def test_local_sharded_subsampled_frame(oracle):
    """chroma-subsampled frames keep the recomputed halo group row (their chroma upsampling reads across the band
    edge): the rank then needs the neighbouring group rows' coefficients too"""
    import jxl_rs_amd
    from jxl_rs_amd import lib, synth
    hs = vs = (1, 0, 1)
    wl = synth.make_vardct(300, 700, mix=synth.MIX_8X8, seed=12, epf_iters=1, hshift=hs, vshift=vs)
    want, _ = run_oracle_frame(oracle, wl)
    ctxs = [jxl_rs_amd.Context(0, 1) for _ in range(2)]
    try:
        lib.comm_init_local(ctxs)
        for r, c in enumerate(ctxs):
            upload_band(c, wl, band_groups(wl, r, 2, extra=1))
        lib.frames_run_sharded_local(ctxs)
        lib.frames_allgather_local(ctxs)
        for r, c in enumerate(ctxs):
            c.sync()
            got = c.read_planes()
            for ch in range(3):
                assert bit_equal(got[ch], want[ch]), f"rank {r} plane {ch}: {diff_report(got[ch], want[ch])}"
    finally:
        for c in ctxs:
            c.close()


def test_local_sharded_subsampled_frame_without_stages(oracle):
    """the JPEG-recompression case: 4:2:0, no Gaborish / EPF / noise / upsampling.  A single-GPU run defers the chroma
    upsampling until the planes are asked for; a sharded run must upsample before the all-gather, or every rank ends
    up with stale chroma outside its own band (both the plane read and the fused YCbCr -> RGB8 read)"""
    import jxl_rs_amd
    from jxl_rs_amd import lib, synth
    hs = vs = (1, 0, 1)
    wl = synth.make_vardct(300, 700, mix=synth.MIX_8X8, seed=13, epf_iters=0, gab=False, hshift=hs, vshift=vs)
    want, _ = run_oracle_frame(oracle, wl)
    want_rgb = oracle.ycbcr_to_rgb8(want, wl.xsize, wl.ysize, 3)
    ctxs = [jxl_rs_amd.Context(0, 1) for _ in range(2)]
    try:
        lib.comm_init_local(ctxs)
        for r, c in enumerate(ctxs):
            upload_band(c, wl, band_groups(wl, r, 2, extra=1))
        lib.frames_run_sharded_local(ctxs)
        lib.frames_allgather_local(ctxs)
        for r, c in enumerate(ctxs):
            c.sync()
            assert np.array_equal(c.read_ycbcr_rgb8(3), want_rgb), f"rank {r}: YCbCr RGB8 read differs"
            got = c.read_planes()
            for ch in range(3):
                assert bit_equal(got[ch], want[ch]), f"rank {r} plane {ch}: {diff_report(got[ch], want[ch])}"
    finally:
        for c in ctxs:
            c.close()


def test_sharded_upsampled_frame_is_declined():
    import jxl_rs_amd
    from jxl_rs_amd import lib, synth, JxlHipError
    wl = synth.make_vardct(300, 600, mix=synth.MIX_D1, seed=3, epf_iters=1)
    ctxs = [jxl_rs_amd.Context(0, 1) for _ in range(2)]
    try:
        lib.comm_init_local(ctxs)
        for c in ctxs:
            upload_band(c, wl, set(range(wl.coeffs.shape[0])), upsampling=2)
        with pytest.raises(JxlHipError) as e:
            lib.frames_run_sharded_local(ctxs)
        assert e.value.status == lib.ERR_UNSUPPORTED
    finally:
        for c in ctxs:
            c.close()


def test_rccl_transport_single_rank(oracle):
    """the RCCL path with a one-rank communicator: librccl is found, the communicator comes up on the context's
    device, the sharded run + in-place all-gather reproduce the whole frame"""
    import jxl_rs_amd
    from jxl_rs_amd import lib, synth, JxlHipError
    wl = synth.make_vardct(520, 600, mix=synth.MIX_D1, seed=8, epf_iters=2)
    want, _ = run_oracle_frame(oracle, wl)
    c = jxl_rs_amd.Context(0, 1)
    try:
        uid = lib.comm_unique_id()
        assert len(uid) == 128 and any(uid)
        c.comm_init(uid, 0, 1)
        with pytest.raises(JxlHipError):  # one communicator per context
            c.comm_init(uid, 0, 1)
        upload_band(c, wl, set(range(wl.coeffs.shape[0])))
        assert c.comm_band() == (0, 1, 0, wl.ygroups)
        c.frame_run_sharded()
        c.frame_allgather()
        c.sync()
        got = c.read_planes()
        for ch in range(3):
            assert bit_equal(got[ch], want[ch]), f"plane {ch}: {diff_report(got[ch], want[ch])}"
        # generic in-place gather of a device buffer (the join of band-sharded Modular work)
        addr, n_i32 = c.coeff_buffer()
        c.comm_allgather(addr, 4096)
        c.sync()
        c.comm_destroy()
        c.frame_run()  # back to a plain single-GPU context
        c.sync()
    finally:
        c.close()


@pytest.mark.parametrize("n", [2, 3])
def test_modular_rct_and_palette_band_sharded(oracle, n):
    """RCT / non-delta Palette across ranks: every rank transforms its share of the samples, the planes are joined
    with the in-place all-gather; every rank must end with the oracle's whole planes."""
    import ctypes as C
    import jxl_rs_amd
    from helpers import DeviceArray
    from jxl_rs_amd import lib
    from jxl_rs_amd.shard import sample_share
    rng = np.random.default_rng(90 + n)
    h, w = 777, 1001
    total = h * w
    planes = [rng.integers(-300, 300, size=total).astype(np.int32) for _ in range(3)]
    want = oracle.rct([p.reshape(h, w) for p in planes], 6, 3)
    pal = rng.integers(0, 256, size=(3, 200)).astype(np.int32)
    idx = rng.integers(-2, 260, size=total).astype(np.int32)  # incl. implicit entries on both sides
    want_pal = oracle.palette(idx.reshape(h, w), pal, 200, 3, 8)
    ctxs = [jxl_rs_amd.Context(0, 1) for _ in range(n)]
    bufs = []
    try:
        lib.comm_init_local(ctxs)
        _, _, count = sample_share(total, 0, n)
        dev = [[DeviceArray(nbytes=4 * n * count) for _ in range(3)] for _ in range(n)]
        out = [DeviceArray(nbytes=4 * 3 * n * count) for _ in range(n)]
        t_idx, t_pal = DeviceArray(idx), DeviceArray(pal)
        bufs = [b for row in dev for b in row] + out + [t_idx, t_pal]
        for r, c in enumerate(ctxs):
            i0, i1, _ = sample_share(total, r, n)
            for ch in range(3):  # a rank holds only its share of the inputs
                dev[r][ch].upload(planes[ch][i0:i1], 4 * i0)
            if i1 > i0:
                c._chk(c.L.jxlh_rct(c._ctx, C.c_void_p(dev[r][0].ptr + 4 * i0), C.c_void_p(dev[r][1].ptr + 4 * i0),
                                    C.c_void_p(dev[r][2].ptr + 4 * i0), i1 - i0, 6, 3), "rct")
                # palette of the share, expanded straight into the full-size planes (channel stride n * count)
                c._chk(c.L.jxlh_palette_strided(c._ctx, C.c_void_p(t_idx.ptr + 4 * i0), i1 - i0, C.c_void_p(t_pal.ptr),
                                                200, 200, 3, 8, C.c_void_p(out[r].ptr + 4 * i0), n * count),
                       "palette_strided")
        for ch in range(3):
            lib.comm_allgather_local(ctxs, [dev[r][ch].ptr for r in range(n)], 4 * count)
            lib.comm_allgather_local(ctxs, [out[r].ptr + 4 * ch * n * count for r in range(n)], 4 * count)
        for c in ctxs:
            c.sync()
        for r in range(n):
            for ch in range(3):
                got = dev[r][ch].download(np.int32, total).reshape(h, w)
                assert np.array_equal(got, want[ch]), f"rct rank {r} channel {ch}"
                gp = out[r].download(np.int32, total, 4 * ch * n * count).reshape(h, w)
                assert np.array_equal(gp, want_pal[ch]), f"palette rank {r} channel {ch}"
    finally:
        for b in bufs:
            b.free()
        for c in ctxs:
            c.close()


@pytest.mark.parametrize("n,size", [(2, (700, 500)), (3, (257, 331)), (2, (2048, 1024))])
def test_modular_config4_pipeline_sharded(oracle, n, size):
    """BASELINE configs[3] as ONE multi-rank pipeline (in-process ranks): replicated squeeze chain -> RCT on the own
    sample share -> palette on the own share -> in-place all-gathers; every rank must end with the oracle's whole
    planes (chain + RCT) and whole palette expansion"""
    import jxl_rs_amd
    from jxl_rs_amd import lib, synth
    from jxl_rs_amd.modular import ModularChain, run_pipeline_local
    w, h = size
    planes = synth.make_modular_planes(w, h, seed=40 + n)
    ctxs = [jxl_rs_amd.Context(0, 1) for _ in range(n)]
    chains = []
    try:
        lib.comm_init_local(ctxs)
        chains = [ModularChain(c, w, h, planes=planes, world=n) for c in ctxs]
        run_pipeline_local(chains, ctxs)
        want_planes, want_pal = helpers.modular_pipeline_oracle(chains[0], oracle)
        for r, ch in enumerate(chains):
            got_planes, got_pal = ch.pipeline_result()
            for c in range(3):
                assert np.array_equal(got_planes[c], want_planes[c]), f"rank {r}: chain + RCT channel {c}"
                assert np.array_equal(got_pal[c], want_pal[c]), f"rank {r}: palette channel {c}"
    finally:
        for ch in chains:
            ch.free()
        for c in ctxs:
            c.close()


def test_modular_config4_pipeline_rccl_single_rank(oracle):
    """the same pipeline over the library's RCCL communicator with one rank (what a 1-GPU box can run)"""
    import jxl_rs_amd
    from jxl_rs_amd import lib
    from jxl_rs_amd.modular import ModularChain
    c = jxl_rs_amd.Context(0, 1)
    ch = None
    try:
        c.comm_init(lib.comm_unique_id(), 0, 1)
        ch = ModularChain(c, 515, 260, seed=5, world=1)
        ch.run_pipeline_rccl(0)
        want_planes, want_pal = helpers.modular_pipeline_oracle(ch, oracle)
        got_planes, got_pal = ch.pipeline_result()
        for k in range(3):
            assert np.array_equal(got_planes[k], want_planes[k]) and np.array_equal(got_pal[k], want_pal[k]), k
        c.comm_destroy()
    finally:
        if ch is not None:
            ch.free()
        c.close()


def test_rct_and_palette_on_unaligned_device_subranges(oracle):
    """a share of a plane may start at any sample: the vectorised kernels fall back to scalar accesses when a
    device pointer is not 16-byte aligned"""
    import ctypes as C
    import jxl_rs_amd
    from helpers import DeviceArray
    rng = np.random.default_rng(7)
    n = 10007
    planes = [rng.integers(-1000, 1000, size=n + 3).astype(np.int32) for _ in range(3)]
    pal = rng.integers(0, 256, size=(3, 64)).astype(np.int32)
    idx = rng.integers(-2, 140, size=n + 3).astype(np.int32)
    ctx = jxl_rs_amd.Context(0, 1)
    bufs = []
    try:
        for off in (1, 2, 3):
            dev = [DeviceArray(p) for p in planes]
            t_idx, t_pal, out = DeviceArray(idx), DeviceArray(pal), DeviceArray(nbytes=4 * 3 * (n + 8))
            bufs += dev + [t_idx, t_pal, out]
            ctx._chk(ctx.L.jxlh_rct(ctx._ctx, C.c_void_p(dev[0].ptr + 4 * off), C.c_void_p(dev[1].ptr + 4 * off),
                                    C.c_void_p(dev[2].ptr + 4 * off), n, 5, 1), "rct")
            ctx._chk(ctx.L.jxlh_palette_strided(ctx._ctx, C.c_void_p(t_idx.ptr + 4 * off), n, C.c_void_p(t_pal.ptr), 64, 64, 3,
                                                8, C.c_void_p(out.ptr + 4 * off), n + 5), "palette_strided")
            ctx.sync()
            want = oracle.rct([p[off:off + n].reshape(1, n) for p in planes], 5, 1)
            for ch in range(3):
                got = dev[ch].download(np.int32, n, 4 * off)
                assert np.array_equal(got, want[ch].reshape(-1)), (off, ch)
                assert np.array_equal(dev[ch].download(np.int32, off), planes[ch][:off])  # untouched before the range
            wp = oracle.palette(idx[off:off + n].reshape(1, n), pal, 64, 3, 8)
            for ch in range(3):
                got = out.download(np.int32, n, 4 * (off + ch * (n + 5)))
                assert np.array_equal(got, wp[ch].reshape(-1)), ("palette", off, ch)
    finally:
        for b in bufs:
            b.free()
        ctx.close()


def _xyb_params(oracle):
    import json
    import os
    k = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kat.json")))["output_stage"]
    return oracle.xyb_params(k["opsin_inverse_matrix"], [k["opsin_bias"]] * 3, 255.0)


@pytest.mark.parametrize("n,bits,channels", [(2, 8, 3), (3, 16, 4), (6, 8, 4)])
def test_local_sharded_output_gather(oracle, n, bits, channels):
    """jxlh_frames_allgather_output_local (round 6): every rank converts ITS band to interleaved 8 / 16-bit sRGB and the
    bands are gathered -- a quarter of the bytes of the f32 plane gather.  Every rank's image == the oracle's conversion
    of the oracle's whole frame, byte for byte."""
    import jxl_rs_amd
    from jxl_rs_amd import lib, synth
    from jxl_rs_amd.lib import DeviceArray
    wl = synth.make_vardct(520, 1000, mix=synth.MIX_ALL, seed=77 + n, epf_iters=2)
    want_planes, _ = run_oracle_frame(oracle, wl)
    xp = _xyb_params(oracle)
    want = (oracle.xyb_to_rgb8 if bits == 8 else oracle.xyb_to_rgb16)(xp, want_planes, wl.xsize, wl.ysize, channels)
    bpr = wl.xsize * channels * (bits // 8)
    ctxs = [jxl_rs_amd.Context(0, 1) for _ in range(n)]
    per = -(-wl.ygroups // n)
    bufs = [DeviceArray(nbytes=n * per * 256 * bpr, device=0) for _ in range(n)]
    try:
        lib.comm_init_local(ctxs)
        for r, c in enumerate(ctxs):
            upload_band(c, wl, band_groups(wl, r, n))
        desc = jxl_rs_amd.Context.output_desc(xyb_params=xp, bits=bits, channels=channels)
        for rep in range(2):
            lib.frames_run_sharded_local(ctxs)
            lib.frames_allgather_output_local(ctxs, desc, [b.ptr for b in bufs], bpr)
            for r, c in enumerate(ctxs):
                c.sync()
                got = bufs[r].download(np.uint8 if bits == 8 else np.uint16, wl.ysize * wl.xsize * channels)
                assert np.array_equal(got.reshape(want.shape), want), f"rank {r}, rep {rep}"
    finally:
        for b in bufs:
            b.free()
        for c in ctxs:
            c.close()


def test_rccl_output_gather_single_rank(oracle):
    """the same through the library's RCCL communicator (one rank: what a 1-GPU box can run)"""
    import jxl_rs_amd
    from jxl_rs_amd import lib, synth
    from jxl_rs_amd.lib import DeviceArray
    wl = synth.make_vardct(520, 600, mix=synth.MIX_D1, seed=18, epf_iters=2)
    want_planes, _ = run_oracle_frame(oracle, wl)
    xp = _xyb_params(oracle)
    want = oracle.xyb_to_rgb8(xp, want_planes, wl.xsize, wl.ysize, 3)
    bpr = wl.xsize * 3
    c = jxl_rs_amd.Context(0, 1)
    buf = DeviceArray(nbytes=wl.ygroups * 256 * bpr, device=0)
    try:
        c.comm_init(lib.comm_unique_id(), 0, 1)
        upload_band(c, wl, set(range(wl.coeffs.shape[0])))
        c.frame_run_sharded()
        c.frame_allgather_output(jxl_rs_amd.Context.output_desc(xyb_params=xp), buf.ptr, bpr)
        c.sync()
        got = buf.download(np.uint8, wl.ysize * bpr).reshape(want.shape)
        assert np.array_equal(got, want)
        c.comm_destroy()
    finally:
        buf.free()
        c.close()
