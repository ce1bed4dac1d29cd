"""Two PROCESSES, one GPU each, over the library's RCCL transport: jxlh_comm_init with a shared
unique id, jxlh_frame_run_sharded (transforms on the own band, ncclSend / ncclRecv of the edge block rows, filters),
jxlh_frame_allgather (and jxlh_frame_allgather_output: the converted 8-bit image), and the Modular pipeline (replicated squeeze chain, RCT + palette on the rank's share, six
all-gathers) -- every rank must end with the oracle's whole frame bit for bit.  This is the only test in which
ncclSend / ncclRecv meet a neighbour; it needs two visible devices and is skipped (but collected) on a one-GPU box,
where tests/test_gpu_sharding.py covers the same band logic with the in-process transport and a one-rank communicator.
The reference's counterpart is the join of the group-parallel render (frame/render.rs:461-479)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _device_count():
    """through the HIP runtime the library itself is linked against -- NOT torch: torch bundles its own runtime, and
    initialising it in this process between the library's load and its first call leaves the library without a device
    (every jxlh_ctx_create of the suite then fails; measured round 5)"""
    try:
        import ctypes as C
        from jxl_rs_amd.lib import DeviceArray
        n = C.c_int(0)
        return n.value if DeviceArray.hip().hipGetDeviceCount(C.byref(n)) == 0 else 0
    except Exception:  # no library, no runtime, no devices
        return 0


def _worker(rank, world, uid_q, res_q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import jxl_rs_amd
        from jxl_rs_amd import lib, synth
        from jxl_rs_amd.modular import ModularChain
        from oracle.oracle import Oracle
        import helpers
        from test_gpu_sharding import band_groups, upload_band

        o = Oracle(fused=True)
        uid = uid_q.get(timeout=120)
        # ---- VarDCT: one frame in bands of group rows
        wl = synth.make_vardct(520, 1000, mix=synth.MIX_ALL, seed=33, epf_iters=2)  # 4 group rows -> 2 + 2
        want, _ = helpers.run_oracle_frame(o, wl)
        c = jxl_rs_amd.Context(rank, 1)
        c.comm_init(uid[0], rank, world)
        upload_band(c, wl, band_groups(wl, rank, world))  # the other band's groups are poisoned
        assert c.comm_band()[:2] == (rank, world)
        for rep in range(2):
            c.frame_run_sharded()
            c.frame_allgather()
            c.sync()
            got = c.read_planes()
            for ch in range(3):
                assert helpers.bit_equal(got[ch], want[ch]), f"rank {rank} plane {ch} rep {rep}: {helpers.diff_report(got[ch], want[ch])}"
        # ---- the converted-image gather (round 6): every rank's interleaved 8-bit sRGB image == the oracle's
        import json
        from jxl_rs_amd.lib import DeviceArray
        k = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_kat.json")))["output_stage"]
        xp = o.xyb_params(k["opsin_inverse_matrix"], [k["opsin_bias"]] * 3, 255.0)
        want_rgb = o.xyb_to_rgb8(xp, want, wl.xsize, wl.ysize, 3)
        bpr = wl.xsize * 3
        per = -(-wl.ygroups // world)
        img = DeviceArray(nbytes=world * per * 256 * bpr, device=rank)
        c.frame_run_sharded()
        c.frame_allgather_output(jxl_rs_amd.Context.output_desc(xyb_params=xp), img.ptr, bpr)
        c.sync()
        got_rgb = img.download(np.uint8, wl.ysize * bpr).reshape(want_rgb.shape)
        assert np.array_equal(got_rgb, want_rgb), f"rank {rank}: converted-image gather"
        img.free()
        c.comm_destroy()
        c.close()
        # ---- Modular: replicated chain, sharded RCT + palette, six all-gathers
        m = jxl_rs_amd.Context(rank, 1)
        m.comm_init(uid[1], rank, world)
        chain = ModularChain(m, 515, 260, seed=5, world=world)
        chain.run_pipeline_rccl(rank)
        want_planes, want_pal = helpers.modular_pipeline_oracle(chain, o)
        got_planes, got_pal = chain.pipeline_result()
        for k in range(3):
            assert np.array_equal(got_planes[k], want_planes[k]) and np.array_equal(got_pal[k], want_pal[k]), (rank, k)
        m.comm_destroy()
        chain.free()
        m.close()
        res_q.put((rank, "ok"))
    except BaseException as e:  # the parent reports it
        import traceback
        res_q.put((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__))[-3000:]))


def _run_world(world):
    import multiprocessing as mp
    from jxl_rs_amd import lib
    ctx = mp.get_context("spawn")
    uid_qs = [ctx.Queue() for _ in range(world)]
    res_q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, uid_qs[r], res_q)) for r in range(world)]
    for p in procs:
        p.start()
    ids = (lib.comm_unique_id(), lib.comm_unique_id())  # VarDCT communicator, Modular communicator
    for q in uid_qs:
        q.put(ids)
    results = {}
    try:
        for _ in range(world):
            rank, msg = res_q.get(timeout=600)
            results[rank] = msg
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    assert results == {r: "ok" for r in range(world)}, results


@pytest.mark.skipif(_device_count() < 2, reason="needs two visible GPUs (one process per GPU over RCCL)")
def test_two_processes_share_a_frame_over_rccl():
    _run_world(2)


def test_the_same_worker_as_a_single_spawned_rank():
    """the harness and the worker's protocol with world = 1 (a spawned process, its own HIP context and communicator):
    what a one-GPU box can run of the test above"""
    _run_world(1)
