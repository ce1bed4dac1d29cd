"""Multi-rank path on CPU (gloo, world_size 2 and 3): band partition of group rows, transforms on the own band,
halo EXCHANGE of the edge block rows with the neighbour ranks (send / recv), filters on the band, all-gather and
reassembly must reproduce the whole-frame result bit for bit.  The compute stand-in is the oracle (HIP kernels
cannot run here); the protocol is the one `jxlh_frame_run_sharded` + `jxlh_frame_allgather` implement over RCCL
and `bench.py --gpus N` drives; tests/test_gpu_sharding.py checks the device side of it with the in-process
transport, tests/test_gpu_parity.py::test_band_runs_equal_whole_frame the halo-recompute form."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_band_partition_covers_everything():
    from jxl_rs_amd.shard import band_for_rank
    for ygroups in (1, 2, 3, 7, 32, 33):
        for world in (1, 2, 4, 8):
            rows = []
            for r in range(world):
                r0, r1, per = band_for_rank(ygroups, r, world)
                assert 0 <= r0 <= r1 <= ygroups and r1 - r0 <= per
                rows += list(range(r0, r1))
            assert rows == list(range(ygroups))


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from jxl_rs_amd import synth
    from jxl_rs_amd.shard import assemble, band_for_rank, band_pixel_rows
    from oracle.oracle import Oracle
    from helpers import oracle_params_from

    o = Oracle(fused=True)
    wl = synth.make_vardct(300, 700, mix=synth.MIX_D1, seed=21, epf_iters=2)  # 3 group rows -> 2 + 1 or 1 + 1 + 1
    p = oracle_params_from(o, wl)
    lf = o.adaptive_lf_smoothing(p, o.dequant_lf(p, *wl.lf_q))
    row0, row1, per = band_for_rank(wl.ygroups, rank, world)
    # the rank holds only its own band's coefficients
    coeffs = np.full_like(wl.coeffs, 0x3FFF)
    coeffs[row0 * wl.xgroups:row1 * wl.xgroups] = wl.coeffs[row0 * wl.xgroups:row1 * wl.xgroups]

    def exchange(planes):
        """one block row (8 pixel rows x 3 channels) per band edge, like the grouped ncclSend / ncclRecv"""
        reqs, recvs = [], []
        for nb, send_rows, recv_rows in ((rank - 1, (row0 * 256, row0 * 256 + 8), (row0 * 256 - 8, row0 * 256)),
                                         (rank + 1, (row1 * 256 - 8, row1 * 256), (row1 * 256, row1 * 256 + 8))):
            if nb < 0 or nb >= world or row0 >= row1:
                continue
            n0, n1, _ = band_for_rank(wl.ygroups, nb, world)
            if n0 >= n1:
                continue
            out = torch.from_numpy(np.stack([pl[send_rows[0]:send_rows[1]] for pl in planes]).copy())
            buf = torch.zeros_like(out)
            reqs.append(dist.isend(out, nb))
            reqs.append(dist.irecv(buf, nb))
            recvs.append((buf, recv_rows))
        for r in reqs:
            r.wait()
        for buf, (a, b) in recvs:
            for c in range(3):
                planes[c][a:b] = buf[c].numpy()

    planes = o.vardct_band(p, coeffs, wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob, lf,
                           wl.tables, row0, row1, exchange=exchange)
    y0, y1 = band_pixel_rows(row0, row1, wl.ysize)
    band = torch.zeros((3, per * 256, wl.xsize), dtype=torch.float32)
    for c in range(3):
        band[c, : y1 - y0] = torch.from_numpy(planes[c][y0:y1, : wl.xsize].copy())
    full = torch.zeros((world * 3, per * 256, wl.xsize), dtype=torch.float32)  # concatenated along dim 0
    dist.all_gather_into_tensor(full, band)
    frame = assemble(full.view(world, 3, per * 256, wl.xsize).numpy(), wl.ygroups, world, wl.ysize)
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), frame)
    # ---- the second gather form (round 6, jxlh_frame_allgather_output): every rank converts ITS band's rows to
    # interleaved 8-bit sRGB and the bands are gathered in place in the whole image (a quarter of the bytes)
    import json
    k = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_kat.json")))["output_stage"]
    xp = o.xyb_params(k["opsin_inverse_matrix"], [k["opsin_bias"]] * 3, 255.0)
    image = torch.zeros((world, per * 256, wl.xsize, 3), dtype=torch.uint8)
    if y1 > y0:
        rgb = o.xyb_to_rgb8(xp, [np.ascontiguousarray(planes[c][y0:y1, : wl.xsize]) for c in range(3)], wl.xsize, y1 - y0, 3)
        image[rank, : y1 - y0] = torch.from_numpy(rgb.reshape(y1 - y0, wl.xsize, 3).copy())
    mine = image[rank].clone()
    dist.all_gather_into_tensor(image.view(world * per * 256, wl.xsize, 3), mine)
    rows = []
    for r in range(world):
        a0, a1, _ = band_for_rank(wl.ygroups, r, world)
        b0, b1 = band_pixel_rows(a0, a1, wl.ysize)
        rows.append(image[r, : b1 - b0].numpy())
    np.save(os.path.join(out_dir, f"rgb_rank{rank}.npy"), np.concatenate(rows))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_band_sharding_with_halo_exchange_reassembles_the_frame(tmp_path, oracle, world):
    import torch.multiprocessing as mp
    from jxl_rs_amd import synth
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import run_oracle_frame
    port = 29500 + (os.getpid() * 7 + world) % 2000
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    wl = synth.make_vardct(300, 700, mix=synth.MIX_D1, seed=21, epf_iters=2)
    want, _ = run_oracle_frame(oracle, wl)
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npy")
        assert got.shape == (3, 700, 300)
        for c in range(3):
            assert np.array_equal(got[c].view(np.uint32), want[c].view(np.uint32)), (r, c)
    # the converted-image gather: every rank's image == the conversion of the whole frame
    import json
    k = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_kat.json")))["output_stage"]
    xp = oracle.xyb_params(k["opsin_inverse_matrix"], [k["opsin_bias"]] * 3, 255.0)
    want_rgb = oracle.xyb_to_rgb8(xp, want, wl.xsize, wl.ysize, 3).reshape(wl.ysize, wl.xsize, 3)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"rgb_rank{r}.npy"), want_rgb), r


# ---------------------------------------------------------------------------------------------------------------
# BASELINE configs[3] across ranks: replicated squeeze chain -> RCT and palette on the own sample share -> all-gather.
# The protocol jxl_rs_amd.modular.ModularChain.run_pipeline_rccl runs over the library's RCCL communicator (and
# tests/test_gpu_sharding.py::test_modular_config4_pipeline_sharded over the in-process transport), with the oracle as
# the compute stand-in and gloo as the transport.
def _modular_worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from jxl_rs_amd import synth
    from jxl_rs_amd.shard import sample_share
    from oracle.oracle import Oracle
    o = Oracle(fused=True)
    w, h = 301, 207
    base, residuals, steps = synth.make_modular_planes(w, h, seed=77)
    cur = [b.copy() for b in base]
    for (hz, ow, oh), res in zip(steps, residuals):     # the recurrence is serial along a line: every rank runs it
        cur = [o.unsqueeze_h(cur[c], res[c], ow) if hz else o.unsqueeze_v(cur[c], res[c], oh) for c in range(3)]
    n = w * h
    i0, i1, count = sample_share(n, rank, world)
    flat = [np.zeros(world * count, np.int32) for _ in range(3)]
    share = o.rct([cur[c].reshape(-1)[i0:i1].reshape(1, -1) for c in range(3)], 6, 0) if i1 > i0 else None
    rng = np.random.default_rng(256)
    idx = rng.integers(-3, 300, size=n).astype(np.int32)
    pal = rng.integers(0, 256, size=(3, 256)).astype(np.int32)
    pshare = o.palette(idx[i0:i1].reshape(1, -1), pal, 256, 3, 8) if i1 > i0 else None
    out = {}
    for name, sh in (("rct", share), ("pal", pshare)):
        for c in range(3):
            mine = torch.zeros(count, dtype=torch.int32)
            if sh is not None:
                mine[: i1 - i0] = torch.from_numpy(np.asarray(sh[c]).reshape(-1).copy())
            full = torch.zeros(world * count, dtype=torch.int32)
            dist.all_gather_into_tensor(full, mine)
            out[f"{name}{c}"] = full.numpy()[:n].reshape(h, w)
    np.savez(os.path.join(out_dir, f"modular_rank{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_modular_pipeline_shares_reassemble(tmp_path, oracle, world):
    import torch.multiprocessing as mp
    from jxl_rs_amd import synth
    port = 31500 + (os.getpid() * 5 + world) % 2000
    mp.spawn(_modular_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    w, h = 301, 207
    base, residuals, steps = synth.make_modular_planes(w, h, seed=77)
    cur = [b.copy() for b in base]
    for (hz, ow, oh), res in zip(steps, residuals):
        cur = [oracle.unsqueeze_h(cur[c], res[c], ow) if hz else oracle.unsqueeze_v(cur[c], res[c], oh) for c in range(3)]
    want = oracle.rct(cur, 6, 0)
    rng = np.random.default_rng(256)
    idx = rng.integers(-3, 300, size=w * h).astype(np.int32)
    pal = rng.integers(0, 256, size=(3, 256)).astype(np.int32)
    want_pal = oracle.palette(idx.reshape(h, w), pal, 256, 3, 8)
    for r in range(world):
        got = np.load(tmp_path / f"modular_rank{r}.npz")
        for c in range(3):
            assert np.array_equal(got[f"rct{c}"], want[c]), (r, c)
            assert np.array_equal(got[f"pal{c}"], want_pal[c]), (r, c)
