"""Multi-rank path on CPU (gloo, world_size 2): band partition of group rows, per-rank band
computation with halo recompute, all-gather and reassembly must reproduce the whole-frame result
bit for bit.  The compute stand-in is the oracle (HIP kernels cannot run here); the band logic is
the one `jxlh_frame_run(row0, row1)` implements and `bench.py --strong` drives, and
tests/test_gpu_parity.py::test_band_runs_equal_whole_frame checks the device side of it."""
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
    wl = synth.make_vardct(300, 700, mix=synth.MIX_D1, seed=21, epf_iters=2)  # 3 group rows -> 2 + 1
    p = oracle_params_from(o, wl)
    lf = o.adaptive_lf_smoothing(p, o.dequant_lf(p, *wl.lf_q))
    row0, row1, per = band_for_rank(wl.ygroups, rank, world)
    planes = o.vardct_band(p, wl.coeffs, wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob, lf,
                           wl.tables, row0, row1)
    y0, y1 = band_pixel_rows(row0, row1, wl.ysize)
    band = torch.zeros((3, per * 256, wl.xsize), dtype=torch.float32)
    for c in range(3):
        band[c, : y1 - y0] = torch.from_numpy(planes[c][y0:y1, : wl.xsize].copy())
    full = torch.zeros((world * 3, per * 256, wl.xsize), dtype=torch.float32)  # concatenated along dim 0
    dist.all_gather_into_tensor(full, band)
    frame = assemble(full.view(world, 3, per * 256, wl.xsize).numpy(), wl.ygroups, world, wl.ysize)
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), frame)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_band_sharding_reassembles_the_frame(tmp_path, oracle):
    import torch.multiprocessing as mp
    from jxl_rs_amd import synth
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import run_oracle_frame
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    wl = synth.make_vardct(300, 700, mix=synth.MIX_D1, seed=21, epf_iters=2)
    want, _ = run_oracle_frame(oracle, wl)
    for r in range(2):
        got = np.load(tmp_path / f"rank{r}.npy")
        assert got.shape == (3, 700, 300)
        for c in range(3):
            assert np.array_equal(got[c].view(np.uint32), want[c].view(np.uint32)), (r, c)
