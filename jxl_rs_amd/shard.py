"""Sharding of one VarDCT frame across ranks: contiguous bands of group rows.

K1 has no cross-group dependence; the filters need a 4..7 pixel halo.  The library (csrc/comm.hip) runs K1 on
exactly the rank's band, exchanges one block row per band edge with the neighbour rank and all-gathers the
finished planes (`jxlh_frame_run_sharded`, `jxlh_frame_allgather`); `jxlh_frame_run(row0, row1)` is the
stand-alone form that recomputes a halo group row instead.  The helpers here mirror the library's partition for
host code that distributes coefficient groups to ranks and for the CPU tests."""


def band_for_rank(ygroups, rank, world):
    """[row0, row1) group rows owned by `rank`; the last ranks may be empty on tiny frames."""
    per = (ygroups + world - 1) // world
    row0 = min(rank * per, ygroups)
    row1 = min((rank + 1) * per, ygroups)
    return row0, row1, per


def band_pixel_rows(row0, row1, ysize, group_dim=256):
    return row0 * group_dim, min(row1 * group_dim, ysize)


def assemble(gathered, ygroups, world, ysize, group_dim=256):
    """gathered: [world][3][per*group_dim][xsize] (all_gather output) -> [3][ysize][xsize]."""
    import numpy as np
    per = (ygroups + world - 1) // world
    chunks = []
    for r in range(world):
        row0, row1, _ = band_for_rank(ygroups, r, world)
        y0, y1 = band_pixel_rows(row0, row1, ysize, group_dim)
        if y1 > y0:
            chunks.append(gathered[r][:, : y1 - y0, :])
    return np.concatenate(chunks, axis=1) if chunks else None


def sample_share(n, rank, world, align=4):
    """[i0, i1) samples of a per-sample Modular transform (RCT, non-delta Palette) that `rank` runs; shares are
    ceil(n / world) rounded up to `align` samples (16-byte aligned starts for the vectorised kernels), the count per
    rank being what the in-place all-gather moves (the buffers must hold world * count samples)."""
    count = -(-n // world)
    count = -(-count // align) * align
    i0 = min(rank * count, n)
    return i0, min(i0 + count, n), count
