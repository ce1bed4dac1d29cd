"""BASELINE configs[3] as one device-resident unit: the inverse of the default squeeze transform of a w x h image on
three channels (modular/transforms/squeeze.rs:39-105, the decoder's step order), the YCoCg RCT after it
(rct.rs:118-157) and a 256-colour palette expansion (palette.rs:182-199).  `run_chain()` is THE sequence bench.py
times and tests/test_gpu_fullsize.py compares with the oracle at 8192 x 8192: one jxlh_unsqueeze_chain call.

Harness code over the C ABI (device buffers through the HIP runtime, no torch); the inputs follow SURVEY.md 8(d).
"""
import ctypes as C

import numpy as np

from . import synth
from .lib import DeviceArray
from .shard import sample_share


class ModularChain:
    def __init__(self, ctx, w, h, seed=84, rct=(6, 0), planes=None, world=1, palette=None):
        """planes: (base, residuals, steps) as synth.make_modular_planes returns them (default: generated from seed).
        world > 1: output planes sized for the in-place all-gather of `world` equal shares (run_pipeline_sharded);
        palette: (index plane [h, w], table [3, n]) for the pipeline's palette step (default: generated)."""
        self.ctx, self.w, self.h, self.rct = ctx, w, h, rct
        self.world = world
        self.share = sample_share(w * h, 0, world)[2]     # samples per rank the all-gather moves
        self.padded = self.share * world
        self.base, self.residuals, self.steps = planes if planes is not None else synth.make_modular_planes(w, h, seed=seed)
        self.base_h, self.base_w = self.base[0].shape
        self.d_base = [DeviceArray(b, device=ctx.device) for b in self.base]
        self.d_res = [[DeviceArray(r, device=ctx.device) if r.size else None for r in lvl] for lvl in self.residuals]
        self.d_out = [DeviceArray(nbytes=max(w * h, self.padded) * 4, device=ctx.device) for _ in range(3)]
        self.palette = palette
        self.d_idx = self.d_pal = self.d_pout = None
        self.levels = []
        for (hz, ow, oh), res, dres in zip(self.steps, self.residuals, self.d_res):
            self.levels.append((hz, ow, oh, [d.ptr if d is not None else None for d in dres], max(res[0].shape[1], 1)))
        # samples every level writes (3 planes): the chain's algorithmic traffic is 8 B per written sample
        self.samples_written = 3 * sum(ow * oh for _, ow, oh in self.steps)

    def run_chain(self):
        """unsqueeze levels smallest first, the last one fused with the RCT: one ABI call, asynchronous"""
        self.ctx.unsqueeze_chain(self.levels, [d.ptr for d in self.d_base], self.base_w, self.base_w, self.base_h,
                                 [d.ptr for d in self.d_out], self.w, rct=self.rct)

    # ---- BASELINE configs[3] across GPUs: "Squeeze + RCT + Palette, group shard + all-gather".  The squeeze
    # recurrence is serial along whole lines (replicas only), so every rank runs the chain; the
    # per-sample transforms behind it are sharded -- rank r runs the RCT in place on ITS share of the chain's output
    # and expands ITS share of the palette indices -- and the planes are joined by in-place all-gathers.
    def _ensure_palette(self):
        if self.d_idx is not None:
            return
        if self.palette is None:
            rng = np.random.default_rng(256)
            self.palette = (rng.integers(-3, 300, size=(self.h, self.w)).astype(np.int32),
                            rng.integers(0, 256, size=(3, 256)).astype(np.int32))
        idx, pal = self.palette
        self.d_idx, self.d_pal = DeviceArray(idx, device=self.ctx.device), DeviceArray(pal, device=self.ctx.device)
        self.d_pout = DeviceArray(nbytes=3 * self.padded * 4, device=self.ctx.device)

    def run_local_shares(self, rank):
        """this rank's part before the joins: replicated chain (no RCT), RCT + palette on the own sample share"""
        self._ensure_palette()
        ctx, n = self.ctx, self.w * self.h
        ctx.unsqueeze_chain(self.levels, [d.ptr for d in self.d_base], self.base_w, self.base_w, self.base_h,
                            [d.ptr for d in self.d_out], self.w, rct=None)
        i0, i1, _ = sample_share(n, rank, self.world)
        if i1 > i0 and self.rct is not None:
            p = [C.c_void_p(d.ptr + 4 * i0) for d in self.d_out]
            ctx._chk(ctx.L.jxlh_rct(ctx._ctx, p[0], p[1], p[2], i1 - i0, self.rct[0], self.rct[1]), "rct")
        if i1 > i0:
            ncol = self.palette[1].shape[1]
            ctx._chk(ctx.L.jxlh_palette_strided(ctx._ctx, C.c_void_p(self.d_idx.ptr + 4 * i0), i1 - i0,
                                                C.c_void_p(self.d_pal.ptr), ncol, ncol, 3, 8,
                                                C.c_void_p(self.d_pout.ptr + 4 * i0), self.padded), "palette_strided")

    def gather_buffers(self):
        """(device pointer, bytes per rank) of every buffer the pipeline joins: 3 RCT planes, 3 palette planes"""
        return ([(d.ptr, 4 * self.share) for d in self.d_out] +
                [(self.d_pout.ptr + 4 * ch * self.padded, 4 * self.share) for ch in range(3)])

    def run_pipeline_rccl(self, rank):
        """one process per GPU: the library's RCCL communicator (ctx.comm_init) joins the shares"""
        self.run_local_shares(rank)
        for ptr, nbytes in self.gather_buffers():
            self.ctx.comm_allgather(ptr, nbytes)

    def pipeline_result(self):
        self.ctx.sync()
        n = self.w * self.h
        planes = [d.download(np.int32, n).reshape(self.h, self.w) for d in self.d_out]
        pal = [self.d_pout.download(np.int32, n, 4 * ch * self.padded).reshape(self.h, self.w) for ch in range(3)]
        return planes, pal

    def result(self):
        self.ctx.sync()
        return [d.download(np.int32, self.w * self.h).reshape(self.h, self.w) for d in self.d_out]

    def free(self):
        for d in self.d_base + self.d_out + [d for lvl in self.d_res for d in lvl if d is not None]:
            d.free()
        for d in (self.d_idx, self.d_pal, self.d_pout):
            if d is not None:
                d.free()


def run_pipeline_local(chains, ctxs):
    """in-process ranks (one context per rank, jxlh_comm_init_local): every rank's shares, then the joins"""
    from . import lib
    for r, ch in enumerate(chains):
        ch.run_local_shares(r)
    for k in range(6):
        bufs = [ch.gather_buffers()[k] for ch in chains]
        lib.comm_allgather_local(ctxs, [b[0] for b in bufs], bufs[0][1])
