"""BASELINE configs[3] as one device-resident unit: the inverse of the default squeeze transform of a w x h image on
three channels (modular/transforms/squeeze.rs:39-105, the decoder's step order), the YCoCg RCT after it
(rct.rs:118-157) and a 256-colour palette expansion (palette.rs:182-199).  `run_chain()` is THE sequence bench.py
times and tests/test_gpu_fullsize.py compares with the oracle at 8192 x 8192: one jxlh_unsqueeze_chain call.

Harness code over the C ABI (device buffers through the HIP runtime, no torch); the inputs follow SURVEY.md 8(d).
"""
import numpy as np

from . import synth
from .lib import DeviceArray


class ModularChain:
    def __init__(self, ctx, w, h, seed=84, rct=(6, 0), planes=None):
        """planes: (base, residuals, steps) as synth.make_modular_planes returns them (default: generated from seed)"""
        self.ctx, self.w, self.h, self.rct = ctx, w, h, rct
        self.base, self.residuals, self.steps = planes if planes is not None else synth.make_modular_planes(w, h, seed=seed)
        self.base_h, self.base_w = self.base[0].shape
        self.d_base = [DeviceArray(b) for b in self.base]
        self.d_res = [[DeviceArray(r) if r.size else None for r in lvl] for lvl in self.residuals]
        self.d_out = [DeviceArray(nbytes=w * h * 4) for _ in range(3)]
        self.levels = []
        for (hz, ow, oh), res, dres in zip(self.steps, self.residuals, self.d_res):
            self.levels.append((hz, ow, oh, [d.ptr if d is not None else None for d in dres], max(res[0].shape[1], 1)))
        # samples every level writes (3 planes): the chain's algorithmic traffic is 8 B per written sample
        self.samples_written = 3 * sum(ow * oh for _, ow, oh in self.steps)

    def run_chain(self):
        """unsqueeze levels smallest first, the last one fused with the RCT: one ABI call, asynchronous"""
        self.ctx.unsqueeze_chain(self.levels, [d.ptr for d in self.d_base], self.base_w, self.base_w, self.base_h,
                                 [d.ptr for d in self.d_out], self.w, rct=self.rct)

    def result(self):
        self.ctx.sync()
        return [d.download(np.int32, self.w * self.h).reshape(self.h, self.w) for d in self.d_out]

    def oracle_result(self, oracle):
        cur = [b.copy() for b in self.base]
        for (hz, ow, oh), res in zip(self.steps, self.residuals):
            cur = [oracle.unsqueeze_h(cur[c], res[c], ow) if hz else oracle.unsqueeze_v(cur[c], res[c], oh) for c in range(3)]
        return oracle.rct(cur, *self.rct) if self.rct is not None else cur

    def free(self):
        for d in self.d_base + self.d_out + [d for lvl in self.d_res for d in lvl if d is not None]:
            d.free()
