"""Shared helpers for the parity tests: run the same workload through the CPU oracle and
through the HIP path (via the C ABI), and forward-squeeze for round-trip properties."""
import numpy as np


def _set_shifts(p, wl):
    for c in range(3):
        p.hshift[c] = wl.opts.get("hshift", (0, 0, 0))[c]
        p.vshift[c] = wl.opts.get("vshift", (0, 0, 0))[c]


def is_subsampled(wl):
    return any(wl.opts.get("hshift", (0, 0, 0))) or any(wl.opts.get("vshift", (0, 0, 0)))


def oracle_params_from(o, wl, **over):
    p = o.default_params(wl.xsize, wl.ysize)
    p.epf_iters = wl.opts.get("epf_iters", 2)
    p.gab = 1 if wl.opts.get("gab", True) else 0
    p.do_lf_smoothing = 1 if wl.opts.get("lf_smoothing", True) else 0
    _set_shifts(p, wl)
    p.xsize_blocks, p.ysize_blocks = wl.xblocks, wl.yblocks
    for k, v in over.items():
        setattr(p, k, v)
    return p


def gpu_params_from(ctx, wl, **over):
    p = ctx.default_params(wl.xsize, wl.ysize)
    p.epf_iters = wl.opts.get("epf_iters", 2)
    p.gab = 1 if wl.opts.get("gab", True) else 0
    p.do_lf_smoothing = 1 if wl.opts.get("lf_smoothing", True) else 0
    _set_shifts(p, wl)
    for k, v in over.items():
        setattr(p, k, v)
    return p


def run_oracle_frame(o, wl, num_threads=8, tables=None, **over):
    """Whole chain on the CPU oracle.  Returns (planes cropped to the frame, smoothed LF)."""
    p = oracle_params_from(o, wl, **over)
    if is_subsampled(wl):  # no chroma-from-luma; channel c <- its own coded plane (Y, X, B order in lf_q)
        lf = [o.dequant_lf_channel(p, 0, wl.lf_q[1]), o.dequant_lf_channel(p, 1, wl.lf_q[0]),
              o.dequant_lf_channel(p, 2, wl.lf_q[2])]
    else:
        lf = o.dequant_lf(p, *wl.lf_q)
    tables = wl.tables if tables is None else tables
    planes, lf_sm = o.vardct_frame(p, wl.coeffs, wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob, lf,
                                   tables, num_threads=num_threads)
    return [pl[:wl.ysize, :wl.xsize].copy() for pl in planes], lf_sm


def upload_frame(ctx, wl, **over):
    p = gpu_params_from(ctx, wl, **over)
    ctx.frame_begin(p)
    ctx.set_dequant_tables(wl.tables)
    ctx.set_lf_quantized(*wl.lf_q)
    ctx.set_hf_meta(wl.transform_map, wl.raw_quant, wl.epf_map, wl.ytox, wl.ytob)
    for g in range(wl.coeffs.shape[0]):
        ctx.submit_group(g, wl.coeffs[g])
    ctx.slot_wait(0)
    return p


def run_gpu_frame(ctx, wl, **over):
    upload_frame(ctx, wl, **over)
    ctx.frame_run()
    ctx.sync()
    return ctx.read_planes(), ctx.read_lf()


def bit_equal(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))


def diff_report(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    d = np.abs(a - b)
    bad = np.argwhere(np.asarray(a, dtype=np.float32).view(np.uint32) != np.asarray(b, dtype=np.float32).view(np.uint32))
    first = bad[:5].tolist()
    return f"max abs {d.max():.3e}, rmse {np.sqrt((d * d).mean()):.3e}, mismatches {len(bad)}/{a.size}, first {first}"


# ---- forward squeeze (encoder side; only used to build round-trip properties) ----
def _tendency(b, a, n):
    """smooth_tendency_scalar (squeeze.rs:143-168), vectorised over int64 arrays."""
    b = b.astype(np.int64)
    a = a.astype(np.int64)
    n = n.astype(np.int64)
    diff = np.zeros_like(a)
    up = (b >= a) & (a >= n)
    dn = (b <= a) & (a <= n) & ~up
    d1 = (4 * b - 3 * n - a + 6)
    d1 = np.where(d1 >= 0, d1 // 12, -((-d1) // 12))
    d1 = np.where(d1 - (d1 & 1) > 2 * (b - a), 2 * (b - a) + 1, d1)
    d1 = np.where(d1 + (d1 & 1) > 2 * (a - n), 2 * (a - n), d1)
    d2 = (4 * b - 3 * n - a - 6)
    d2 = np.where(d2 >= 0, d2 // 12, -((-d2) // 12))
    d2 = np.where(d2 + (d2 & 1) < 2 * (b - a), 2 * (b - a) - 1, d2)
    d2 = np.where(d2 - (d2 & 1) < 2 * (a - n), 2 * (a - n), d2)
    diff = np.where(up, d1, diff)
    diff = np.where(dn, d2, diff)
    return diff


def forward_squeeze_h(img):
    """img [h, w] int32 -> (avg [h, ceil(w/2)], res [h, floor(w/2)]) such that the decoder's
    horizontal unsqueeze reproduces img exactly."""
    img = img.astype(np.int64)
    h, w = img.shape
    nr = w // 2
    na = w - nr
    avg = np.zeros((h, na), dtype=np.int64)
    a = img[:, 0:2 * nr:2]
    b = img[:, 1:2 * nr:2]
    avg[:, :nr] = (a + b + (a > b)) >> 1
    if w & 1:
        avg[:, nr] = img[:, w - 1]
    res = np.zeros((h, nr), dtype=np.int64)
    for x in range(nr):
        prev = avg[:, 0] if x == 0 else img[:, 2 * x - 1]
        nxt = avg[:, x + 1] if x + 1 < na else avg[:, x]
        res[:, x] = (a[:, x] - b[:, x]) - _tendency(prev, avg[:, x], nxt)
    return avg.astype(np.int32), res.astype(np.int32)


def forward_squeeze_v(img):
    a, r = forward_squeeze_h(np.ascontiguousarray(img.T))
    return np.ascontiguousarray(a.T), np.ascontiguousarray(r.T)


from jxl_rs_amd.lib import DeviceArray  # noqa: E402,F401  (device buffers for tests that hand DEVICE pointers to the C ABI)


# ---- Modular chain (jxl_rs_amd.modular.ModularChain): what the oracle makes of the same planes ----
def modular_chain_oracle(chain, oracle):
    """the chain's levels one by one on the CPU oracle, then the RCT"""
    cur = [b.copy() for b in chain.base]
    for (hz, ow, oh), res in zip(chain.steps, chain.residuals):
        cur = [oracle.unsqueeze_h(cur[c], res[c], ow) if hz else oracle.unsqueeze_v(cur[c], res[c], oh) for c in range(3)]
    return oracle.rct(cur, *chain.rct) if chain.rct is not None else cur


def modular_pipeline_oracle(chain, oracle):
    """chain + RCT and the palette expansion of BASELINE configs[3]"""
    idx, pal = chain.palette
    return modular_chain_oracle(chain, oracle), list(oracle.palette(idx, pal, pal.shape[1], 3, 8))
