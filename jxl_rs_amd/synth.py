"""Synthetic VarDCT / Modular workloads in the hot path's *input* format.

There is no JPEG XL encoder and no Rust toolchain in this environment, so inputs are
generated directly as what the host-side entropy decoder of jxl-rs would hand over:
per-group dense i32 coefficient slabs (frame/group.rs:437-440), the HfMetadata maps
(frame/mod.rs:169-176), the quantised LF image and the dequantisation tables.

Everything here is deterministic in (config, seed) via numpy's PCG64.
"""
import math
from dataclasses import dataclass, field

import numpy as np

COVERED_X = [1, 1, 1, 1, 2, 4, 1, 2, 1, 4, 2, 4, 1, 1, 1, 1, 1, 1, 8, 4, 8, 16, 8, 16, 32, 16, 32]
COVERED_Y = [1, 1, 1, 1, 2, 4, 2, 1, 4, 1, 4, 2, 1, 1, 1, 1, 1, 1, 8, 8, 4, 16, 16, 8, 32, 32, 16]
TABLE_FOR_TYPE = [0, 1, 2, 3, 4, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 10, 10, 11, 12, 12, 13, 14, 14, 15, 16, 16]
REQ_X = [1, 1, 1, 1, 2, 4, 1, 1, 2, 1, 1, 8, 4, 16, 8, 32, 16]
REQ_Y = [1, 1, 1, 1, 2, 4, 2, 4, 4, 1, 1, 8, 8, 16, 16, 32, 32]

# type mixes (type id -> area weight)
MIX_DCT8 = {0: 1.0}
MIX_D1 = {0: 0.50, 4: 0.10, 6: 0.08, 7: 0.08, 5: 0.08, 10: 0.04, 11: 0.04, 8: 0.04, 9: 0.04}
# every 8x8 transform: what a chroma-subsampled (JPEG-recompression) frame may hold
MIX_8X8 = {0: 0.55, 1: 0.05, 2: 0.05, 3: 0.05, 12: 0.05, 13: 0.05, 14: 0.05, 15: 0.05, 16: 0.05, 17: 0.05}
MIX_ALL = {0: 0.30, 4: 0.06, 6: 0.05, 7: 0.05, 5: 0.06, 10: 0.03, 11: 0.03, 8: 0.03, 9: 0.03,
           1: 0.012, 2: 0.012, 3: 0.012, 12: 0.012, 13: 0.012, 14: 0.01, 15: 0.01, 16: 0.01, 17: 0.01,
           18: 0.06, 19: 0.03, 20: 0.03, 21: 0.04, 22: 0.02, 23: 0.02, 24: 0.04, 25: 0.02, 26: 0.02}


# ----------------------------------------------------------------------------------------
# default dequantisation matrices (numpy restatement of the spec's library tables; the same
# data jxl-rs computes in DequantMatrices::get_library_table, frame/quant_weights.rs:378-1079)
def _f(x):
    return np.float32(x)


def _mult(v):
    v = _f(v)
    return _f(1) + v if v > 0 else _f(1) / (_f(1) - v)


def _bands(params):
    b = [_f(params[0])]
    for p in params[1:]:
        b.append(_f(b[-1] * _mult(p)))
    return np.array(b, dtype=np.float32)


def _get_quant_weights(rows, cols, params3):
    out = np.zeros((3, rows, cols), dtype=np.float32)
    for c in range(3):
        bands = _bands(params3[c])
        nb = len(bands)
        scale = _f(nb - 1) / (_f(math.sqrt(2.0)) + _f(1e-6))
        rcpcol = _f(scale / _f(cols - 1))
        rcprow = _f(scale / _f(rows - 1))
        dy = (np.arange(rows, dtype=np.float32) * rcprow).astype(np.float32)
        dx = (np.arange(cols, dtype=np.float32) * rcpcol).astype(np.float32)
        dist = np.sqrt((dx[None, :] * dx[None, :]).astype(np.float32) + (dy[:, None] * dy[:, None]).astype(np.float32),
                       dtype=np.float32)
        if nb == 1:
            out[c] = bands[0]
            continue
        idxf = np.floor(dist).astype(np.float32)
        frac = (dist - idxf).astype(np.float32)
        idx = idxf.astype(np.int64)
        a = bands[idx]
        b = bands[np.minimum(idx + 1, nb - 1)]
        out[c] = (np.power((b / a).astype(np.float32), frac, dtype=np.float32) * a).astype(np.float32)
    return out


_DCT8 = [[3150.0, 0.0, -0.4, -0.4, -0.4, -2.0], [560.0, 0.0, -0.3, -0.3, -0.3, -0.3], [512.0, -2.0, -1.0, 0.0, -1.0, -2.0]]
_DCT4X4 = [[2200.0, 0.0, 0.0, 0.0], [392.0, 0.0, 0.0, 0.0], [112.0, -0.25, -0.25, -0.5]]
_DCT16 = [[8996.8725711814115328, -1.3000777393353804, -0.49424529824571225, -0.439093774457103443,
           -0.6350101832695744, -0.90177264050827612, -1.6162099239887414],
          [3191.48366296844234752, -0.67424582104194355, -0.80745813428471001, -0.44925837484843441,
           -0.35865440981033403, -0.31322389111877305, -0.37615025315725483],
          [1157.50408145487200256, -2.0531423165804414, -1.4, -0.50687130033378396, -0.42708730624733904,
           -1.4856834539296244, -4.9209142884401604]]
_DCT32 = [[15718.40830982518931456, -1.025, -0.98, -0.9012, -0.4, -0.48819395464, -0.421064, -0.27],
          [7305.7636810695983104, -0.8041958212306401, -0.7633036457487539, -0.55660379990111464,
           -0.49785304658857626, -0.43699592683512467, -0.40180866526242109, -0.27321683125358037],
          [3803.53173721215041536, -3.060733579805728, -2.0413270132490346, -2.0235650159727417,
           -0.5495389509954993, -0.4, -0.4, -0.3]]
_DCT8X16 = [[7240.7734393502, -0.7, -0.7, -0.2, -0.2, -0.2, -0.5], [1448.15468787004, -0.5, -0.5, -0.5, -0.2, -0.2, -0.2],
            [506.854140754517, -1.4, -0.2, -0.5, -0.5, -1.5, -3.6]]
_DCT8X32 = [[16283.2494710648897, -1.7812845336559429, -1.6309059012653515, -1.0382179034313539, -0.85, -0.7, -0.9,
             -1.2360638576849587],
            [5089.15750884921511936, -0.320049391452786891, -0.35362849922161446, -0.30340000000000003, -0.61, -0.5,
             -0.5, -0.6],
            [3397.77603275308720128, -0.321327362693153371, -0.34507619223117997, -0.70340000000000003, -0.9, -1.0,
             -1.0, -1.1754605576265209]]
_DCT16X32 = [[13844.97076442300573, -0.97113799999999995, -0.658, -0.42026, -0.22712, -0.2206, -0.226, -0.6],
             [4798.964084220744293, -0.61125308982767057, -0.83770786552491361, -0.79014862079498627,
              -0.2692727459704829, -0.38272769465388551, -0.22924222653091453, -0.20719098826199578],
             [1807.236946760964614, -1.2, -1.2, -0.7, -0.7, -0.7, -0.4, -0.5]]
_DCT4X8 = [[2198.050556016380522, -0.96269623020744692, -0.76194253026666783, -0.6551140670773547],
           [764.3655248643528689, -0.92630200888366945, -0.9675229603596517, -0.27845290869168118],
           [527.107573587542228, -1.4594385811273854, -1.450082094097871593, -1.5843722511996204]]
_LARGE_TAIL = [[-1.025, -0.78, -0.65012, -0.19041574084286472, -0.20819395464, -0.421064, -0.32733845535848671],
               [-0.3041958212306401, -0.3633036457487539, -0.35660379990111464, -0.3443074455424403,
                -0.33699592683512467, -0.30180866526242109, -0.27321683125358037],
               [-1.2, -1.2, -0.8, -0.7, -0.7, -0.4, -0.5]]
_BASE_SQ = [26629.073922049845, 9311.3238710010046, 4992.2486445538634]
_BASE_RC = [23629.073922049845, 8611.3238710010046, 4492.2486445538634]


def _large(mul, base):
    return [[float(_f(mul) * _f(base[c]))] + _LARGE_TAIL[c] for c in range(3)]


def _interpolate(pos, mx, bands):
    scaled = _f(_f(pos) * _f(len(bands) - 1) / _f(mx))
    idx = int(scaled)
    a, b = bands[idx], bands[idx + 1]
    return _f(a * np.power(_f(b / a), _f(scaled - _f(idx)), dtype=np.float32))


def library_dequant_table(t):
    """Table t (0..16): float32 [3 * 64*REQ_X*REQ_Y], inverse weights, channel-major."""
    rows, cols = 8 * REQ_X[t], 8 * REQ_Y[t]
    w = np.zeros((3, rows, cols), dtype=np.float32)
    if t == 0:
        w = _get_quant_weights(rows, cols, _DCT8)
    elif t == 1:
        xyb = [[280.0, 3160.0, 3160.0], [60.0, 864.0, 864.0], [18.0, 200.0, 200.0]]
        for c in range(3):
            w[c, :, :] = xyb[c][0]
            w[c, 0, 1] = xyb[c][1]
            w[c, 1, 0] = xyb[c][1]
            w[c, 1, 1] = xyb[c][2]
    elif t == 2:
        xyb = [[3840.0, 2560.0, 1280.0, 640.0, 480.0, 300.0], [960.0, 640.0, 320.0, 180.0, 140.0, 120.0],
               [640.0, 320.0, 128.0, 64.0, 32.0, 16.0]]
        for c in range(3):
            w[c, 0, 0] = float(0xBAD)
            w[c, 0, 1] = w[c, 1, 0] = xyb[c][0]
            w[c, 1, 1] = xyb[c][1]
            w[c, 0:2, 2:4] = xyb[c][2]
            w[c, 2:4, 0:2] = xyb[c][2]
            w[c, 2:4, 2:4] = xyb[c][3]
            w[c, 0:4, 4:8] = xyb[c][4]
            w[c, 4:8, 0:4] = xyb[c][4]
            w[c, 4:8, 4:8] = xyb[c][5]
    elif t == 3:
        w44 = _get_quant_weights(4, 4, _DCT4X4)
        w = np.repeat(np.repeat(w44, 2, axis=1), 2, axis=2)
    elif t == 4:
        w = _get_quant_weights(rows, cols, _DCT16)
    elif t == 5:
        w = _get_quant_weights(rows, cols, _DCT32)
    elif t == 6:
        w = _get_quant_weights(rows, cols, _DCT8X16)
    elif t == 7:
        w = _get_quant_weights(rows, cols, _DCT8X32)
    elif t == 8:
        w = _get_quant_weights(rows, cols, _DCT16X32)
    elif t == 9:
        w48 = _get_quant_weights(4, 8, _DCT4X8)
        w = np.repeat(w48, 2, axis=1)
    elif t == 10:
        afvw = [[3072.0, 3072.0, 256.0, 256.0, 256.0, 414.0, 0.0, 0.0, 0.0],
                [1024.0, 1024.0, 50.0, 50.0, 50.0, 58.0, 0.0, 0.0, 0.0],
                [384.0, 384.0, 12.0, 12.0, 12.0, 22.0, -0.25, -0.25, -0.25]]
        freqs = [0xBAD, 0xBAD, 0.8517778890324296, 5.37778436506804, 0xBAD, 0xBAD, 4.734747904497923,
                 5.449245381693219, 1.6598270267479331, 4.0, 7.275749096817861, 10.423227632456525,
                 2.662932286148962, 7.630657783650829, 8.962388608184032, 12.97166202570235]
        w48 = _get_quant_weights(4, 8, _DCT4X8)
        w44 = _get_quant_weights(4, 4, _DCT4X4)
        lo = _f(0.8517778890324296)
        hi = _f(_f(12.97166202570235) - lo + _f(1e-6))
        for c in range(3):
            bands = _bands([afvw[c][5]] + afvw[c][6:9])
            w[c, 0, 0] = 1.0
            w[c, 1, 0] = afvw[c][0]
            w[c, 0, 1] = afvw[c][1]
            w[c, 2, 0] = afvw[c][2]
            w[c, 0, 2] = afvw[c][3]
            w[c, 2, 2] = afvw[c][4]
            for y in range(4):
                for x in range(4):
                    if x < 2 and y < 2:
                        continue
                    w[c, 2 * y, 2 * x] = _interpolate(_f(freqs[y * 4 + x]) - lo, hi, bands)
            for y in range(4):
                for x in range(8):
                    if x == 0 and y == 0:
                        continue
                    w[c, 2 * y + 1, x] = w48[c, y, x]
            for y in range(4):
                for x in range(4):
                    if x == 0 and y == 0:
                        continue
                    w[c, 2 * y, 2 * x + 1] = w44[c, y, x]
    elif t == 11:
        w = _get_quant_weights(rows, cols, _large(0.9, _BASE_SQ))
    elif t == 12:
        w = _get_quant_weights(rows, cols, _large(0.65, _BASE_RC))
    elif t == 13:
        w = _get_quant_weights(rows, cols, _large(1.8, _BASE_SQ))
    elif t == 14:
        w = _get_quant_weights(rows, cols, _large(1.3, _BASE_RC))
    elif t == 15:
        w = _get_quant_weights(rows, cols, _large(3.6, _BASE_SQ))
    elif t == 16:
        w = _get_quant_weights(rows, cols, _large(2.6, _BASE_RC))
    return (np.float32(1.0) / w.astype(np.float32)).astype(np.float32).reshape(-1)


_TABLE_CACHE = None


def library_dequant_tables():
    global _TABLE_CACHE
    if _TABLE_CACHE is None:
        _TABLE_CACHE = [library_dequant_table(t) for t in range(17)]
    return _TABLE_CACHE


# ----------------------------------------------------------------------------------------
def random_group_tiling(rng, bw, bh, mix, aligned=False):
    """Random valid varblock tiling of a bw x bh (blocks) group.  Returns the u8 transform map
    (bit 7 = top-left block, frame/group.rs:468-473) and the varblock list (bx, by, type) in
    raster order of the top-left block.  Varblocks never leave the group
    (frame/modular/mod.rs:1061-1064).  aligned=False: otherwise unaligned, as the format allows.
    aligned=True: every varblock starts on a multiple of its own size in each direction -- the law SURVEY.md section
    8(d) gives for the synthetic inputs, and what libjxl's encoder emits (its AC-strategy search merges blocks inside
    64x64 tiles at positions aligned to the merged size)."""
    types = list(mix.keys())
    area = np.array([COVERED_X[t] * COVERED_Y[t] for t in types], dtype=np.float64)
    # area weights -> pick probability per placement
    prob = np.array([mix[t] for t in types], dtype=np.float64) / area
    prob /= prob.sum()
    tmap = np.zeros((bh, bw), dtype=np.uint8)
    covered = np.zeros((bh, bw), dtype=bool)
    placed_at = {}
    # The raster-order fill below almost never finds room for a 64..256-pixel varblock (it would have to start on a
    # still-empty 8..32-block square), so those types are seeded first: per type, as many placements as its area
    # share of the group calls for (stochastic rounding), at free positions aligned to the varblock's own size.
    if aligned:
        # every multi-block type is seeded, largest first, at random free positions aligned to its own size, as many
        # as its area share calls for; the raster fill below then only adds 8x8 types.  (Filling in raster order with
        # the alignment as an extra condition starves the large shapes: 81 % DCT8 instead of the mix's 50 %.)
        for t in sorted((t for t in types if COVERED_X[t] * COVERED_Y[t] > 1), key=lambda t: (-COVERED_X[t] * COVERED_Y[t], t)):
            cx, cy = COVERED_X[t], COVERED_Y[t]
            want = mix[t] / sum(mix.values()) * bw * bh / (cx * cy)
            count = int(want) + (1 if rng.random() < want - int(want) else 0)
            cand = [(x, y) for y in range(0, bh - cy + 1, cy) for x in range(0, bw - cx + 1, cx)]
            for i in rng.permutation(len(cand)):
                if count <= 0:
                    break
                x, y = cand[int(i)]
                if covered[y:y + cy, x:x + cx].any():
                    continue
                covered[y:y + cy, x:x + cx] = True
                tmap[y:y + cy, x:x + cx] = t
                tmap[y, x] = t | 0x80
                placed_at[(x, y)] = t
                count -= 1
        small = [t for t in types if COVERED_X[t] * COVERED_Y[t] == 1]
        if small:
            types = small
            prob = np.array([mix[t] for t in types], dtype=np.float64)
            prob /= prob.sum()
    for t in sorted((t for t in types if COVERED_X[t] * COVERED_Y[t] >= 64 and not aligned), key=lambda t: -COVERED_X[t] * COVERED_Y[t]):
        cx, cy = COVERED_X[t], COVERED_Y[t]
        want = mix[t] * bw * bh / (cx * cy)
        count = int(want) + (1 if rng.random() < want - int(want) else 0)
        for _ in range(count):
            free = [(x, y) for y in range(0, bh - cy + 1, cy) for x in range(0, bw - cx + 1, cx)
                    if not covered[y:y + cy, x:x + cx].any()]
            if not free:
                break
            x, y = free[int(rng.integers(len(free)))]
            covered[y:y + cy, x:x + cx] = True
            tmap[y:y + cy, x:x + cx] = t
            tmap[y, x] = t | 0x80
            placed_at[(x, y)] = t
    blocks = []
    for by in range(bh):
        for bx in range(bw):
            if (bx, by) in placed_at:
                blocks.append((bx, by, placed_at[(bx, by)]))
                continue
            if covered[by, bx]:
                continue
            order = rng.choice(len(types), size=min(4, len(types)), replace=False, p=prob)
            placed = False
            for oi in order:
                t = types[oi]
                cx, cy = COVERED_X[t], COVERED_Y[t]
                if aligned and (bx % cx or by % cy):
                    continue
                if bx + cx <= bw and by + cy <= bh and not covered[by:by + cy, bx:bx + cx].any():
                    placed = True
                    break
            if not placed:
                t, cx, cy = 0, 1, 1
            covered[by:by + cy, bx:bx + cx] = True
            tmap[by:by + cy, bx:bx + cx] = t
            tmap[by, bx] = t | 0x80
            blocks.append((bx, by, t))
    return tmap, blocks


@dataclass
class VarDctWorkload:
    xsize: int
    ysize: int
    transform_map: np.ndarray   # u8 [yblocks, xblocks]
    raw_quant: np.ndarray       # i32 [yblocks, xblocks]
    epf_map: np.ndarray         # u8 [yblocks, xblocks]
    ytox: np.ndarray            # i8 [ceil(yb/8), ceil(xb/8)]
    ytob: np.ndarray
    lf_q: list                  # 3 x i32 [yblocks, xblocks], coded order Y, X, B
    coeffs: np.ndarray          # i32 [ngroups, 3, 65536] (X, Y, B)
    tables: list                # 17 float32 arrays
    opts: dict = field(default_factory=dict)

    @property
    def xblocks(self):  # FrameHeader::size_blocks: whole blocks of the coarsest channel
        mh = max(self.opts.get("hshift", (0, 0, 0)))
        return -(-self.xsize // (8 << mh)) << mh

    @property
    def yblocks(self):
        mv = max(self.opts.get("vshift", (0, 0, 0)))
        return -(-self.ysize // (8 << mv)) << mv

    @property
    def xgroups(self):
        return (self.xsize + 255) // 256

    @property
    def ygroups(self):
        return (self.ysize + 255) // 256


def _coeff_block(rng, t, n):
    """n varblocks of type t: i32 [n, 3, cx*cy*64] with a d1-like sparse Laplacian-ish law
    (SURVEY.md section 8d): P(nonzero) = min(1, 0.9 e^{-3.5 rho}) * {0.5, 1, 0.7}; magnitude 1+Geom(1/2)."""
    cx, cy = COVERED_X[t], COVERED_Y[t]
    mn, mx = min(cx, cy) * 8, max(cx, cy) * 8
    r = np.arange(mn, dtype=np.float32)[:, None] / mn
    c = np.arange(mx, dtype=np.float32)[None, :] / mx
    rho = np.sqrt(r * r + c * c)
    p = np.minimum(1.0, 0.9 * np.exp(-3.5 * rho)).astype(np.float32)
    chan = np.array([0.5, 1.0, 0.7], dtype=np.float32)[None, :, None, None]
    u = rng.random((n, 3, mn, mx), dtype=np.float32)
    nz = u < (p[None, None] * chan)
    mag = rng.geometric(0.5, size=(n, 3, mn, mx)).astype(np.int32)
    np.minimum(mag, 4095, out=mag)
    sign = rng.integers(0, 2, size=(n, 3, mn, mx), dtype=np.int32) * 2 - 1
    out = np.where(nz, mag * sign, 0).astype(np.int32)
    # LLF corner (first cx*cy natural-order positions) is overwritten by LLF-from-LF anyway
    out[:, :, : min(cx, cy), : max(cx, cy)] = 0
    return out.reshape(n, 3, mn * mx)


def make_vardct(xsize, ysize, mix=None, seed=0, unique_groups=None, epf_iters=2, gab=True, lf_smoothing=True,
                coeff_scale=1, hshift=(0, 0, 0), vshift=(0, 0, 0), aligned=False):
    """Builds a VarDCT workload.  unique_groups: generate only that many distinct group contents
    and reuse them round-robin (host-side generation time for 8K/16K frames); group *positions*,
    maps and LF are always generated for the whole frame."""
    mix = MIX_D1 if mix is None else mix
    rng = np.random.default_rng([0x4A584C, seed, xsize, ysize])
    mh, mv = max(hshift), max(vshift)
    if mh or mv:
        assert all(COVERED_X[t] == 1 and COVERED_Y[t] == 1 for t in mix), "sub-sampled frames hold 8x8 transforms only"
    xb, yb = -(-xsize // (8 << mh)) << mh, -(-ysize // (8 << mv)) << mv
    xg, yg = (xsize + 255) // 256, (ysize + 255) // 256
    ngroups = xg * yg
    transform_map = np.zeros((yb, xb), dtype=np.uint8)
    raw_quant = np.zeros((yb, xb), dtype=np.int32)
    coeffs = np.zeros((ngroups, 3, 65536), dtype=np.int32)
    cache = {}
    for g in range(ngroups):
        gx, gy = g % xg, g // xg
        bx0, by0 = gx * 32, gy * 32
        bw, bh = min(32, xb - bx0), min(32, yb - by0)
        key = None
        if unique_groups is not None and bw == 32 and bh == 32:
            key = g % unique_groups
        if key is not None and key in cache:
            tmap, rq, slab = cache[key]
        else:
            tmap, blocks = random_group_tiling(rng, bw, bh, mix, aligned)
            rq = np.zeros((bh, bw), dtype=np.int32)
            slab = np.zeros((3, 65536), dtype=np.int32)
            # coefficients: varblocks back to back in raster order of their top-left block
            offs = np.cumsum([0] + [COVERED_X[t] * COVERED_Y[t] * 64 for (_, _, t) in blocks])
            by_type = {}
            for i, (bx, by, t) in enumerate(blocks):
                by_type.setdefault(t, []).append(i)
                q = int(rng.integers(2, 17))
                rq[by:by + COVERED_Y[t], bx:bx + COVERED_X[t]] = q
            for t, idxs in by_type.items():
                data = _coeff_block(rng, t, len(idxs)) * coeff_scale
                n = COVERED_X[t] * COVERED_Y[t] * 64
                for j, i in enumerate(idxs):
                    slab[:, offs[i]:offs[i] + n] = data[j]
            if mh or mv:
                # a decoder never writes coefficients for a channel at blocks the channel does not hold
                # (frame/group.rs:521-524): the slab stays zero there (a group is an even number of blocks)
                for i, (bx, by, t) in enumerate(blocks):
                    for c in range(3):
                        if (bx % (1 << hshift[c])) or (by % (1 << vshift[c])):
                            slab[c, offs[i]:offs[i + 1]] = 0
            if key is not None:
                cache[key] = (tmap, rq, slab)
        transform_map[by0:by0 + bh, bx0:bx0 + bw] = tmap
        raw_quant[by0:by0 + bh, bx0:bx0 + bw] = rq
        coeffs[g] = slab
    epf_map = rng.integers(0, 8, size=(yb, xb), dtype=np.uint8)
    cw, ch = (xb + 7) // 8, (yb + 7) // 8
    ytox = rng.integers(-16, 17, size=(ch, cw)).astype(np.int8)
    ytob = rng.integers(-16, 17, size=(ch, cw)).astype(np.int8)
    # quantised LF such that the dequantised image is smooth-ish
    yy, xx = np.mgrid[0:yb, 0:xb].astype(np.float64)
    Y = 0.5 + 0.25 * np.sin(2 * np.pi * xx / 97) * np.cos(2 * np.pi * yy / 61) + rng.uniform(-0.01, 0.01, (yb, xb))
    X = 0.01 * np.sin(2 * np.pi * xx / 31)
    B = 0.9 * Y + rng.uniform(-0.01, 0.01, (yb, xb))
    inv_quant_lf = 65536.0 / (21845.0 * 16.0)
    fac = [inv_quant_lf / 4096.0, inv_quant_lf / 512.0, inv_quant_lf / 256.0]
    qy = np.round(Y / fac[1]).astype(np.int32)
    qx = np.round(X / fac[0]).astype(np.int32)
    qb = np.round((B - Y) / fac[2]).astype(np.int32)  # B = y*cfl_b(=1) + qb*fac
    return VarDctWorkload(xsize, ysize, transform_map, raw_quant, epf_map, ytox, ytob, [qy, qx, qb], coeffs,
                          library_dequant_tables(),
                          dict(epf_iters=epf_iters, gab=gab, lf_smoothing=lf_smoothing, seed=seed,
                               hshift=tuple(hshift), vshift=tuple(vshift), aligned=bool(aligned)))


def apply_opts(params, wl):
    """Applies the workload's frame options onto a FrameParams-like ctypes struct."""
    params.epf_iters = wl.opts.get("epf_iters", 2)
    params.gab = 1 if wl.opts.get("gab", True) else 0
    params.do_lf_smoothing = 1 if wl.opts.get("lf_smoothing", True) else 0
    for c in range(3):
        params.hshift[c] = wl.opts.get("hshift", (0, 0, 0))[c]
        params.vshift[c] = wl.opts.get("vshift", (0, 0, 0))[c]
    return params


# ----------------------------------------------------------------------------------------
# Modular
def default_squeeze_steps(w, h):
    """Inverse-order list of (horizontal, out_w, out_h) for the in-place part of default_squeeze
    (modular/transforms/squeeze.rs:71-105) on a w x h channel, as the decoder applies them."""
    steps = []
    cw, ch = w, h
    fwd = []
    if cw <= ch and ch > 8:
        fwd.append((False, cw, ch))
        ch = (ch + 1) // 2
    while cw > 8 or ch > 8:
        if cw > 8:
            fwd.append((True, cw, ch))
            cw = (cw + 1) // 2
        if ch > 8:
            fwd.append((False, cw, ch))
            ch = (ch + 1) // 2
    for horizontal, ow, oh in reversed(fwd):
        steps.append((horizontal, ow, oh))
    return steps, (cw, ch)


def make_modular_planes(w, h, seed=0, nchan=3):
    """Residual / average planes for a full default-squeeze chain: returns (base planes, list of
    residual planes per inverse step) with 8-bit-range averages and Laplacian(b=3) residuals."""
    rng = np.random.default_rng([0x4D4F44, seed, w, h])
    steps, (bw, bh) = default_squeeze_steps(w, h)
    base = [rng.integers(0, 256, size=(bh, bw), dtype=np.int32) for _ in range(nchan)]
    residuals = []
    for horizontal, ow, oh in steps:
        rw, rh = (ow // 2, oh) if horizontal else (ow, oh // 2)
        residuals.append([np.round(rng.laplace(0.0, 3.0, size=(rh, rw))).astype(np.int32) for _ in range(nchan)])
    return base, residuals, steps


def to_sparse(group_coeffs):
    """Sparse transport form of one group's dense [3, 65536] i32 slab (include/jxl_hip.h,
    jxlh_submit_group_sparse): (pairs uint32 [n0+n1+n2] little-endian {u16 pos; i16 val}, n[3],
    wide uint32 [k, 2] = (channel * 65536 + pos, value) for values outside i16)."""
    g = np.asarray(group_coeffs).reshape(3, -1)
    runs, n, wide = [], [], []
    for c in range(3):
        pos = np.flatnonzero(g[c])
        val = g[c][pos]
        fits = (val >= -32768) & (val <= 32767)
        p16, v16 = pos[fits].astype(np.uint32), val[fits].astype(np.int64)
        runs.append(p16 | ((v16 & 0xFFFF).astype(np.uint32) << np.uint32(16)))
        n.append(len(p16))
        if (~fits).any():
            wide.append(np.stack([(c * 65536 + pos[~fits]).astype(np.uint32),
                                  val[~fits].astype(np.int32).view(np.uint32)], axis=1))
    pairs = np.concatenate(runs).astype(np.uint32) if runs else np.zeros(0, np.uint32)
    widea = np.concatenate(wide).astype(np.uint32) if wide else np.zeros((0, 2), np.uint32)
    return pairs, np.asarray(n, dtype=np.uint32), widea


def to_sparse8(group_coeffs):
    """3-byte transport form (jxlh_submit_groups_sparse8): (pos uint16 [n0+n1+n2], val int8 [n0+n1+n2], n[3],
    wide uint32 [k, 2] = (channel * 65536 + pos, value) for values outside i8)."""
    g = np.asarray(group_coeffs).reshape(3, -1)
    ps, vs, n, wide = [], [], [], []
    for c in range(3):
        pos = np.flatnonzero(g[c])
        val = g[c][pos]
        fits = (val >= -128) & (val <= 127)
        ps.append(pos[fits].astype(np.uint16))
        vs.append(val[fits].astype(np.int8))
        n.append(int(fits.sum()))
        if (~fits).any():
            wide.append(np.stack([(c * 65536 + pos[~fits]).astype(np.uint32),
                                  val[~fits].astype(np.int32).view(np.uint32)], axis=1))
    widea = np.concatenate(wide).astype(np.uint32) if wide else np.zeros((0, 2), np.uint32)
    return (np.concatenate(ps) if ps else np.zeros(0, np.uint16), np.concatenate(vs) if vs else np.zeros(0, np.int8),
            np.asarray(n, dtype=np.uint32), widea)


def to_sparse4(group_coeffs):
    """2-byte transport form (jxlh_submit_groups_sparse4): (entries uint16 -- (pos & 4095) | (val & 15) << 12, grouped by
    channel and 4096-coefficient segment --, seg_counts uint16 [3, 16], pos8 uint16, val8 int8, n8 uint32 [3] -- the
    3-byte overflow updates whose value is outside [-8, 7] --, wide uint32 [k, 2] for values outside i8)."""
    g = np.asarray(group_coeffs).reshape(3, -1)
    ents, counts, ps, vs, n8, wide = [], np.zeros((3, 16), np.uint16), [], [], [], []
    for c in range(3):
        pos = np.flatnonzero(g[c])
        val = g[c][pos]
        nib = (val >= -8) & (val <= 7)
        byte = ~nib & (val >= -128) & (val <= 127)
        big = ~nib & ~byte
        p4, v4 = pos[nib], val[nib]
        ents.append(((p4 & 4095) | ((v4 & 15) << 12)).astype(np.uint16))  # np.flatnonzero is sorted: segment order
        counts[c] = np.bincount(p4 >> 12, minlength=16).astype(np.uint16)
        ps.append(pos[byte].astype(np.uint16))
        vs.append(val[byte].astype(np.int8))
        n8.append(int(byte.sum()))
        if big.any():
            wide.append(np.stack([(c * 65536 + pos[big]).astype(np.uint32), val[big].astype(np.int32).view(np.uint32)], axis=1))
    widea = np.concatenate(wide).astype(np.uint32) if wide else np.zeros((0, 2), np.uint32)
    return (np.concatenate(ents), counts, np.concatenate(ps), np.concatenate(vs), np.asarray(n8, dtype=np.uint32), widea)


def to_slots(group_coeffs, bits12=False, split=False):
    """Slot-bucketed transport form (jxlh_submit_groups_slots): (entries uint16 -- (pos & 63) | (val & 1023) << 6, ordered by
    channel and 64-coefficient slot --, slot_counts uint8 [3, 1024], n uint32 [3], wide uint32 [k, 2] for values outside
    [-512, 511]).  bits12 (JXLH_GROUP_ENTRIES12): entries = uint8 bytes, 12-bit entries (value in [-32, 31]) packed two
    per three bytes, every channel's run closed to an even count with a zero update in slot 1023.
    split: values outside the range become repeated in-range entries at their position (what jxlh_host_pack_slots does:
    they add up on the device) instead of going to `wide`; the numpy reference of the C packer (same entry order)."""
    g = np.asarray(group_coeffs).reshape(3, -1)
    ents, counts, n, wide = [], np.zeros((3, 1024), np.uint8), [], []
    lo, hi, vmask = (-32, 31, 63) if bits12 else (-512, 511, 1023)
    for c in range(3):
        pos = np.flatnonzero(g[c])
        val = g[c][pos]
        if split and len(val):
            step = np.where(val < 0, -lo, hi).astype(np.int64)
            k = np.maximum(1, (np.abs(val.astype(np.int64)) + step - 1) // step)
            big = k > 96  # (the C packer's kMaxSplit: such a value goes to `wide` whole)
            if big.any():
                wide.append(np.stack([(c * 65536 + pos[big]).astype(np.uint32), val[big].astype(np.int32).view(np.uint32)], axis=1))
                pos, val, step, k = pos[~big], val[~big], step[~big], k[~big]
            pos = np.repeat(pos, k)
            first = np.cumsum(k) - k
            idx = np.arange(len(pos)) - np.repeat(first, k)  # 0 .. k-1 inside a value's pieces
            vv, ss, kk = np.repeat(val.astype(np.int64), k), np.repeat(step, k), np.repeat(k, k)
            full = np.sign(vv) * ss
            val = np.where(idx < kk - 1, full, vv - full * (kk - 1)).astype(np.int32)
        fits = (val >= lo) & (val <= hi)
        p, v = pos[fits], val[fits]
        e = ((p & 63) | ((v & vmask) << 6)).astype(np.uint16)  # np.flatnonzero is sorted: slot order
        cnt = np.bincount(p >> 6, minlength=1024)
        if bits12:
            if len(e) & 1:
                e = np.concatenate([e, np.zeros(1, np.uint16)])  # += 0 at position 0 of the last slot
                cnt[1023] += 1
            e0, e1 = e[0::2].astype(np.uint32), e[1::2].astype(np.uint32)
            b = np.empty((len(e0), 3), np.uint8)
            b[:, 0] = e0 & 255
            b[:, 1] = (e0 >> 8) | ((e1 & 15) << 4)
            b[:, 2] = e1 >> 4
            ents.append(b.reshape(-1))
            assert cnt.max(initial=0) <= 255
            counts[c] = cnt.astype(np.uint8)
            n.append(len(e))
            if (~fits).any():
                wide.append(np.stack([(c * 65536 + pos[~fits]).astype(np.uint32), val[~fits].astype(np.int32).view(np.uint32)], axis=1))
            continue
        ents.append(e)
        assert cnt.max(initial=0) <= 255
        counts[c] = cnt.astype(np.uint8)
        n.append(len(p))
        if (~fits).any():
            wide.append(np.stack([(c * 65536 + pos[~fits]).astype(np.uint32), val[~fits].astype(np.int32).view(np.uint32)], axis=1))
    widea = np.concatenate(wide).astype(np.uint32) if wide else np.zeros((0, 2), np.uint32)
    return np.concatenate(ents), counts, np.asarray(n, dtype=np.uint32), widea
