"""ctypes binding of the CPU oracle (oracle/jxlo.h).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (jxl_rs_amd) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

NUM_TRANSFORMS = 27
NUM_QUANT_TABLES = 17


class FrameParams(C.Structure):
    """Mirror of JxloFrameParams (jxlo.h)."""
    _fields_ = [
        ("xsize", C.c_int32), ("ysize", C.c_int32),
        ("xsize_blocks", C.c_int32), ("ysize_blocks", C.c_int32),
        ("group_dim", C.c_int32),
        ("global_scale", C.c_uint32), ("quant_lf", C.c_uint32),
        ("lf_quant_factors", C.c_float * 3),
        ("quant_biases", C.c_float * 4),
        ("x_qm_scale", C.c_uint32), ("b_qm_scale", C.c_uint32),
        ("color_factor", C.c_uint32),
        ("base_correlation_x", C.c_float), ("base_correlation_b", C.c_float),
        ("ytox_lf", C.c_int32), ("ytob_lf", C.c_int32),
        ("gab", C.c_int32),
        ("gab_w1", C.c_float * 3), ("gab_w2", C.c_float * 3),
        ("epf_iters", C.c_int32),
        ("epf_sharp_lut", C.c_float * 8),
        ("epf_channel_scale", C.c_float * 3),
        ("epf_quant_mul", C.c_float), ("epf_pass0_sigma_scale", C.c_float),
        ("epf_pass2_sigma_scale", C.c_float), ("epf_border_sad_mul", C.c_float),
        ("do_lf_smoothing", C.c_int32),
        ("hshift", C.c_int32 * 3), ("vshift", C.c_int32 * 3),
    ]


def build(force=False):
    """Compile both oracle builds in place (make is incremental)."""
    args = ["make", "-C", _HERE, "-s"]
    if force:
        args.append("-B")
    subprocess.run(args, check=True)


def _ptr(a, ty):
    return a.ctypes.data_as(C.POINTER(ty))


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


class Oracle:
    def __init__(self, fused=True):
        name = "libjxlo_fused.so" if fused else "libjxlo_unfused.so"
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            build()
        self.lib = L = C.CDLL(path)
        self.fused = fused
        assert bool(L.jxlo_is_fused()) == fused
        fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)
        dp = C.POINTER(C.c_double)
        L.jxlo_xyb_params.argtypes = [fp, fp, C.c_float, fp]
        L.jxlo_xyb_params.restype = None
        L.jxlo_xyb_to_linear.argtypes = [fp, fp, fp, fp, C.c_size_t]
        L.jxlo_xyb_to_linear.restype = None
        L.jxlo_linear_to_srgb.argtypes = [fp, C.c_size_t]
        L.jxlo_linear_to_srgb.restype = None
        L.jxlo_f32_to_u8.argtypes = [C.c_float, C.c_size_t, C.c_size_t, C.c_int, C.c_int]
        L.jxlo_f32_to_u8.restype = C.c_uint8
        L.jxlo_xyb_to_rgb8.argtypes = [fp, fp, fp, fp, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_uint8),
                                       C.c_size_t, C.c_int]
        L.jxlo_xyb_to_rgb8.restype = None
        L.jxlo_xyb_to_rgb16.argtypes = [fp, fp, fp, fp, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_uint16),
                                        C.c_size_t, C.c_int]
        L.jxlo_xyb_to_rgb16.restype = None
        L.jxlo_f32_to_u16.argtypes = [C.c_float, C.c_int]
        L.jxlo_f32_to_u16.restype = C.c_uint16
        L.jxlo_expand_sparse.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                         C.c_uint32, C.POINTER(C.c_int32)]
        L.jxlo_expand_sparse.restype = None
        L.jxlo_idct1d.argtypes = [fp, C.c_int, C.c_int]
        L.jxlo_rdct1d.argtypes = [fp, C.c_int, C.c_int]
        L.jxlo_idct2d.argtypes = [fp, C.c_int, C.c_int]
        L.jxlo_rdct2d.argtypes = [fp, C.c_int, C.c_int, fp]
        L.jxlo_transform_to_pixels.argtypes = [C.c_int, fp, fp]
        L.jxlo_slow_idct2d.argtypes = [dp, C.c_int, C.c_int, dp]
        L.jxlo_slow_rdct2d.argtypes = [dp, C.c_int, C.c_int, dp]
        L.jxlo_slow_idct1d.argtypes = [dp, C.c_int, dp]
        L.jxlo_slow_dct1d.argtypes = [dp, C.c_int, dp]
        L.jxlo_idct_weights.restype = fp
        L.jxlo_rdct_scales.restype = fp
        L.jxlo_library_dequant_table.argtypes = [C.c_int, fp]
        L.jxlo_natural_coeff_order.argtypes = [C.c_int, C.POINTER(C.c_uint32)]
        L.jxlo_default_frame_params.argtypes = [C.POINTER(FrameParams), C.c_int, C.c_int]
        L.jxlo_dequant_lf.argtypes = [C.POINTER(FrameParams), ip, ip, ip, C.c_float, C.c_size_t, fp, fp, fp]
        pf3 = C.POINTER(fp)
        L.jxlo_adaptive_lf_smoothing.argtypes = [C.POINTER(FrameParams), pf3, C.c_int, C.c_int, pf3]
        L.jxlo_sigma_map.argtypes = [C.POINTER(FrameParams), ip, C.POINTER(C.c_uint8), fp]
        L.jxlo_decode_group.argtypes = [C.POINTER(FrameParams), C.c_int, ip, C.POINTER(C.c_uint8), ip,
                                        C.POINTER(C.c_int8), C.POINTER(C.c_int8), pf3, pf3, pf3, C.c_size_t]
        L.jxlo_gaborish.argtypes = [fp, C.c_int, C.c_int, C.c_size_t, C.c_float, C.c_float, fp]
        L.jxlo_epf.argtypes = [C.c_int, C.POINTER(FrameParams), pf3, C.c_int, C.c_int, C.c_size_t, fp,
                               C.c_size_t, pf3]
        L.jxlo_gaborish_rows.argtypes = [fp, C.c_int, C.c_int, C.c_size_t, C.c_float, C.c_float, fp, C.c_int, C.c_int]
        L.jxlo_epf_rows.argtypes = [C.c_int, C.POINTER(FrameParams), pf3, C.c_int, C.c_int, C.c_size_t, fp,
                                    C.c_size_t, pf3, C.c_int, C.c_int]
        L.jxlo_vardct_frame.argtypes = [C.POINTER(FrameParams), ip, C.POINTER(C.c_uint8), ip,
                                        C.POINTER(C.c_uint8), C.POINTER(C.c_int8), C.POINTER(C.c_int8),
                                        pf3, pf3, pf3, pf3, C.c_size_t, C.c_int]
        L.jxlo_dequant_lf_channel.argtypes = [C.POINTER(FrameParams), C.c_int, ip, C.c_float, C.c_size_t, fp]
        for fn in (L.jxlo_chroma_upsample_h, L.jxlo_chroma_upsample_v):
            fn.argtypes = [fp, C.c_int, C.c_int, C.c_size_t, fp, C.c_size_t]
            fn.restype = None
        L.jxlo_upsample_kernels.argtypes = [C.c_int, fp, fp]
        L.jxlo_upsample.argtypes = [C.c_int, fp, fp, C.c_int, C.c_int, C.c_size_t, fp, C.c_size_t]
        u64p = C.POINTER(C.c_uint64)
        L.jxlo_xorshift_seed.argtypes = [C.c_uint64, u64p]
        L.jxlo_xorshift_seeds.argtypes = [C.c_uint32] * 4 + [u64p]
        L.jxlo_xorshift_fill.argtypes = [u64p, u64p]
        L.jxlo_noise_generate.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, pf3, C.c_size_t]
        L.jxlo_noise_convolve.argtypes = [fp, C.c_int, C.c_int, C.c_size_t, fp, C.c_size_t]
        L.jxlo_noise_strength.argtypes = [fp, C.c_float]
        L.jxlo_noise_strength.restype = C.c_float
        L.jxlo_noise_add.argtypes = [fp, C.c_float, C.c_float, fp, fp, fp, fp, fp, fp, C.c_size_t]
        L.jxlo_fast_powf.argtypes = [C.c_float, C.c_float, C.c_int]
        L.jxlo_fast_powf.restype = C.c_float
        L.jxlo_from_linear.argtypes = [C.c_int, C.c_float, fp, fp, fp, fp, C.c_size_t]
        L.jxlo_xyb_to_rgb_tf.argtypes = [fp, C.c_int, C.c_float, fp, fp, fp, fp, C.c_size_t, C.c_size_t, C.c_size_t,
                                         C.c_int, C.c_void_p, C.c_size_t, C.c_int]
        L.jxlo_ycbcr_to_rgb.argtypes = [fp, fp, fp, C.c_size_t]
        L.jxlo_ycbcr_to_rgb8.argtypes = [fp, fp, fp, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_uint8),
                                         C.c_size_t, C.c_int]
        L.jxlo_ycbcr_to_rgb16.argtypes = [fp, fp, fp, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_uint16),
                                          C.c_size_t, C.c_int]
        L.jxlo_rct.argtypes = [ip, ip, ip, C.c_size_t, C.c_int, C.c_int]
        L.jxlo_palette.argtypes = [ip, C.c_size_t, ip, C.c_int, C.c_size_t, C.c_int, C.c_int, ip]
        L.jxlo_palette_delta.argtypes = [ip, C.c_int, C.c_int, ip, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_int, ip]
        L.jxlo_i32_to_u8.argtypes = [ip, C.c_size_t, C.c_int32, C.c_int32, C.POINTER(C.c_uint8)]
        L.jxlo_modular_to_f32.argtypes = [ip, C.c_size_t, C.c_int, fp]
        L.jxlo_float_samples_to_f32.argtypes = [ip, C.c_size_t, C.c_uint32, C.c_uint32, fp]
        L.jxlo_modular_xyb_to_f32.argtypes = [ip, ip, ip, C.c_size_t, fp, fp, fp, fp]
        u32p = C.POINTER(C.c_uint32)
        L.jxlo_wp_new.argtypes = [u32p, C.c_int]
        L.jxlo_wp_new.restype = C.c_void_p
        L.jxlo_wp_free.argtypes = [C.c_void_p]
        L.jxlo_wp_free.restype = None
        L.jxlo_wp_predict.argtypes = [C.c_void_p, C.c_int, C.c_int, ip, ip]
        L.jxlo_wp_predict.restype = C.c_int64
        L.jxlo_wp_update.argtypes = [C.c_void_p, C.c_int32, C.c_int, C.c_int]
        L.jxlo_wp_update.restype = None
        L.jxlo_palette_delta_wp.argtypes = [ip, C.c_int, C.c_int, ip, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int,
                                            u32p, ip]
        L.jxlo_palette_delta_wp.restype = None
        L.jxlo_unsqueeze_h.argtypes = [ip, C.c_size_t, ip, C.c_size_t, C.c_int, C.c_int, ip, C.c_size_t]
        L.jxlo_unsqueeze_v.argtypes = [ip, C.c_size_t, ip, C.c_size_t, C.c_int, C.c_int, ip, C.c_size_t]
        L.jxlo_unsqueeze_h_simd.argtypes = L.jxlo_unsqueeze_h.argtypes
        L.jxlo_unsqueeze_v_simd.argtypes = L.jxlo_unsqueeze_v.argtypes
        L.jxlo_smooth_convolve_2d.argtypes = [fp, C.c_int, ip]
        L.jxlo_smooth_convolve_1d.argtypes = [fp, C.c_int, ip]
        L.jxlo_smooth_unsqueeze.argtypes = [C.c_int, ip, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, ip,
                                            C.c_size_t, C.c_int, C.c_int, C.c_int]
        for f in (L.jxlo_smooth_convolve_2d, L.jxlo_smooth_convolve_1d, L.jxlo_smooth_unsqueeze):
            f.restype = None
        L.jxlo_smooth_tendency.argtypes = [C.c_int64] * 3
        L.jxlo_smooth_tendency.restype = C.c_int64
        L.jxlo_smooth_tendency_i32.argtypes = [C.c_int32] * 3
        L.jxlo_smooth_tendency_i32.restype = C.c_int32
        self.covered_x = [L.jxlo_covered_blocks_x(t) for t in range(NUM_TRANSFORMS)]
        self.covered_y = [L.jxlo_covered_blocks_y(t) for t in range(NUM_TRANSFORMS)]
        self.table_for_type = [L.jxlo_quant_table_for_type(t) for t in range(NUM_TRANSFORMS)]
        self.table_size = [L.jxlo_quant_table_size(t) for t in range(NUM_QUANT_TABLES)]
        self._tables = None

    # ---- transforms ----
    def idct1d(self, x):
        a = _f32(x).copy()
        self.lib.jxlo_idct1d(_ptr(a, C.c_float), a.size, 1)
        return a

    def rdct1d(self, x):
        a = _f32(x).copy()
        self.lib.jxlo_rdct1d(_ptr(a, C.c_float), a.size, 1)
        return a

    def idct2d(self, coeffs, rows, cols):
        a = _f32(coeffs).reshape(-1).copy()
        assert a.size == rows * cols
        self.lib.jxlo_idct2d(_ptr(a, C.c_float), rows, cols)
        return a.reshape(rows, cols)

    def rdct2d(self, lf):
        lf = _f32(lf)
        rows, cols = lf.shape
        mn, mx = min(rows, cols), max(rows, cols)
        out = np.zeros(mn * 8 * mx + mx, dtype=np.float32)
        out = np.zeros((mn, 8 * mx), dtype=np.float32)
        tmp = lf.copy()
        self.lib.jxlo_rdct2d(_ptr(tmp, C.c_float), rows, cols, _ptr(out, C.c_float))
        return out[:, :mx].copy()

    def transform_to_pixels(self, ttype, lf, coeffs):
        cx, cy = self.covered_x[ttype], self.covered_y[ttype]
        lfb = _f32(lf).reshape(-1).copy()
        assert lfb.size == cx * cy
        buf = _f32(coeffs).reshape(-1).copy()
        assert buf.size == cx * cy * 64
        self.lib.jxlo_transform_to_pixels(ttype, _ptr(lfb, C.c_float), _ptr(buf, C.c_float))
        return buf.reshape(cy * 8, cx * 8)

    def slow_idct2d(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        rows, cols = x.shape
        out = np.zeros((rows, cols), dtype=np.float64)
        self.lib.jxlo_slow_idct2d(_ptr(x, C.c_double), rows, cols, _ptr(out, C.c_double))
        return out

    def slow_rdct2d(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        rows, cols = x.shape
        out = np.zeros((min(rows, cols), max(rows, cols)), dtype=np.float64)
        self.lib.jxlo_slow_rdct2d(_ptr(x, C.c_double), rows, cols, _ptr(out, C.c_double))
        return out

    def slow_idct1d(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.zeros_like(x)
        self.lib.jxlo_slow_idct1d(_ptr(x, C.c_double), x.size, _ptr(out, C.c_double))
        return out

    def slow_dct1d(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.zeros_like(x)
        self.lib.jxlo_slow_dct1d(_ptr(x, C.c_double), x.size, _ptr(out, C.c_double))
        return out

    def idct_weights(self, n):
        p = self.lib.jxlo_idct_weights(n)
        return np.ctypeslib.as_array(p, shape=(n // 2,)).copy()

    def rdct_scales(self, n):
        p = self.lib.jxlo_rdct_scales(n)
        return np.ctypeslib.as_array(p, shape=(n,)).copy()

    # ---- tables ----
    def library_dequant_table(self, t):
        out = np.zeros(3 * self.table_size[t], dtype=np.float32)
        rc = self.lib.jxlo_library_dequant_table(t, _ptr(out, C.c_float))
        assert rc == 0
        return out

    def library_dequant_tables(self):
        if self._tables is None:
            self._tables = [self.library_dequant_table(t) for t in range(NUM_QUANT_TABLES)]
        return self._tables

    def natural_coeff_order(self, ttype):
        n = self.covered_x[ttype] * self.covered_y[ttype] * 64
        out = np.zeros(n, dtype=np.uint32)
        self.lib.jxlo_natural_coeff_order(ttype, _ptr(out, C.c_uint32))
        return out

    # ---- frame level ----
    def default_params(self, xsize, ysize):
        p = FrameParams()
        self.lib.jxlo_default_frame_params(C.byref(p), xsize, ysize)
        return p

    @staticmethod
    def _p3(arrs, ty=C.c_float):
        T = C.POINTER(ty) * len(arrs)
        return T(*[_ptr(a, ty) for a in arrs])

    def dequant_lf(self, p, qy, qx, qb, mul=1.0):
        qy, qx, qb = [np.ascontiguousarray(a, dtype=np.int32) for a in (qy, qx, qb)]
        out = [np.zeros(qy.shape, dtype=np.float32) for _ in range(3)]
        self.lib.jxlo_dequant_lf(C.byref(p), _ptr(qy, C.c_int32), _ptr(qx, C.c_int32), _ptr(qb, C.c_int32),
                                 C.c_float(mul), qy.size, *[_ptr(o, C.c_float) for o in out])
        return out  # X, Y, B

    def adaptive_lf_smoothing(self, p, lf):
        lf = [_f32(a) for a in lf]
        h, w = lf[0].shape
        out = [np.zeros((h, w), dtype=np.float32) for _ in range(3)]
        self.lib.jxlo_adaptive_lf_smoothing(C.byref(p), self._p3(lf), w, h, self._p3(out))
        return out

    def sigma_map(self, p, raw_quant, epf_map):
        rq = np.ascontiguousarray(raw_quant, dtype=np.int32)
        em = np.ascontiguousarray(epf_map, dtype=np.uint8)
        out = np.zeros(rq.shape, dtype=np.float32)
        self.lib.jxlo_sigma_map(C.byref(p), _ptr(rq, C.c_int32), _ptr(em, C.c_uint8), _ptr(out, C.c_float))
        return out

    def decode_group(self, p, group, coeffs, transform_map, raw_quant, ytox, ytob, lf, tables, planes):
        """planes: list of 3 float32 2-D arrays (modified in place), contiguous rows."""
        stride = planes[0].strides[0] // 4
        tm = np.ascontiguousarray(transform_map, dtype=np.uint8)
        rq = np.ascontiguousarray(raw_quant, dtype=np.int32)
        yx = np.ascontiguousarray(ytox, dtype=np.int8)
        yb = np.ascontiguousarray(ytob, dtype=np.int8)
        co = np.ascontiguousarray(coeffs, dtype=np.int32)
        lf = [_f32(a) for a in lf]
        self.lib.jxlo_decode_group(C.byref(p), group, _ptr(co, C.c_int32), _ptr(tm, C.c_uint8),
                                   _ptr(rq, C.c_int32), _ptr(yx, C.c_int8), _ptr(yb, C.c_int8),
                                   self._p3(lf), self._p3(tables), self._p3(planes), stride)

    def gaborish(self, plane, w1, w2, w=None, h=None):
        plane = _f32(plane)
        H, S = plane.shape
        w = S if w is None else w
        h = H if h is None else h
        out = np.zeros_like(plane)
        self.lib.jxlo_gaborish(_ptr(plane, C.c_float), w, h, S, C.c_float(w1), C.c_float(w2),
                               _ptr(out, C.c_float))
        return out

    def epf(self, stage, p, planes, inv_sigma, w=None, h=None):
        planes = [_f32(a) for a in planes]
        H, S = planes[0].shape
        w = S if w is None else w
        h = H if h is None else h
        sig = _f32(inv_sigma)
        out = [np.zeros_like(planes[0]) for _ in range(3)]
        self.lib.jxlo_epf(stage, C.byref(p), self._p3(planes), w, h, S, _ptr(sig, C.c_float),
                          sig.shape[1], self._p3(out))
        return out

    def vardct_band(self, p, coeffs, transform_map, raw_quant, epf_map, ytox, ytob, lf, tables, row0, row1,
                    exchange=None):
        """Band of group rows [row0, row1) the way a rank computes it.  exchange is None: K1 on the band plus
        one halo group row on each side (mirrors jxlh_frame_run(row0, row1)).  exchange = callable(planes):
        K1 on exactly the band; the callable then fills the block row above / below the band from the
        neighbour ranks (mirrors jxlh_frame_run_sharded).  Every stage then runs on the band's rows extended
        by the later stages' borders.  lf must already be smoothed.  Returns 3 planes holding valid data on
        the band's pixel rows."""
        bw, bh = p.xsize_blocks, p.ysize_blocks
        stride = bw * 8
        xg = (p.xsize + 255) // 256
        yg = (p.ysize + 255) // 256
        cur = [np.zeros((bh * 8, stride), dtype=np.float32) for _ in range(3)]
        oth = [np.zeros((bh * 8, stride), dtype=np.float32) for _ in range(3)]
        co = np.ascontiguousarray(coeffs, dtype=np.int32)
        gr0, gr1 = (row0, row1) if exchange is not None else (max(row0 - 1, 0), min(row1 + 1, yg))
        for g in range(gr0 * xg, gr1 * xg):
            self.decode_group(p, g, co[g], transform_map, raw_quant, ytox, ytob, lf, tables, cur)
        if exchange is not None:
            exchange(cur)
        sigma = self.sigma_map(p, raw_quant, epf_map)
        stages, borders = [], []
        if p.gab:
            stages.append(-1); borders.append(1)
        if p.epf_iters >= 3:
            stages.append(0); borders.append(3)
        if p.epf_iters >= 1:
            stages.append(1); borders.append(2)
        if p.epf_iters >= 2:
            stages.append(2); borders.append(1)
        y_lo, y_hi = row0 * 256, min(row1 * 256, p.ysize)
        for i, st in enumerate(stages):
            later = sum(borders[i + 1:])
            y0, y1 = max(0, y_lo - later), min(p.ysize, y_hi + later)
            if st < 0:
                for c in range(3):
                    self.lib.jxlo_gaborish_rows(_ptr(cur[c], C.c_float), p.xsize, p.ysize, stride,
                                                C.c_float(p.gab_w1[c]), C.c_float(p.gab_w2[c]),
                                                _ptr(oth[c], C.c_float), y0, y1)
            else:
                self.lib.jxlo_epf_rows(st, C.byref(p), self._p3(cur), p.xsize, p.ysize, stride,
                                       _ptr(sigma, C.c_float), sigma.shape[1], self._p3(oth), y0, y1)
            cur, oth = oth, cur
        return cur

    def vardct_frame(self, p, coeffs, transform_map, raw_quant, epf_map, ytox, ytob, lf, tables,
                     num_threads=1, buffers=None):
        """Runs the whole chain; returns 3 planes (padded to whole blocks).  buffers = (planes, tmp) of an earlier
        call are reused when given (timing loops: no allocation / first-touch page faults in the timed region)."""
        bw, bh = p.xsize_blocks, p.ysize_blocks
        stride = bw * 8
        if buffers is not None:
            planes, tmp = buffers
        else:
            planes = [np.zeros((bh * 8, stride), dtype=np.float32) for _ in range(3)]
            tmp = [np.zeros((bh * 8, stride), dtype=np.float32) for _ in range(3)]
        self.last_buffers = (planes, tmp)
        lf = [_f32(a).copy() for a in lf]
        tm = np.ascontiguousarray(transform_map, dtype=np.uint8)
        rq = np.ascontiguousarray(raw_quant, dtype=np.int32)
        em = np.ascontiguousarray(epf_map, dtype=np.uint8)
        yx = np.ascontiguousarray(ytox, dtype=np.int8)
        yb = np.ascontiguousarray(ytob, dtype=np.int8)
        co = np.ascontiguousarray(coeffs, dtype=np.int32)
        self.lib.jxlo_vardct_frame(C.byref(p), _ptr(co, C.c_int32), _ptr(tm, C.c_uint8), _ptr(rq, C.c_int32),
                                   _ptr(em, C.c_uint8), _ptr(yx, C.c_int8), _ptr(yb, C.c_int8),
                                   self._p3(lf), self._p3(tables), self._p3(planes), self._p3(tmp),
                                   stride, num_threads)
        return planes, lf

    # ---- output stages (XYB -> linear -> sRGB -> u8) ----
    def xyb_params(self, inverse_matrix, opsin_biases, intensity_target=255.0):
        m = np.ascontiguousarray(inverse_matrix, dtype=np.float32)
        b = np.ascontiguousarray(opsin_biases, dtype=np.float32)
        out = np.zeros(16, dtype=np.float32)   # JxloXybParams: mat[9], bias_cbrt[3], scaled_bias[3], intensity_scale
        self.lib.jxlo_xyb_params(_ptr(m, C.c_float), _ptr(b, C.c_float), C.c_float(intensity_target),
                                 _ptr(out, C.c_float))
        return out

    def xyb_to_linear(self, params, x, y, b):
        rows = [np.ascontiguousarray(a, dtype=np.float32).copy().reshape(-1) for a in (x, y, b)]
        p = np.ascontiguousarray(params, dtype=np.float32)
        self.lib.jxlo_xyb_to_linear(_ptr(p, C.c_float), _ptr(rows[0], C.c_float), _ptr(rows[1], C.c_float),
                                    _ptr(rows[2], C.c_float), rows[0].size)
        return rows

    def linear_to_srgb(self, v):
        v = np.ascontiguousarray(v, dtype=np.float32).copy()
        self.lib.jxlo_linear_to_srgb(_ptr(v.reshape(-1), C.c_float), v.size)
        return v

    def f32_to_u8(self, v, x, y, channel, bit_depth=8):
        return int(self.lib.jxlo_f32_to_u8(C.c_float(v), x, y, channel, bit_depth))

    def xyb_to_rgb8(self, params, planes, w, h, channels=3):
        pl = [np.ascontiguousarray(a, dtype=np.float32) for a in planes]
        stride = pl[0].shape[1]
        p = np.ascontiguousarray(params, dtype=np.float32)
        out = np.zeros((h, w, channels), dtype=np.uint8)
        self.lib.jxlo_xyb_to_rgb8(_ptr(p, C.c_float), _ptr(pl[0], C.c_float), _ptr(pl[1], C.c_float),
                                  _ptr(pl[2], C.c_float), w, h, stride, _ptr(out, C.c_uint8), w * channels, channels)
        return out

    def xyb_to_rgb16(self, params, planes, w, h, channels=3):
        pl = [np.ascontiguousarray(a, dtype=np.float32) for a in planes]
        stride = pl[0].shape[1]
        p = np.ascontiguousarray(params, dtype=np.float32)
        out = np.zeros((h, w, channels), dtype=np.uint16)
        self.lib.jxlo_xyb_to_rgb16(_ptr(p, C.c_float), _ptr(pl[0], C.c_float), _ptr(pl[1], C.c_float),
                                   _ptr(pl[2], C.c_float), w, h, stride, _ptr(out, C.c_uint16), w * channels, channels)
        return out

    # ---- sub-sampled (JPEG-recompression) frames ----
    def dequant_lf_channel(self, p, c, q, mul=1.0):
        q = np.ascontiguousarray(q, dtype=np.int32)
        out = np.zeros(q.shape, dtype=np.float32)
        self.lib.jxlo_dequant_lf_channel(C.byref(p), c, _ptr(q, C.c_int32), C.c_float(mul), q.size, _ptr(out, C.c_float))
        return out

    def chroma_upsample(self, plane, horizontal):
        plane = _f32(plane)
        hs, ws = plane.shape
        out = np.zeros((hs, 2 * ws) if horizontal else (2 * hs, ws), dtype=np.float32)
        fn = self.lib.jxlo_chroma_upsample_h if horizontal else self.lib.jxlo_chroma_upsample_v
        fn(_ptr(plane, C.c_float), ws, hs, ws, _ptr(out, C.c_float), out.shape[1])
        return out

    def upsample_kernels(self, n, weights=None):
        flat = np.zeros((n, n, 5, 5), dtype=np.float32)
        wp = None if weights is None else _ptr(np.ascontiguousarray(weights, dtype=np.float32), C.c_float)
        self.lib.jxlo_upsample_kernels(n, wp, _ptr(flat, C.c_float))
        return flat

    def upsample(self, n, plane, weights=None):
        plane = _f32(plane)
        h, w = plane.shape
        out = np.zeros((h * n, w * n), dtype=np.float32)
        wts = None if weights is None else np.ascontiguousarray(weights, dtype=np.float32)
        self.lib.jxlo_upsample(n, None if wts is None else _ptr(wts, C.c_float), _ptr(plane, C.c_float), w, h, w,
                               _ptr(out, C.c_float), w * n)
        return out

    # ---- noise synthesis ----
    def xorshift_golden(self, seed, nvec):
        st = np.zeros(16, dtype=np.uint64)
        self.lib.jxlo_xorshift_seed(seed, _ptr(st, C.c_uint64))
        out = np.zeros((nvec, 8), dtype=np.uint64)
        for i in range(nvec):
            self.lib.jxlo_xorshift_fill(_ptr(st, C.c_uint64), _ptr(out[i], C.c_uint64))
        return out

    def noise_generate(self, visible, nonvisible, w, h, group_dim=256):
        out = [np.zeros((h, w), dtype=np.float32) for _ in range(3)]
        self.lib.jxlo_noise_generate(visible, nonvisible, w, h, group_dim, self._p3(out), w)
        return out

    def noise_convolve(self, plane):
        plane = _f32(plane)
        h, w = plane.shape
        out = np.zeros_like(plane)
        self.lib.jxlo_noise_convolve(_ptr(plane, C.c_float), w, h, w, _ptr(out, C.c_float), w)
        return out

    def noise_strength(self, lut, v):
        lut = np.ascontiguousarray(lut, dtype=np.float32)
        return float(self.lib.jxlo_noise_strength(_ptr(lut, C.c_float), C.c_float(v)))

    def noise_add(self, lut, ytox, ytob, planes, rnd):
        lut = np.ascontiguousarray(lut, dtype=np.float32)
        pl = [_f32(a).copy() for a in planes]
        rn = [_f32(a) for a in rnd]
        self.lib.jxlo_noise_add(_ptr(lut, C.c_float), C.c_float(ytox), C.c_float(ytob), *[_ptr(a, C.c_float) for a in pl],
                                *[_ptr(a, C.c_float) for a in rn], pl[0].size)
        return pl

    TF_KINDS = {"linear": 0, "srgb": 1, "bt709": 2, "pq": 3, "hlg": 4, "gamma": 5}

    def from_linear(self, kind, rgb, param=0.0, lum=(0.2627, 0.678, 0.0593)):
        a = [_f32(v).copy() for v in rgb]
        lum = np.ascontiguousarray(lum, dtype=np.float32)
        self.lib.jxlo_from_linear(self.TF_KINDS[kind], C.c_float(param), _ptr(lum, C.c_float),
                                  *[_ptr(v, C.c_float) for v in a], a[0].size)
        return a

    def fast_powf(self, base, e, simd=True):
        return float(self.lib.jxlo_fast_powf(C.c_float(base), C.c_float(e), 1 if simd else 0))

    def xyb_to_rgb_tf(self, params, kind, planes, w, h, channels=3, bits=8, param=0.0, lum=(0.2627, 0.678, 0.0593)):
        pl = [np.ascontiguousarray(a, dtype=np.float32) for a in planes]
        p = np.ascontiguousarray(params, dtype=np.float32)
        lum = np.ascontiguousarray(lum, dtype=np.float32)
        out = np.zeros((h, w, channels), dtype=np.uint8 if bits == 8 else np.uint16)
        self.lib.jxlo_xyb_to_rgb_tf(_ptr(p, C.c_float), self.TF_KINDS[kind], C.c_float(param), _ptr(lum, C.c_float),
                                    _ptr(pl[0], C.c_float), _ptr(pl[1], C.c_float), _ptr(pl[2], C.c_float), w, h,
                                    pl[0].shape[1], bits, out.ctypes.data, w * channels, channels)
        return out

    def ycbcr_to_rgb(self, cb, y, cr):
        a = [_f32(v).copy() for v in (cb, y, cr)]
        self.lib.jxlo_ycbcr_to_rgb(_ptr(a[0], C.c_float), _ptr(a[1], C.c_float), _ptr(a[2], C.c_float), a[0].size)
        return a  # R, G, B

    def ycbcr_to_rgb8(self, planes, w, h, channels=3):
        pl = [np.ascontiguousarray(a, dtype=np.float32) for a in planes]
        out = np.zeros((h, w, channels), dtype=np.uint8)
        self.lib.jxlo_ycbcr_to_rgb8(_ptr(pl[0], C.c_float), _ptr(pl[1], C.c_float), _ptr(pl[2], C.c_float), w, h,
                                    pl[0].shape[1], _ptr(out, C.c_uint8), w * channels, channels)
        return out

    def ycbcr_to_rgb16(self, planes, w, h, channels=3):
        pl = [np.ascontiguousarray(a, dtype=np.float32) for a in planes]
        out = np.zeros((h, w, channels), dtype=np.uint16)
        self.lib.jxlo_ycbcr_to_rgb16(_ptr(pl[0], C.c_float), _ptr(pl[1], C.c_float), _ptr(pl[2], C.c_float), w, h,
                                     pl[0].shape[1], _ptr(out, C.c_uint16), w * channels, channels)
        return out

    # ---- sparse coefficient transport ----
    def expand_sparse(self, pairs, n, wide=None):
        pairs = np.ascontiguousarray(pairs, dtype=np.uint32)
        n = np.ascontiguousarray(n, dtype=np.uint32)
        wide = np.zeros((0, 2), np.uint32) if wide is None else np.ascontiguousarray(wide, dtype=np.uint32)
        slab = np.zeros((3, 65536), dtype=np.int32)
        self.lib.jxlo_expand_sparse(_ptr(pairs, C.c_uint32), _ptr(n, C.c_uint32), _ptr(wide, C.c_uint32),
                                    len(wide), _ptr(slab, C.c_int32))
        return slab

    # ---- modular ----
    def palette_delta(self, index, palette, num_colors, num_deltas, bit_depth, predictor):
        """index [h, w]; palette [nb_channels, num_colors + num_deltas] -> [nb_channels, h, w]"""
        index = np.ascontiguousarray(index, dtype=np.int32)
        palette = np.ascontiguousarray(palette, dtype=np.int32)
        h, w = index.shape
        nb = palette.shape[0]
        out = np.zeros((nb, h, w), dtype=np.int32)
        self.lib.jxlo_palette_delta(_ptr(index, C.c_int32), w, h, _ptr(palette, C.c_int32), num_colors, num_deltas,
                                    palette.shape[1], nb, bit_depth, predictor, _ptr(out, C.c_int32))
        return out

    def i32_to_u8(self, plane, multiplier, maxv):
        a = np.ascontiguousarray(plane, dtype=np.int32)
        out = np.zeros(a.shape, dtype=np.uint8)
        self.lib.jxlo_i32_to_u8(_ptr(a, C.c_int32), a.size, multiplier, maxv, _ptr(out, C.c_uint8))
        return out

    def modular_to_f32(self, plane, bits, exp_bits=0):
        """ConvertModularToF32Stage: integer samples (exp_bits 0) or `bits`-bit floats with exp_bits exponent bits"""
        a = np.ascontiguousarray(plane, dtype=np.int32)
        out = np.zeros(a.shape, dtype=np.float32)
        if exp_bits:
            self.lib.jxlo_float_samples_to_f32(_ptr(a, C.c_int32), a.size, bits, exp_bits, _ptr(out, C.c_float))
        else:
            self.lib.jxlo_modular_to_f32(_ptr(a, C.c_int32), a.size, bits, _ptr(out, C.c_float))
        return out

    def modular_xyb_to_f32(self, y, x, b, scale):
        y, x, b = [np.ascontiguousarray(v, dtype=np.int32) for v in (y, x, b)]
        sc = np.ascontiguousarray(scale, dtype=np.float32)
        out = [np.zeros(y.shape, dtype=np.float32) for _ in range(3)]
        self.lib.jxlo_modular_xyb_to_f32(_ptr(y, C.c_int32), _ptr(x, C.c_int32), _ptr(b, C.c_int32), y.size,
                                         _ptr(sc, C.c_float), *[_ptr(o, C.c_float) for o in out])
        return out  # X, Y, B

    def rct(self, planes, op, perm):
        ps = [np.ascontiguousarray(a, dtype=np.int32).copy() for a in planes]
        self.lib.jxlo_rct(_ptr(ps[0], C.c_int32), _ptr(ps[1], C.c_int32), _ptr(ps[2], C.c_int32),
                          ps[0].size, op, perm)
        return ps

    def palette(self, index, palette, num_colors, nb_channels, bit_depth):
        idx = np.ascontiguousarray(index, dtype=np.int32)
        pal = np.ascontiguousarray(palette, dtype=np.int32)
        out = np.zeros((nb_channels,) + idx.shape, dtype=np.int32)
        self.lib.jxlo_palette(_ptr(idx, C.c_int32), idx.size, _ptr(pal, C.c_int32), num_colors,
                              pal.shape[1], nb_channels, bit_depth, _ptr(out, C.c_int32))
        return out

    def palette_delta_wp(self, index, palette, num_colors, num_deltas, nb_channels, bit_depth, wp_header):
        """wp_header: (p1c, p2c, p3ca, p3cb, p3cc, p3cd, p3ce, w0, w1, w2, w3)"""
        idx = np.ascontiguousarray(index, dtype=np.int32)
        pal = np.ascontiguousarray(palette, dtype=np.int32)
        hdr = np.ascontiguousarray(wp_header, dtype=np.uint32)
        h, w = idx.shape
        out = np.zeros((nb_channels, h, w), dtype=np.int32)
        self.lib.jxlo_palette_delta_wp(_ptr(idx, C.c_int32), w, h, _ptr(pal, C.c_int32), num_colors, num_deltas,
                                       pal.shape[1], nb_channels, bit_depth, _ptr(hdr, C.c_uint32), _ptr(out, C.c_int32))
        return out

    def unsqueeze_h(self, avg, res, out_w, simd_form=False):
        """simd_form: every step in the SIMD back-ends' wrapping i32 form instead of the scalar i64 definition"""
        avg = np.ascontiguousarray(avg, dtype=np.int32)
        res = np.ascontiguousarray(res, dtype=np.int32)
        h = avg.shape[0]
        out = np.zeros((h, out_w), dtype=np.int32)
        (self.lib.jxlo_unsqueeze_h_simd if simd_form else self.lib.jxlo_unsqueeze_h)(_ptr(avg, C.c_int32), avg.shape[1], _ptr(res, C.c_int32),
                                  max(res.shape[1], 1), out_w, h, _ptr(out, C.c_int32), out_w)
        return out

    SMOOTH_H, SMOOTH_V, SMOOTH_2D = 0, 1, 2

    def smooth_convolve(self, n25, two_d, cvt_rne=False):
        n = np.ascontiguousarray(n25, dtype=np.float32).reshape(25)
        out = np.zeros(4 if two_d else 2, dtype=np.int32)
        fn = self.lib.jxlo_smooth_convolve_2d if two_d else self.lib.jxlo_smooth_convolve_1d
        fn(_ptr(n, C.c_float), int(cvt_rne), _ptr(out, C.c_int32))
        return out

    def smooth_unsqueeze(self, kind, avg, out_w, out_h, x0=0, y0=0, cvt_rne=False, out=None):
        """smooth_{h,v,2d}_unsqueeze on the whole average channel `avg`; the out_w x out_h rectangle at (x0, y0)."""
        avg = np.ascontiguousarray(avg, dtype=np.int32)
        if out is None:
            out = np.zeros((out_h, out_w), dtype=np.int32)
        self.lib.jxlo_smooth_unsqueeze(kind, _ptr(avg, C.c_int32), avg.shape[1], avg.shape[1], avg.shape[0], x0, y0,
                                       _ptr(out, C.c_int32), out.shape[1], out_w, out_h, int(cvt_rne))
        return out

    def unsqueeze_v(self, avg, res, out_h, simd_form=False):
        avg = np.ascontiguousarray(avg, dtype=np.int32)
        res = np.ascontiguousarray(res, dtype=np.int32)
        w = avg.shape[1]
        out = np.zeros((out_h, w), dtype=np.int32)
        (self.lib.jxlo_unsqueeze_v_simd if simd_form else self.lib.jxlo_unsqueeze_v)(_ptr(avg, C.c_int32), w, _ptr(res, C.c_int32), w, w, out_h,
                                  _ptr(out, C.c_int32), w)
        return out
