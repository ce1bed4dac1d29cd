// Output stages for an XYB frame shown as 8-bit sRGB, fused into ONE elementwise pass over the
// filtered planes (12 B/px in, 3-4 B/px out instead of three f32 round trips and a 12 B/px D2H):
//   XybStage             jxl/src/render/stages/xyb.rs:208-240   (cube + opsin inverse matrix, FMAs)
//   FromLinearStage/sRGB jxl/src/color/tf.rs:13-44, util/rational_poly.rs:20-35 (sqrt, two Horner
//                        chains with FMAs, IEEE division)
//   ConvertF32ToU8Stage  jxl/src/render/stages/convert.rs:570-606 (x255, 32x32 dither table,
//                        clamp, round to nearest even like the AVX2 store) + interleaved save
// in the order frame/render.rs:757-762, :118 chains them.  Bit-exact vs the oracle's FMA build.
#include "jxlh_internal.h"

namespace jxlh {
namespace {

__constant__ float kDitherDev[32 * 32] = {
#include "dither_table.inc"
};

constexpr int kOutThreads = 256;

__device__ __forceinline__ float linear_to_srgb(float x) {
  constexpr float P0 = -5.135152395e-4f, P1 = 5.287254571e-3f, P2 = 3.903842876e-1f, P3 = 1.474205315f,
                  P4 = 7.352629620e-1f;
  constexpr float Q0 = 1.004519624e-2f, Q1 = 3.036675394e-1f, Q2 = 1.340816930f, Q3 = 9.258482155e-1f,
                  Q4 = 2.424867759e-2f;
  const float a = __builtin_fabsf(x);
  const float t = __builtin_sqrtf(a);
  float yp = __builtin_fmaf(P4, t, P3);
  yp = __builtin_fmaf(yp, t, P2);
  yp = __builtin_fmaf(yp, t, P1);
  yp = __builtin_fmaf(yp, t, P0);
  float yq = __builtin_fmaf(Q4, t, Q3);
  yq = __builtin_fmaf(yq, t, Q2);
  yq = __builtin_fmaf(yq, t, Q1);
  yq = __builtin_fmaf(yq, t, Q0);
  const float r = (0.0031308f > a) ? a * 12.92f : yp / yq;
  return __builtin_copysignf(r, x);
}

#include "tf_constants.inc"

// util/rational_poly.rs:20-35 (FMA Horner, the SIMD form) and :13-17 (plain mul + add, the scalar form)
template <int NP, int NQ>
__device__ __forceinline__ float ratpoly_fma(float x, const float (&p)[NP], const float (&q)[NQ]) {
  float yp = p[NP - 1], yq = q[NQ - 1];
#pragma unroll
  for (int i = NP - 2; i >= 0; i--) yp = __builtin_fmaf(yp, x, p[i]);
#pragma unroll
  for (int i = NQ - 2; i >= 0; i--) yq = __builtin_fmaf(yq, x, q[i]);
  return yp / yq;
}
template <int NP, int NQ>
__device__ __forceinline__ float ratpoly_plain(float x, const float (&p)[NP], const float (&q)[NQ]) {
  float yp = p[NP - 1], yq = q[NQ - 1];
#pragma unroll
  for (int i = NP - 2; i >= 0; i--) yp = yp * x + p[i];
#pragma unroll
  for (int i = NQ - 2; i >= 0; i--) yq = yq * x + q[i];
  return yp / yq;
}

template <bool SIMD>
__device__ __forceinline__ float fast_log2f_dev(float x) {  // util/fast_math.rs:127-149
  const int32_t x_bits = __float_as_int(x);
  const int32_t exp_bits = (int32_t)((uint32_t)x_bits - 0x3f2aaaabu);
  const int32_t exp_shifted = exp_bits >> 23;
  const float mantissa = __int_as_float((int32_t)((uint32_t)x_bits - ((uint32_t)exp_shifted << 23)));
  const float m1 = mantissa - 1.0f;
  const float poly = SIMD ? ratpoly_fma(m1, kTf_LOG2F_P, kTf_LOG2F_Q) : ratpoly_plain(m1, kTf_LOG2F_P, kTf_LOG2F_Q);
  return poly + (float)exp_shifted;
}
template <bool SIMD>
__device__ __forceinline__ float fast_pow2f_dev(float x) {  // util/fast_math.rs:79-114
  const float x_floor = __builtin_floorf(x);
  const float e = __int_as_float((int32_t)(((uint32_t)((int32_t)x_floor + 127)) << 23));
  const float frac = x - x_floor;
  float num = frac + kTf_POW2F_NUMER[0], den;
  if constexpr (SIMD) {
    num = __builtin_fmaf(num, frac, kTf_POW2F_NUMER[1]);
    num = __builtin_fmaf(num, frac, kTf_POW2F_NUMER[2]);
    num = num * e;
    den = __builtin_fmaf(kTf_POW2F_DENOM[0], frac, kTf_POW2F_DENOM[1]);
    den = __builtin_fmaf(den, frac, kTf_POW2F_DENOM[2]);
    den = __builtin_fmaf(den, frac, kTf_POW2F_DENOM[3]);
  } else {
    num = num * frac + kTf_POW2F_NUMER[1];
    num = num * frac + kTf_POW2F_NUMER[2];
    num = num * e;
    den = kTf_POW2F_DENOM[0] * frac + kTf_POW2F_DENOM[1];
    den = den * frac + kTf_POW2F_DENOM[2];
    den = den * frac + kTf_POW2F_DENOM[3];
  }
  return num / den;
}
template <bool SIMD>
__device__ __forceinline__ float fast_powf_dev(float base, float e) {
  return fast_pow2f_dev<SIMD>(fast_log2f_dev<SIMD>(base) * e);
}

__device__ __forceinline__ float linear_to_bt709(float x) {  // color/tf.rs:115-148
  const float a = __builtin_fabsf(x);
  const float r = (0.018f > a) ? a * 4.5f : ratpoly_fma(__builtin_sqrtf(a), kTf_BT709_P, kTf_BT709_Q);
  return __builtin_copysignf(r, x);
}
__device__ __forceinline__ float linear_to_pq(float y_mult, float x) {  // color/tf.rs:288-314
  const float a = __builtin_fabsf(x);
  const float a_1_4 = __builtin_sqrtf(__builtin_sqrtf(a * y_mult));
  const float y_small = ratpoly_fma(a_1_4, kTf_PQ_INV_EOTF_P_SMALL, kTf_PQ_INV_EOTF_Q_SMALL);
  const float y_large = ratpoly_fma(a_1_4, kTf_PQ_INV_EOTF_P, kTf_PQ_INV_EOTF_Q);
  return __builtin_copysignf((1e-4f > a) ? y_small : y_large, x);
}
__device__ __forceinline__ float scene_to_hlg(float x) {  // color/tf.rs:482-497
  constexpr double kA = 0.17883277, kB = 1.0 - 4.0 * kA, kC = 0.5599107295;
  constexpr float k = (float)(kA * 0.693147180559945309417232121458176568), hb = (float)kB, hc = (float)kC;
  const float a = __builtin_fabsf(x);
  const float y = (a <= 1.0f / 12.0f) ? __builtin_sqrtf(3.0f * a) : k * fast_log2f_dev<false>(12.0f * a - hb) + hc;
  return __builtin_copysignf(y, x);
}

// FromLinearStage (render/stages/from_linear.rs:57-112) on one pixel
template <int TF>
__device__ __forceinline__ void from_linear(const TfParamsDev& t, float& r, float& g, float& b) {
  if constexpr (TF == kTfSrgb) {
    r = linear_to_srgb(r);
    g = linear_to_srgb(g);
    b = linear_to_srgb(b);
  } else if constexpr (TF == kTfBt709) {
    r = linear_to_bt709(r);
    g = linear_to_bt709(g);
    b = linear_to_bt709(b);
  } else if constexpr (TF == kTfPq) {
    const float y_mult = t.param * (1.0f / 10000.0f);
    r = linear_to_pq(y_mult, r);
    g = linear_to_pq(y_mult, g);
    b = linear_to_pq(y_mult, b);
  } else if constexpr (TF == kTfHlg) {
    if (!(__builtin_fabsf(t.param) < 0.1f)) {  // hlg_ootf_inner (color/tf.rs:379-393), exponent from the host
      const float mixed = __builtin_fmaf(r, t.lum[0], __builtin_fmaf(g, t.lum[1], b * t.lum[2]));
      const float mult = fast_powf_dev<false>(mixed, t.param);
      r *= mult;
      g *= mult;
      b *= mult;
    }
    r = scene_to_hlg(r);
    g = scene_to_hlg(g);
    b = scene_to_hlg(b);
  } else if constexpr (TF == kTfGamma) {
    r = __builtin_copysignf(fast_powf_dev<true>(__builtin_fabsf(r), t.param), r);
    g = __builtin_copysignf(fast_powf_dev<true>(__builtin_fabsf(g), t.param), g);
    b = __builtin_copysignf(fast_powf_dev<true>(__builtin_fabsf(b), t.param), b);
  }
}

// Samples of one pixel -> display-referred R, G, B in [0, 1] nominal.
//   YCBCR = false: XybStage (xyb.rs:220-240) + the sRGB transfer function (frame/render.rs:757-762)
//   YCBCR = true : YcbcrToRgbStage on planes ordered Cb, Y, Cr (render/stages/ycbcr.rs:35-78); such frames
//                  are not XYB-encoded, so no transfer-function stage follows (frame/render.rs:755-763)
//   MODE = kTfLinear .. kTfGamma: XybStage, then that transfer function;  kModeYcbcr;  kModeNone: planes are RGB already
template <int MODE>
__device__ __forceinline__ void to_display_rgb(const XybParamsDev& p, const TfParamsDev& t, float c0, float c1, float c2,
                                               float& r, float& g, float& b) {
  if constexpr (MODE == kModeNone) {
    r = c0;
    g = c1;
    b = c2;
  } else if constexpr (MODE == kModeYcbcr) {
    constexpr float k128 = 128.0f / 255.0f, kCrToR = 1.402f, kCrToG = -0.299f * 1.402f / 0.587f,
                    kCbToG = -0.114f * 1.772f / 0.587f, kCbToB = 1.772f;
    const float y = c1 + k128;
    r = __builtin_fmaf(c2, kCrToR, y);
    g = __builtin_fmaf(c2, kCrToG, __builtin_fmaf(c0, kCbToG, y));
    b = __builtin_fmaf(c0, kCbToB, y);
  } else {
    float l = c1 + c0 - p.bias_cbrt[0];
    float m = c1 - c0 - p.bias_cbrt[1];
    float s = c2 - p.bias_cbrt[2];
    const float l2 = l * l, m2 = m * m, s2 = s * s;
    const float sl = l * p.intensity_scale, sm = m * p.intensity_scale, ss = s * p.intensity_scale;
    l = __builtin_fmaf(l2, sl, p.scaled_bias[0]);
    m = __builtin_fmaf(m2, sm, p.scaled_bias[1]);
    s = __builtin_fmaf(s2, ss, p.scaled_bias[2]);
    r = __builtin_fmaf(p.mat[0], l, __builtin_fmaf(p.mat[1], m, p.mat[2] * s));
    g = __builtin_fmaf(p.mat[3], l, __builtin_fmaf(p.mat[4], m, p.mat[5] * s));
    b = __builtin_fmaf(p.mat[6], l, __builtin_fmaf(p.mat[7], m, p.mat[8] * s));
    from_linear<MODE>(t, r, g, b);
  }
}

__device__ __forceinline__ uint32_t to_u8(float v, const float* __restrict__ dither, int x, int y, int c) {
  const float d = dither[((y + c * 13) & 31) * 32 + ((x + c * 23) & 31)];
  const float dithered = v * 255.0f + d;
  float clamped = dithered > 0.0f ? dithered : 0.0f;
  clamped = clamped < 255.0f ? clamped : 255.0f;
  return (uint32_t)__builtin_rintf(clamped);
}

__device__ __forceinline__ uint32_t to_u16(float v) {  // f32_to_u16_simd (convert.rs:743-761), 16-bit
  float clamped = v > 0.0f ? v : 0.0f;
  clamped = clamped < 1.0f ? clamped : 1.0f;
  return (uint32_t)__builtin_rintf(clamped * 65535.0f);
}

// one thread = 4 consecutive pixels of one row; 16-bit samples (no dither), little endian
template <int CH, int MODE>
__global__ __launch_bounds__(kOutThreads) void k_xyb_to_rgb16(const float* __restrict__ px, const float* __restrict__ py,
                                                              const float* __restrict__ pb, uint32_t stride, int w,
                                                              int y0, int rows, const XybParamsDev p, const TfParamsDev t,
                                                              uint16_t* __restrict__ out, size_t out_stride_elems) {
  const int x4 = (blockIdx.x * kOutThreads + threadIdx.x) * 4;
  const int r = blockIdx.y;
  if (x4 >= w || r >= rows) return;
  const size_t in = (size_t)(y0 + r) * stride + x4;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    if (x4 + i >= w) break;
    float rr, gg, bb;
    to_display_rgb<MODE>(p, t, px[in + i], py[in + i], pb[in + i], rr, gg, bb);
    uint16_t* o = out + (size_t)r * out_stride_elems + (size_t)(x4 + i) * CH;
    o[0] = (uint16_t)to_u16(rr);
    o[1] = (uint16_t)to_u16(gg);
    o[2] = (uint16_t)to_u16(bb);
    if constexpr (CH == 4) o[3] = 65535;
  }
}

// one thread = 4 consecutive pixels of one row
template <int CH, int MODE>
__global__ __launch_bounds__(kOutThreads) void k_xyb_to_rgb8(const float* __restrict__ px, const float* __restrict__ py,
                                                             const float* __restrict__ pb, uint32_t stride, int w,
                                                             int y0, int rows, const XybParamsDev p, const TfParamsDev t,
                                                             uint8_t* __restrict__ out, size_t out_stride, int aligned) {
  __shared__ float s_dither[32 * 32];
  for (int i = threadIdx.x; i < 32 * 32; i += kOutThreads) s_dither[i] = kDitherDev[i];
  __syncthreads();
  const int x4 = (blockIdx.x * kOutThreads + threadIdx.x) * 4;
  const int r = blockIdx.y;
  if (x4 >= w || r >= rows) return;
  const int y = y0 + r;
  const size_t in = (size_t)y * stride + x4;
  float vx[4], vy[4], vb[4];
  if (x4 + 4 <= w) {
    const float4 a = *reinterpret_cast<const float4*>(px + in), b = *reinterpret_cast<const float4*>(py + in),
                 c = *reinterpret_cast<const float4*>(pb + in);
    vx[0] = a.x; vx[1] = a.y; vx[2] = a.z; vx[3] = a.w;
    vy[0] = b.x; vy[1] = b.y; vy[2] = b.z; vy[3] = b.w;
    vb[0] = c.x; vb[1] = c.y; vb[2] = c.z; vb[3] = c.w;
  } else {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const bool ok = x4 + i < w;
      vx[i] = ok ? px[in + i] : 0.0f;
      vy[i] = ok ? py[in + i] : 0.0f;
      vb[i] = ok ? pb[in + i] : 0.0f;
    }
  }
  uint32_t q[4][3];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    float rr, gg, bb;
    to_display_rgb<MODE>(p, t, vx[i], vy[i], vb[i], rr, gg, bb);
    q[i][0] = to_u8(rr, s_dither, x4 + i, y, 0);
    q[i][1] = to_u8(gg, s_dither, x4 + i, y, 1);
    q[i][2] = to_u8(bb, s_dither, x4 + i, y, 2);
  }
  uint8_t* o = out + (size_t)r * out_stride + (size_t)x4 * CH;
  if (aligned && x4 + 4 <= w) {
    uint32_t* o32 = reinterpret_cast<uint32_t*>(o);
    if constexpr (CH == 3) {
      o32[0] = q[0][0] | (q[0][1] << 8) | (q[0][2] << 16) | (q[1][0] << 24);
      o32[1] = q[1][1] | (q[1][2] << 8) | (q[2][0] << 16) | (q[2][1] << 24);
      o32[2] = q[2][2] | (q[3][0] << 8) | (q[3][1] << 16) | (q[3][2] << 24);
    } else {
#pragma unroll
      for (int i = 0; i < 4; i++) o32[i] = q[i][0] | (q[i][1] << 8) | (q[i][2] << 16) | 0xff000000u;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if (x4 + i < w) {
        o[i * CH] = (uint8_t)q[i][0];
        o[i * CH + 1] = (uint8_t)q[i][1];
        o[i * CH + 2] = (uint8_t)q[i][2];
        if constexpr (CH == 4) o[i * CH + 3] = 255;
      }
    }
  }
}

// YCbCr frame with sub-sampled chroma, straight from K1's output: HorizontalChromaUpsample / VerticalChromaUpsample
// (the arithmetic of k_chroma.hip: blend = fma(neighbour, 0.25, centre * 0.75), horizontal stage first, mirrored at the
// edges of the sub-sampled channel) evaluated per output pixel, then YcbcrToRgbStage and the integer conversion.
// Used when nothing sits between the transforms and the output (no filters, upsampling or noise): the full-resolution
// chroma planes are then never written or read back.
__device__ __forceinline__ int mirror_idx(int v, int s) {
  while (v < 0 || v >= s) v = v < 0 ? -v - 1 : 2 * s - v - 1;
  return v;
}
__device__ __forceinline__ float chroma_blend(float neighbour, float centre) {
  return __builtin_fmaf(neighbour, 0.25f, centre * 0.75f);
}

template <int CH, bool U16>
__global__ __launch_bounds__(kOutThreads) void k_ycbcr_sub_to_rgb(const SubPlanesDev sp, uint32_t stride, int w, int y0,
                                                                  int rows, void* __restrict__ out, size_t out_stride,
                                                                  int aligned) {
  __shared__ float s_dither[32 * 32];
  if constexpr (!U16) {
    for (int i = threadIdx.x; i < 32 * 32; i += kOutThreads) s_dither[i] = kDitherDev[i];
    __syncthreads();
  }
  const int x4 = (blockIdx.x * kOutThreads + threadIdx.x) * 4;
  const int r = blockIdx.y;
  if (x4 >= w || r >= rows) return;
  const int y = y0 + r;
  float v[3][4];
  const bool whole = x4 + 4 <= w;
  auto load4 = [&](const float* __restrict__ row, int x0, int n, float (&o)[4]) {  // row[x0 .. x0+3], zero past n
    if (x0 + 4 <= n && ((reinterpret_cast<uintptr_t>(row + x0) & 15) == 0)) {
      const float4 t = *reinterpret_cast<const float4*>(row + x0);
      o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
    } else {
#pragma unroll
      for (int i = 0; i < 4; i++) o[i] = x0 + i < n ? row[x0 + i] : 0.0f;
    }
  };
  auto mirror1 = [](int v, int n) {  // one reflection, then clamped: exact for the +-1 / +-2 excursions used here
    v = v < 0 ? -v - 1 : (v >= n ? 2 * n - v - 1 : v);
    return v < 0 ? 0 : (v >= n ? n - 1 : v);
  };
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const float* __restrict__ p = sp.p[c];
    const int hs = sp.hs[c], vs = sp.vs[c];
    if (!(hs | vs)) {
      load4(p + (size_t)y * stride, x4, w, v[c]);
      continue;
    }
    const int cw = sp.cw[c], ch = sp.ch[c];
    const int sy = y >> vs;
    const float* __restrict__ row_c = p + (size_t)sy * stride;
    const float* __restrict__ row_n = p + (size_t)(vs ? mirror1((y & 1) ? sy + 1 : sy - 1, ch) : sy) * stride;
    float hc[4], hn[4];  // the horizontal stage's output at the four pixels, centre row and neighbour row
    if (hs) {
      // pixels x4 .. x4+3 sit on sub-samples s0, s0, s0+1, s0+1 (x4 is a multiple of 4) with horizontal neighbours
      // s0-1, s0+1, s0, s0+2
      const int s0 = x4 >> 1;
      const int ia = mirror1(s0 - 1, cw), ib = min(s0, cw - 1), ic = mirror1(s0 + 1, cw), id = mirror1(s0 + 2, cw);
      const float ca = row_c[ia], cb = row_c[ib], cc = row_c[ic], cd = row_c[id];
      hc[0] = chroma_blend(ca, cb);
      hc[1] = chroma_blend(cc, cb);
      hc[2] = chroma_blend(cb, cc);
      hc[3] = chroma_blend(cd, cc);
      if (vs) {
        const float na = row_n[ia], nb = row_n[ib], nc = row_n[ic], nd = row_n[id];
        hn[0] = chroma_blend(na, nb);
        hn[1] = chroma_blend(nc, nb);
        hn[2] = chroma_blend(nb, nc);
        hn[3] = chroma_blend(nd, nc);
      }
    } else {
      load4(row_c, x4, w, hc);
      load4(row_n, x4, w, hn);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) v[c][i] = vs ? chroma_blend(hn[i], hc[i]) : hc[i];
  }
  (void)whole;
  uint32_t q[4][3];
  const XybParamsDev xp = {};
  const TfParamsDev tp = {};
#pragma unroll
  for (int i = 0; i < 4; i++) {
    float rr, gg, bb;
    to_display_rgb<kModeYcbcr>(xp, tp, v[0][i], v[1][i], v[2][i], rr, gg, bb);
    if constexpr (U16) {
      q[i][0] = to_u16(rr);
      q[i][1] = to_u16(gg);
      q[i][2] = to_u16(bb);
    } else {
      q[i][0] = to_u8(rr, s_dither, x4 + i, y, 0);
      q[i][1] = to_u8(gg, s_dither, x4 + i, y, 1);
      q[i][2] = to_u8(bb, s_dither, x4 + i, y, 2);
    }
  }
  if constexpr (U16) {
    uint16_t* o = static_cast<uint16_t*>(out) + (size_t)r * out_stride + (size_t)x4 * CH;
#pragma unroll
    for (int i = 0; i < 4; i++)
      if (x4 + i < w) {
        o[i * CH] = (uint16_t)q[i][0];
        o[i * CH + 1] = (uint16_t)q[i][1];
        o[i * CH + 2] = (uint16_t)q[i][2];
        if constexpr (CH == 4) o[i * CH + 3] = 65535;
      }
  } else {
    uint8_t* o = static_cast<uint8_t*>(out) + (size_t)r * out_stride + (size_t)x4 * CH;
    if (aligned && x4 + 4 <= w) {
      uint32_t* o32 = reinterpret_cast<uint32_t*>(o);
      if constexpr (CH == 3) {
        o32[0] = q[0][0] | (q[0][1] << 8) | (q[0][2] << 16) | (q[1][0] << 24);
        o32[1] = q[1][1] | (q[1][2] << 8) | (q[2][0] << 16) | (q[2][1] << 24);
        o32[2] = q[2][2] | (q[3][0] << 8) | (q[3][1] << 16) | (q[3][2] << 24);
      } else {
#pragma unroll
        for (int i = 0; i < 4; i++) o32[i] = q[i][0] | (q[i][1] << 8) | (q[i][2] << 16) | 0xff000000u;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; i++)
        if (x4 + i < w) {
          o[i * CH] = (uint8_t)q[i][0];
          o[i * CH + 1] = (uint8_t)q[i][1];
          o[i * CH + 2] = (uint8_t)q[i][2];
          if constexpr (CH == 4) o[i * CH + 3] = 255;
        }
    }
  }
}

}  // namespace

// out_stride: bytes (8-bit) or elements (16-bit)
void launch_ycbcr_sub_to_rgb(hipStream_t s, const SubPlanesDev& sp, size_t stride, int w, int y0, int rows, int channels,
                             int bits, void* out, size_t out_stride) {
  if (w <= 0 || rows <= 0) return;
  const dim3 grid((unsigned)(((w + 3) / 4 + kOutThreads - 1) / kOutThreads), (unsigned)rows);
  const int aligned = ((reinterpret_cast<uintptr_t>(out) | out_stride) & 3) == 0;
  if (bits == 8) {
    if (channels == 3)
      hipLaunchKernelGGL((k_ycbcr_sub_to_rgb<3, false>), grid, dim3(kOutThreads), 0, s, sp, (uint32_t)stride, w, y0, rows,
                         out, out_stride, aligned);
    else
      hipLaunchKernelGGL((k_ycbcr_sub_to_rgb<4, false>), grid, dim3(kOutThreads), 0, s, sp, (uint32_t)stride, w, y0, rows,
                         out, out_stride, aligned);
  } else {
    if (channels == 3)
      hipLaunchKernelGGL((k_ycbcr_sub_to_rgb<3, true>), grid, dim3(kOutThreads), 0, s, sp, (uint32_t)stride, w, y0, rows,
                         out, out_stride, aligned);
    else
      hipLaunchKernelGGL((k_ycbcr_sub_to_rgb<4, true>), grid, dim3(kOutThreads), 0, s, sp, (uint32_t)stride, w, y0, rows,
                         out, out_stride, aligned);
  }
}

template <int MODE>
void launch8_mode(hipStream_t s, const float* const planes[3], size_t stride, int w, int y0, int rows, const XybParamsDev& q,
                  const TfParamsDev& t, int channels, uint8_t* out, size_t out_stride) {
  const dim3 grid((unsigned)(((w + 3) / 4 + kOutThreads - 1) / kOutThreads), (unsigned)rows);
  const int aligned = ((reinterpret_cast<uintptr_t>(out) | out_stride) & 3) == 0;
  if (channels == 3)
    hipLaunchKernelGGL((k_xyb_to_rgb8<3, MODE>), grid, dim3(kOutThreads), 0, s, planes[0], planes[1], planes[2],
                       (uint32_t)stride, w, y0, rows, q, t, out, out_stride, aligned);
  else
    hipLaunchKernelGGL((k_xyb_to_rgb8<4, MODE>), grid, dim3(kOutThreads), 0, s, planes[0], planes[1], planes[2],
                       (uint32_t)stride, w, y0, rows, q, t, out, out_stride, aligned);
}
template <int MODE>
void launch16_mode(hipStream_t s, const float* const planes[3], size_t stride, int w, int y0, int rows,
                   const XybParamsDev& q, const TfParamsDev& t, int channels, uint16_t* out, size_t out_stride_elems) {
  const dim3 grid((unsigned)(((w + 3) / 4 + kOutThreads - 1) / kOutThreads), (unsigned)rows);
  if (channels == 3)
    hipLaunchKernelGGL((k_xyb_to_rgb16<3, MODE>), grid, dim3(kOutThreads), 0, s, planes[0], planes[1], planes[2],
                       (uint32_t)stride, w, y0, rows, q, t, out, out_stride_elems);
  else
    hipLaunchKernelGGL((k_xyb_to_rgb16<4, MODE>), grid, dim3(kOutThreads), 0, s, planes[0], planes[1], planes[2],
                       (uint32_t)stride, w, y0, rows, q, t, out, out_stride_elems);
}

// mode: kTfLinear..kTfGamma (XYB frame + that transfer function), kModeYcbcr, kModeNone
void launch_xyb_to_rgb8(hipStream_t s, const float* const planes[3], size_t stride, int w, int y0, int rows, int mode,
                        const XybParamsDev& q, const TfParamsDev& t, int channels, uint8_t* out, size_t out_stride) {
  if (w <= 0 || rows <= 0) return;
  switch (mode) {
    case kTfLinear: launch8_mode<kTfLinear>(s, planes, stride, w, y0, rows, q, t, channels, out, out_stride); break;
    case kTfSrgb: launch8_mode<kTfSrgb>(s, planes, stride, w, y0, rows, q, t, channels, out, out_stride); break;
    case kTfBt709: launch8_mode<kTfBt709>(s, planes, stride, w, y0, rows, q, t, channels, out, out_stride); break;
    case kTfPq: launch8_mode<kTfPq>(s, planes, stride, w, y0, rows, q, t, channels, out, out_stride); break;
    case kTfHlg: launch8_mode<kTfHlg>(s, planes, stride, w, y0, rows, q, t, channels, out, out_stride); break;
    case kTfGamma: launch8_mode<kTfGamma>(s, planes, stride, w, y0, rows, q, t, channels, out, out_stride); break;
    case kModeYcbcr: launch8_mode<kModeYcbcr>(s, planes, stride, w, y0, rows, q, t, channels, out, out_stride); break;
    default: launch8_mode<kModeNone>(s, planes, stride, w, y0, rows, q, t, channels, out, out_stride); break;
  }
}

void launch_xyb_to_rgb16(hipStream_t s, const float* const planes[3], size_t stride, int w, int y0, int rows, int mode,
                         const XybParamsDev& q, const TfParamsDev& t, int channels, uint16_t* out, size_t out_stride_elems) {
  if (w <= 0 || rows <= 0) return;
  switch (mode) {
    case kTfLinear: launch16_mode<kTfLinear>(s, planes, stride, w, y0, rows, q, t, channels, out, out_stride_elems); break;
    case kTfSrgb: launch16_mode<kTfSrgb>(s, planes, stride, w, y0, rows, q, t, channels, out, out_stride_elems); break;
    case kTfBt709: launch16_mode<kTfBt709>(s, planes, stride, w, y0, rows, q, t, channels, out, out_stride_elems); break;
    case kTfPq: launch16_mode<kTfPq>(s, planes, stride, w, y0, rows, q, t, channels, out, out_stride_elems); break;
    case kTfHlg: launch16_mode<kTfHlg>(s, planes, stride, w, y0, rows, q, t, channels, out, out_stride_elems); break;
    case kTfGamma: launch16_mode<kTfGamma>(s, planes, stride, w, y0, rows, q, t, channels, out, out_stride_elems); break;
    case kModeYcbcr: launch16_mode<kModeYcbcr>(s, planes, stride, w, y0, rows, q, t, channels, out, out_stride_elems); break;
    default: launch16_mode<kModeNone>(s, planes, stride, w, y0, rows, q, t, channels, out, out_stride_elems); break;
  }
}

}  // namespace jxlh
