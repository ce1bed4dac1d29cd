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

// Samples of one pixel -> display-referred R, G, B in [0, 1] nominal.
//   YCBCR = false: XybStage (xyb.rs:220-240) + the sRGB transfer function (frame/render.rs:757-762)
//   YCBCR = true : YcbcrToRgbStage on planes ordered Cb, Y, Cr (render/stages/ycbcr.rs:35-78); such frames
//                  are not XYB-encoded, so no transfer-function stage follows (frame/render.rs:755-763)
template <bool YCBCR>
__device__ __forceinline__ void to_display_rgb(const XybParamsDev& p, float c0, float c1, float c2, float& r, float& g,
                                               float& b) {
  if constexpr (YCBCR) {
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
    r = linear_to_srgb(__builtin_fmaf(p.mat[0], l, __builtin_fmaf(p.mat[1], m, p.mat[2] * s)));
    g = linear_to_srgb(__builtin_fmaf(p.mat[3], l, __builtin_fmaf(p.mat[4], m, p.mat[5] * s)));
    b = linear_to_srgb(__builtin_fmaf(p.mat[6], l, __builtin_fmaf(p.mat[7], m, p.mat[8] * s)));
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
template <int CH, bool YCBCR>
__global__ __launch_bounds__(kOutThreads) void k_xyb_to_rgb16(const float* __restrict__ px, const float* __restrict__ py,
                                                              const float* __restrict__ pb, uint32_t stride, int w,
                                                              int y0, int rows, const XybParamsDev p,
                                                              uint16_t* __restrict__ out, size_t out_stride_elems) {
  const int x4 = (blockIdx.x * kOutThreads + threadIdx.x) * 4;
  const int r = blockIdx.y;
  if (x4 >= w || r >= rows) return;
  const size_t in = (size_t)(y0 + r) * stride + x4;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    if (x4 + i >= w) break;
    float rr, gg, bb;
    to_display_rgb<YCBCR>(p, px[in + i], py[in + i], pb[in + i], rr, gg, bb);
    uint16_t* o = out + (size_t)r * out_stride_elems + (size_t)(x4 + i) * CH;
    o[0] = (uint16_t)to_u16(rr);
    o[1] = (uint16_t)to_u16(gg);
    o[2] = (uint16_t)to_u16(bb);
    if constexpr (CH == 4) o[3] = 65535;
  }
}

// one thread = 4 consecutive pixels of one row
template <int CH, bool YCBCR>
__global__ __launch_bounds__(kOutThreads) void k_xyb_to_rgb8(const float* __restrict__ px, const float* __restrict__ py,
                                                             const float* __restrict__ pb, uint32_t stride, int w,
                                                             int y0, int rows, const XybParamsDev p,
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
    to_display_rgb<YCBCR>(p, vx[i], vy[i], vb[i], rr, gg, bb);
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

}  // namespace

void launch_xyb_to_rgb8(hipStream_t s, const float* const planes[3], size_t stride, int w, int y0, int rows,
                        const XybParamsDev* p, int channels, uint8_t* out, size_t out_stride) {
  if (w <= 0 || rows <= 0) return;
  const dim3 grid((unsigned)(((w + 3) / 4 + kOutThreads - 1) / kOutThreads), (unsigned)rows);
  const int aligned = ((reinterpret_cast<uintptr_t>(out) | out_stride) & 3) == 0;
  const XybParamsDev q = p ? *p : XybParamsDev{};
#define JXLH_LAUNCH8(CH, Y)                                                                                      \
  hipLaunchKernelGGL((k_xyb_to_rgb8<CH, Y>), grid, dim3(kOutThreads), 0, s, planes[0], planes[1], planes[2],     \
                     (uint32_t)stride, w, y0, rows, q, out, out_stride, aligned)
  if (p) {
    if (channels == 3) JXLH_LAUNCH8(3, false); else JXLH_LAUNCH8(4, false);
  } else {
    if (channels == 3) JXLH_LAUNCH8(3, true); else JXLH_LAUNCH8(4, true);
  }
#undef JXLH_LAUNCH8
}

void launch_xyb_to_rgb16(hipStream_t s, const float* const planes[3], size_t stride, int w, int y0, int rows,
                         const XybParamsDev* p, int channels, uint16_t* out, size_t out_stride_elems) {
  if (w <= 0 || rows <= 0) return;
  const dim3 grid((unsigned)(((w + 3) / 4 + kOutThreads - 1) / kOutThreads), (unsigned)rows);
  const XybParamsDev q = p ? *p : XybParamsDev{};
#define JXLH_LAUNCH16(CH, Y)                                                                                     \
  hipLaunchKernelGGL((k_xyb_to_rgb16<CH, Y>), grid, dim3(kOutThreads), 0, s, planes[0], planes[1], planes[2],    \
                     (uint32_t)stride, w, y0, rows, q, out, out_stride_elems)
  if (p) {
    if (channels == 3) JXLH_LAUNCH16(3, false); else JXLH_LAUNCH16(4, false);
  } else {
    if (channels == 3) JXLH_LAUNCH16(3, true); else JXLH_LAUNCH16(4, true);
  }
#undef JXLH_LAUNCH16
}

}  // namespace jxlh
