// K2 Gaborish and K3a/b/c EPF0/1/2 as one kernel per stage ("unfused" path: parity
// reference on the device and the stage-level test hooks).  The production path is the
// LDS-tiled fused kernel in k_filters_fused.hip.
//
// Semantics: GaborishStage (jxl/src/render/stages/gaborish.rs:20-27, :83-85), Epf0/1/2Stage
// (jxl/src/render/stages/epf/epf0.rs:87-210, epf1.rs:84-146, epf2.rs:84-136), sad multiplier
// (epf/common.rs:31-41), MIN_SIGMA passthrough.  Edge rule: every tap outside the stage's
// w x h input is fetched at (mirror(x,w), mirror(y,h))
// (jxl/src/render/simple_pipeline/run_stage.rs:129-146, util/mirror.rs:8-19).
// Operation order == reference => bit-exact vs the oracle's FMA build.
#include "jxlh_internal.h"

namespace jxlh {
namespace {

struct Tap {
  const float* p;
  size_t stride;
  int w, h;
  __device__ __forceinline__ float operator()(int x, int y) const {
    return p[(size_t)mirror(y, h) * stride + mirror(x, w)];
  }
};

__global__ void k2_gaborish(const float* __restrict__ in, float* __restrict__ out, int w, int h, size_t stride,
                            float k0, float k1, float k2, int y0, int y1) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = y0 + blockIdx.y;
  if (x >= w || y >= y1) return;
  const Tap t{in, stride, w, h};
  const float p00 = t(x - 1, y - 1), p01 = t(x, y - 1), p02 = t(x + 1, y - 1);
  const float p10 = t(x - 1, y), p11 = t(x, y), p12 = t(x + 1, y);
  const float p20 = t(x - 1, y + 1), p21 = t(x, y + 1), p22 = t(x + 1, y + 1);
  float sum = p11 * k0;
  sum = __builtin_fmaf(k1, p01 + p10 + p21 + p12, sum);
  sum = __builtin_fmaf(k2, p00 + p02 + p20 + p22, sum);
  out[(size_t)y * stride + x] = sum;
}

#define AD(a, b) __builtin_fabsf((a) - (b))

__device__ __forceinline__ float sad_mul_at(int x, int y, float sm, float bsm) {
  const int xm = x & 7, ym = y & 7;
  return (xm == 0 || xm == 7 || ym == 0 || ym == 7) ? bsm : sm;
}

template <int STAGE>
__global__ void k3_epf(const EpfArgs a, int y0, int y1) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = y0 + blockIdx.y;
  if (x >= a.w || y >= y1) return;
  const size_t i = (size_t)y * a.stride + x;
  const float sigma = a.inv_sigma[(size_t)(y >> 3) * a.sigma_stride + (x >> 3)];
  if (sigma < kMinSigma) {
#pragma unroll
    for (int c = 0; c < 3; c++) a.out[c][i] = a.in[c][i];
    return;
  }
  const float inv_sigma = sigma * sad_mul_at(x, y, a.sm, a.bsm);
  if constexpr (STAGE == 0) {
    float sads[12];
#pragma unroll
    for (int k = 0; k < 12; k++) sads[k] = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const Tap t{a.in[c], a.stride, a.w, a.h};
      const float scale = a.scale[c];
#define P(cx, cy) t(x + (cx)-3, y + (cy)-3)
      const float p30 = P(3, 0), p21 = P(2, 1), p31 = P(3, 1), p41 = P(4, 1), p12 = P(1, 2), p22 = P(2, 2),
                  p32 = P(3, 2), p42 = P(4, 2), p52 = P(5, 2), p03 = P(0, 3), p13 = P(1, 3), p23 = P(2, 3),
                  p33 = P(3, 3), p43 = P(4, 3), p53 = P(5, 3), p63 = P(6, 3), p14 = P(1, 4), p24 = P(2, 4),
                  p34 = P(3, 4), p44 = P(4, 4), p54 = P(5, 4), p25 = P(2, 5), p35 = P(3, 5), p45 = P(4, 5),
                  p36 = P(3, 6);
#undef P
      const float d32_30 = AD(p32, p30), d32_21 = AD(p32, p21), d32_31 = AD(p32, p31), d32_41 = AD(p32, p41),
                  d32_12 = AD(p32, p12), d32_22 = AD(p32, p22), d32_42 = AD(p32, p42), d32_52 = AD(p32, p52),
                  d32_23 = AD(p32, p23), d32_34 = AD(p32, p34), d32_43 = AD(p32, p43), d32_33 = AD(p32, p33),
                  d23_21 = AD(p23, p21), d23_12 = AD(p23, p12), d23_22 = AD(p23, p22), d23_03 = AD(p23, p03),
                  d23_13 = AD(p23, p13), d23_33 = AD(p23, p33), d23_43 = AD(p23, p43), d23_14 = AD(p23, p14),
                  d23_24 = AD(p23, p24), d23_34 = AD(p23, p34), d23_25 = AD(p23, p25), d33_31 = AD(p33, p31),
                  d33_22 = AD(p33, p22), d33_42 = AD(p33, p42), d33_13 = AD(p33, p13), d33_43 = AD(p33, p43),
                  d33_53 = AD(p33, p53), d33_24 = AD(p33, p24), d33_34 = AD(p33, p34), d33_44 = AD(p33, p44),
                  d33_35 = AD(p33, p35), d43_41 = AD(p43, p41), d43_42 = AD(p43, p42), d43_52 = AD(p43, p52),
                  d43_53 = AD(p43, p53), d43_63 = AD(p43, p63), d43_34 = AD(p43, p34), d43_44 = AD(p43, p44),
                  d43_54 = AD(p43, p54), d43_45 = AD(p43, p45), d34_14 = AD(p34, p14), d34_24 = AD(p34, p24),
                  d34_44 = AD(p34, p44), d34_54 = AD(p34, p54), d34_25 = AD(p34, p25), d34_35 = AD(p34, p35),
                  d34_45 = AD(p34, p45), d34_36 = AD(p34, p36);
      sads[0] = __builtin_fmaf(scale, d32_30 + d23_21 + d33_31 + d43_41 + d32_34, sads[0]);
      sads[1] = __builtin_fmaf(scale, d32_21 + d23_12 + d33_22 + d32_43 + d23_34, sads[1]);
      sads[2] = __builtin_fmaf(scale, d32_31 + d23_22 + d32_33 + d43_42 + d33_34, sads[2]);
      sads[3] = __builtin_fmaf(scale, d32_41 + d32_23 + d33_42 + d43_52 + d43_34, sads[3]);
      sads[4] = __builtin_fmaf(scale, d32_12 + d23_03 + d33_13 + d23_43 + d34_14, sads[4]);
      sads[5] = __builtin_fmaf(scale, d32_22 + d23_13 + d23_33 + d33_43 + d34_24, sads[5]);
      sads[6] = __builtin_fmaf(scale, d32_42 + d23_33 + d33_43 + d43_53 + d34_44, sads[6]);
      sads[7] = __builtin_fmaf(scale, d32_52 + d23_43 + d33_53 + d43_63 + d34_54, sads[7]);
      sads[8] = __builtin_fmaf(scale, d32_23 + d23_14 + d33_24 + d43_34 + d34_25, sads[8]);
      sads[9] = __builtin_fmaf(scale, d32_33 + d23_24 + d33_34 + d43_44 + d34_35, sads[9]);
      sads[10] = __builtin_fmaf(scale, d32_43 + d23_34 + d33_44 + d43_54 + d34_45, sads[10]);
      sads[11] = __builtin_fmaf(scale, d32_34 + d23_25 + d33_35 + d43_45 + d34_36, sads[11]);
    }
    float wsum = 1.0f;
#pragma unroll
    for (int k = 0; k < 12; k++) {
      sads[k] = fmaxf(__builtin_fmaf(sads[k], inv_sigma, 1.0f), 0.0f);
      wsum += sads[k];
    }
    const float inv_w = 1.0f / wsum;
    constexpr int nbx[12] = {0, -1, 0, 1, -2, -1, 1, 2, -1, 0, 1, 0};
    constexpr int nby[12] = {-2, -1, -1, -1, 0, 0, 0, 0, 1, 1, 1, 2};
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const Tap t{a.in[c], a.stride, a.w, a.h};
      float acc = t(x, y);
#pragma unroll
      for (int k = 11; k >= 0; k--) acc = __builtin_fmaf(t(x + nbx[k], y + nby[k]), sads[k], acc);
      a.out[c][i] = acc * inv_w;
    }
  } else if constexpr (STAGE == 1) {
    float sads[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const Tap t{a.in[c], a.stride, a.w, a.h};
      const float scale = a.scale[c];
#define P(cx, cy) t(x + (cx)-2, y + (cy)-2)
      const float p20 = P(2, 0), p11 = P(1, 1), p21 = P(2, 1), p31 = P(3, 1), p02 = P(0, 2), p12 = P(1, 2),
                  p22 = P(2, 2), p32 = P(3, 2), p42 = P(4, 2), p13 = P(1, 3), p23 = P(2, 3), p33 = P(3, 3),
                  p24 = P(2, 4);
#undef P
      const float d20_21 = AD(p20, p21), d11_21 = AD(p11, p21), d22_21 = AD(p22, p21), d31_21 = AD(p31, p21),
                  d02_12 = AD(p02, p12), d11_12 = AD(p11, p12), d12_22 = AD(p22, p12), d31_32 = AD(p31, p32),
                  d22_32 = AD(p22, p32), d42_32 = AD(p42, p32), d13_12 = AD(p13, p12), d22_23 = AD(p22, p23),
                  d13_23 = AD(p13, p23), d33_23 = AD(p33, p23), d33_32 = AD(p33, p32), d24_23 = AD(p24, p23);
      sads[0] = __builtin_fmaf(d20_21 + d11_12 + d22_21 + d31_32 + d22_23, scale, sads[0]);
      sads[1] = __builtin_fmaf(d11_21 + d02_12 + d12_22 + d22_32 + d13_23, scale, sads[1]);
      sads[2] = __builtin_fmaf(d31_21 + d12_22 + d22_32 + d42_32 + d33_23, scale, sads[2]);
      sads[3] = __builtin_fmaf(d22_21 + d13_12 + d22_23 + d33_32 + d24_23, scale, sads[3]);
    }
    float wsum = 1.0f;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      sads[k] = fmaxf(__builtin_fmaf(sads[k], inv_sigma, 1.0f), 0.0f);
      wsum += sads[k];
    }
    const float inv_w = 1.0f / wsum;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const Tap t{a.in[c], a.stride, a.w, a.h};
      float acc = t(x, y);
      acc = __builtin_fmaf(t(x, y + 1), sads[3], acc);
      acc = __builtin_fmaf(t(x + 1, y), sads[2], acc);
      acc = __builtin_fmaf(t(x - 1, y), sads[1], acc);
      acc = __builtin_fmaf(t(x, y - 1), sads[0], acc);
      a.out[c][i] = acc * inv_w;
    }
  } else {
    const Tap tx{a.in[0], a.stride, a.w, a.h}, ty{a.in[1], a.stride, a.w, a.h}, tb{a.in[2], a.stride, a.w, a.h};
    const float xc = tx(x, y), yc = ty(x, y), bc = tb(x, y);
    float wacc = 1.0f, xa = xc, ya = yc, ba = bc;
    constexpr int nbx[4] = {0, -1, 1, 0};
    constexpr int nby[4] = {-1, 0, 0, 1};
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const float cx = tx(x + nbx[k], y + nby[k]);
      const float cy = ty(x + nbx[k], y + nby[k]);
      const float cb = tb(x + nbx[k], y + nby[k]);
      const float sad = __builtin_fmaf(
          AD(cx, xc), a.scale[0], __builtin_fmaf(AD(cy, yc), a.scale[1], AD(cb, bc) * a.scale[2]));
      const float wgt = fmaxf(__builtin_fmaf(sad, inv_sigma, 1.0f), 0.0f);
      wacc += wgt;
      xa = __builtin_fmaf(wgt, cx, xa);
      ya = __builtin_fmaf(wgt, cy, ya);
      ba = __builtin_fmaf(wgt, cb, ba);
    }
    const float inv_w = 1.0f / wacc;
    a.out[0][i] = xa * inv_w;
    a.out[1][i] = ya * inv_w;
    a.out[2][i] = ba * inv_w;
  }
}

}  // namespace

void launch_gaborish(hipStream_t s, const float* in, float* out, int w, int h, size_t stride, float k0, float k1,
                     float k2, int y0, int y1) {
  if (y1 <= y0 || w <= 0) return;
  hipLaunchKernelGGL(k2_gaborish, dim3((w + 255) / 256, y1 - y0), dim3(256), 0, s, in, out, w, h, stride, k0, k1, k2,
                     y0, y1);
}

void launch_epf(hipStream_t s, int stage, const EpfArgs& a, int y0, int y1) {
  if (y1 <= y0 || a.w <= 0) return;
  const dim3 grid((a.w + 255) / 256, y1 - y0), block(256);
  if (stage == 0) {
    hipLaunchKernelGGL(k3_epf<0>, grid, block, 0, s, a, y0, y1);
  } else if (stage == 1) {
    hipLaunchKernelGGL(k3_epf<1>, grid, block, 0, s, a, y0, y1);
  } else {
    hipLaunchKernelGGL(k3_epf<2>, grid, block, 0, s, a, y0, y1);
  }
}

}  // namespace jxlh
