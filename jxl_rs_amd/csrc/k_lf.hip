// K0a LF dequantisation, K0b adaptive LF smoothing, K3sigma EPF sigma map.
// All three work on the 1/8-resolution block grid (1/64 of the pixels): plain
// coalesced elementwise / 3x3 stencil kernels, HBM-bound and tiny.
//
// Reference: dequant_lf (jxl/src/frame/modular/mod.rs:837-929), adaptive_lf_smoothing
// (jxl/src/frame/adaptive_lf_smoothing.rs:15-125), SigmaSource::new
// (jxl/src/features/epf.rs:35-87).  These are plain scalar Rust in the reference
// (no mul_add), so every a*b+c below is deliberately unfused (-ffp-contract=off).
#include "jxlh_internal.h"

namespace jxlh {
namespace {

__global__ void k0a_dequant_lf(const int32_t* __restrict__ qy, const int32_t* __restrict__ qx,
                               const int32_t* __restrict__ qb, size_t qstride, float* __restrict__ ox,
                               float* __restrict__ oy, float* __restrict__ ob, size_t ostride, int w, int h,
                               float fac_x, float fac_y, float fac_b, float cfl_x, float cfl_b) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= w || y >= h) return;
  const size_t qi = (size_t)y * qstride + x, oi = (size_t)y * ostride + x;
  const float in_x = (float)qx[qi] * fac_x;
  const float in_y = (float)qy[qi] * fac_y;
  const float in_b = (float)qb[qi] * fac_b;
  oy[oi] = in_y;
  ox[oi] = in_y * cfl_x + in_x;
  ob[oi] = in_y * cfl_b + in_b;
}

__global__ void k0a_dequant_lf_plain(const int32_t* __restrict__ qy, const int32_t* __restrict__ qx,
                                     const int32_t* __restrict__ qb, size_t qstride, float* __restrict__ ox,
                                     float* __restrict__ oy, float* __restrict__ ob, size_t ostride, int w, int h,
                                     float fac_x, float fac_y, float fac_b) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= w || y >= h) return;
  const size_t qi = (size_t)y * qstride + x, oi = (size_t)y * ostride + x;
  ox[oi] = (float)qx[qi] * fac_x;
  oy[oi] = (float)qy[qi] * fac_y;
  ob[oi] = (float)qb[qi] * fac_b;
}

constexpr float kWSide = 0.20345139757231578f;
constexpr float kWCorner = 0.0334829185968739f;

struct Planes3 {
  const float* in[3];
  float* out[3];
  float lf_factors[3];
};

__global__ void k0b_lf_smooth(const Planes3 p, int w, int h) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= w || y >= h) return;
  const size_t i = (size_t)y * w + x;
  if (y == 0 || y == h - 1 || x == 0 || x == w - 1) {
#pragma unroll
    for (int c = 0; c < 3; c++) p.out[c][i] = p.in[c][i];
    return;
  }
  const float w_center = 1.0f - 4.0f * (kWSide + kWCorner);
  float gap = 0.5f, mc[3], sm[3];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const float* t = p.in[c] + i - w;
    const float* m = p.in[c] + i;
    const float* b = p.in[c] + i + w;
    const float corner = t[-1] + t[1] + b[-1] + b[1];
    const float side = m[-1] + m[1] + t[0] + b[0];
    mc[c] = m[0];
    sm[c] = corner * kWCorner + side * kWSide + mc[c] * w_center;
    gap = fmaxf(gap, fabsf((mc[c] - sm[c]) / p.lf_factors[c]));
  }
  const float factor = fmaxf(3.0f - 4.0f * gap, 0.0f);
#pragma unroll
  for (int c = 0; c < 3; c++) p.out[c][i] = (sm[c] - mc[c]) * factor + mc[c];
}

struct SharpLut {
  float v[8];
};

__global__ void k3_sigma_map(const int32_t* __restrict__ raw_quant, const uint8_t* __restrict__ epf_map,
                             float* __restrict__ inv_sigma, size_t n, float quant_scale, float epf_quant_mul,
                             const SharpLut lut) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float sigma_quant = epf_quant_mul / (quant_scale * (float)raw_quant[i] * kInvSigmaNum);
  const float sigma = fminf(sigma_quant * lut.v[epf_map[i] & 7], -1e-4f);
  inv_sigma[i] = 1.0f / sigma;
}

}  // namespace

void launch_dequant_lf(hipStream_t s, const int32_t* qy, const int32_t* qx, const int32_t* qb, size_t qstride,
                       float* ox, float* oy, float* ob, size_t ostride, int w, int h, float fac_x, float fac_y,
                       float fac_b, float cfl_x, float cfl_b) {
  if (w <= 0 || h <= 0) return;
  hipLaunchKernelGGL(k0a_dequant_lf, dim3((w + 255) / 256, h), dim3(256), 0, s, qy, qx, qb, qstride, ox, oy, ob,
                     ostride, w, h, fac_x, fac_y, fac_b, cfl_x, cfl_b);
}

void launch_dequant_lf_plain(hipStream_t s, const int32_t* qy, const int32_t* qx, const int32_t* qb, size_t qstride,
                             float* ox, float* oy, float* ob, size_t ostride, int w, int h, float fac_x, float fac_y,
                             float fac_b) {
  if (w <= 0 || h <= 0) return;
  hipLaunchKernelGGL(k0a_dequant_lf_plain, dim3((w + 255) / 256, h), dim3(256), 0, s, qy, qx, qb, qstride, ox, oy, ob,
                     ostride, w, h, fac_x, fac_y, fac_b);
}

void launch_lf_smooth(hipStream_t s, const float* const in[3], float* const out[3], int w, int h,
                      const float lf_factors[3]) {
  if (w <= 0 || h <= 0) return;
  Planes3 p;
  for (int c = 0; c < 3; c++) {
    p.in[c] = in[c];
    p.out[c] = out[c];
    p.lf_factors[c] = lf_factors[c];
  }
  hipLaunchKernelGGL(k0b_lf_smooth, dim3((w + 255) / 256, h), dim3(256), 0, s, p, w, h);
}

namespace {
__global__ __launch_bounds__(256) void k_check_tables(const float* __restrict__ t, size_t n, int* __restrict__ ok) {
  bool bad = false;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float v = t[i];
    bad |= !(v >= 1e-20f) || !(v <= 1e20f);  // (NaN fails the first test.)  Inside this range a non-zero coefficient
                                             // never dequantises to a zero: see run_dct_class, direct path
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicAnd(ok, 0);
}
}  // namespace
void launch_check_tables(hipStream_t s, const float* tables, size_t n, int* ok) {
  (void)hipMemsetAsync(ok, 0, sizeof(int), s);
  if (n == 0) return;
  const int one = 1;
  (void)hipMemcpyAsync(ok, &one, sizeof(int), hipMemcpyHostToDevice, s);  // pageable: staged before the call returns
  hipLaunchKernelGGL(k_check_tables, dim3(64), dim3(256), 0, s, tables, n, ok);
}

void launch_sigma_map(hipStream_t s, const FrameDev& f, float epf_quant_mul, const float* sharp_lut) {
  const size_t n = (size_t)f.xblocks * f.yblocks;
  SharpLut lut;
  for (int i = 0; i < 8; i++) lut.v[i] = sharp_lut[i];
  const float quant_scale = 1.0f / f.inv_global_scale;
  hipLaunchKernelGGL(k3_sigma_map, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, f.raw_quant, f.epf_map,
                     f.inv_sigma, n, quant_scale, epf_quant_mul, lut);
}

}  // namespace jxlh
