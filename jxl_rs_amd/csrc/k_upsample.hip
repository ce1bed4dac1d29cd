// 2x / 4x / 8x upsampling: Upsample<N> of the reference (jxl/src/render/stages/upsample.rs).
// Every input pixel expands into an N x N patch; output (oy, ox) of the patch is a 5x5 convolution of the
// input window around the pixel with kernel[oy][ox] (25 taps, :175-215: three accumulators, tap t feeds
// accumulator t % 3, the first three taps are plain products, the rest FMAs; result (acc0 + acc1) + acc2),
// clamped to the minimum / maximum of the window (compute_minmax, :115-172).  BORDER (2, 2): the window is
// mirrored at the edges of the input image (util/mirror.rs:8-19).
//
// One thread = a short column of input pixels (the window slides down): 25 window values in registers, N*N*25 FMAs against weights whose index
// is uniform across the wavefront (scalar loads of the kernel table), N rows of N contiguous outputs
// (16-byte stores for N >= 4).  Bound: for N = 2 the 4 B/px read + 16 B/px write; for N = 8 the 1600 FMAs
// per input pixel (25 per output pixel) put it near the f32 vector roof instead -- still no MFMA shape:
// the 25-tap kernels differ per output phase and the data is a sliding window.
#include "jxlh_internal.h"

namespace jxlh {
namespace {

__device__ __forceinline__ int mirror_idx(int v, int s) {
  while (v < 0 || v >= s) v = v < 0 ? -v - 1 : 2 * s - v - 1;
  return v;
}

// R consecutive input rows per thread: the 5x5 window slides down, so a row costs 5 loads instead of 25
template <int N, int R>
__global__ __launch_bounds__(256) void k_upsample(const float* __restrict__ in, size_t in_stride, int w, int h,
                                                  const float* __restrict__ kernels, float* __restrict__ out,
                                                  size_t out_stride, int out_w, int out_h) {
  // a workgroup = 256 neighbouring columns: every output row gets 256 * N contiguous floats from it
  const int x = blockIdx.x * 256 + threadIdx.x;
  const int y0 = blockIdx.y * R;
  if (x >= w || y0 >= h) return;
  float win[25];
  int xs[5];
#pragma unroll
  for (int k = 0; k < 5; k++) xs[k] = mirror_idx(x - 2 + k, w);
#pragma unroll
  for (int ky = 0; ky < 4; ky++) {  // rows y0-2 .. y0+1 sit in window rows 1..4; the loop shifts before loading
    const float* __restrict__ row = in + (size_t)mirror_idx(y0 - 2 + ky, h) * in_stride;
#pragma unroll
    for (int kx = 0; kx < 5; kx++) win[(ky + 1) * 5 + kx] = row[xs[kx]];
  }
#pragma unroll 1
  for (int r = 0; r < R; r++) {
    const int y = y0 + r;
    if (y >= h) break;
#pragma unroll
    for (int t = 0; t < 20; t++) win[t] = win[t + 5];
    {
      const float* __restrict__ row = in + (size_t)mirror_idx(y + 2, h) * in_stride;
#pragma unroll
      for (int kx = 0; kx < 5; kx++) win[20 + kx] = row[xs[kx]];
    }
    float mn = win[0], mx = win[0];
#pragma unroll
    for (int t = 1; t < 25; t++) {
      mn = win[t] < mn ? win[t] : mn;
      mx = win[t] > mx ? win[t] : mx;
    }
#pragma unroll
    for (int oy = 0; oy < N; oy++) {
      float v[N];
#pragma unroll
      for (int ox = 0; ox < N; ox++) {
        const float* __restrict__ k = kernels + (oy * N + ox) * 25;
        float a0 = win[0] * k[0], a1 = win[1] * k[1], a2 = win[2] * k[2];
#pragma unroll
        for (int t = 3; t < 25; t += 3) {
          a0 = __builtin_fmaf(win[t], k[t], a0);
          if (t + 1 < 25) a1 = __builtin_fmaf(win[t + 1], k[t + 1], a1);
          if (t + 2 < 25) a2 = __builtin_fmaf(win[t + 2], k[t + 2], a2);
        }
        float q = (a0 + a1) + a2;
        q = q > mn ? q : mn;
        q = q < mx ? q : mx;
        v[ox] = q;
      }
      const int oyy = y * N + oy;
      if (oyy >= out_h) continue;
      float* __restrict__ dst = out + (size_t)oyy * out_stride + (size_t)x * N;
      if (x * N + N <= out_w) {
        if constexpr (N == 2) {
          *reinterpret_cast<float2*>(dst) = make_float2(v[0], v[1]);
        } else {
#pragma unroll
          for (int i = 0; i < N; i += 4) *reinterpret_cast<float4*>(dst + i) = make_float4(v[i], v[i + 1], v[i + 2], v[i + 3]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < N; i++)
          if (x * N + i < out_w) dst[i] = v[i];
      }
    }
  }
}

constexpr int kRows2 = 8, kRows4 = 4, kRows8 = 2;

}  // namespace

// kernels: n*n*25 floats on the device, [(oy*n + ox)*25 + ky*5 + kx]; out rows 16-byte aligned
void launch_upsample(hipStream_t s, int n, const float* in, size_t in_stride, int w, int h, const float* kernels,
                     float* out, size_t out_stride, int out_w, int out_h) {
  if (w <= 0 || h <= 0) return;
  const dim3 block(256);
  auto grid = [&](int rows) { return dim3((w + 255) / 256, (h + rows - 1) / rows); };
  if (n == 2)
    hipLaunchKernelGGL((k_upsample<2, kRows2>), grid(kRows2), block, 0, s, in, in_stride, w, h, kernels, out, out_stride,
                       out_w, out_h);
  else if (n == 4)
    hipLaunchKernelGGL((k_upsample<4, kRows4>), grid(kRows4), block, 0, s, in, in_stride, w, h, kernels, out, out_stride,
                       out_w, out_h);
  else
    hipLaunchKernelGGL((k_upsample<8, kRows8>), grid(kRows8), block, 0, s, in, in_stride, w, h, kernels, out, out_stride,
                       out_w, out_h);
}

}  // namespace jxlh
