// K4 RCT, K5 Palette (non-delta), K6 inverse Squeeze -- Modular transforms on whole i32
// planes (the GPU does not tile: neighbour-border plumbing of transforms/step.rs vanishes).
// All arithmetic is wrapping 32-bit, as in the reference's SIMD paths.
//
// Reference: rct.rs:14-157; palette.rs:24-199; squeeze.rs:107-141 (smooth_tendency_impl),
// :171-185 (unsqueeze_impl), :389-437 (hsqueeze), :576-644 (vsqueeze).
//
// Squeeze is a non-associative recurrence along the squeezed axis (the previous output b
// feeds the next tendency), so the only parallelism is across lines: one lane per line,
// loads software-pipelined ahead of the dependent chain.  Vertical steps are naturally
// coalesced (lanes = adjacent columns); horizontal steps walk rows (lanes = adjacent rows)
// and lean on L1/L2 for the 128-byte lines they share across iterations.
#include "jxlh_internal.h"

namespace jxlh {
namespace {

__device__ __forceinline__ int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
__device__ __forceinline__ int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }

template <int OP>
__device__ __forceinline__ void rct_op(int32_t v0, int32_t v1, int32_t v2, int32_t& w0, int32_t& w1, int32_t& w2) {
  w0 = v0;
  w1 = v1;
  w2 = v2;
  if constexpr (OP == 1) {
    w2 = wadd(v2, v0);
  } else if constexpr (OP == 2) {
    w1 = wadd(v1, v0);
  } else if constexpr (OP == 3) {
    w1 = wadd(v1, v0);
    w2 = wadd(v2, v0);
  } else if constexpr (OP == 4) {
    w1 = wadd(v1, wadd(v0, v2) >> 1);
  } else if constexpr (OP == 5) {
    const int32_t t2 = wadd(v0, v2);
    w1 = wadd(v1, wadd(v0, t2) >> 1);
    w2 = t2;
  } else if constexpr (OP == 6) {
    int32_t y = wsub(v0, v2 >> 1);
    const int32_t g = wadd(v2, y);
    y = wsub(y, v1 >> 1);
    w0 = wadd(y, v1);
    w1 = g;
    w2 = y;
  }
}

// perm: which output plane receives w0/w1/w2 (rct.rs:132-156)
template <int OP>
__global__ void k4_rct(int32_t* __restrict__ p0, int32_t* __restrict__ p1, int32_t* __restrict__ p2, size_t n,
                       int perm) {
  int32_t* o[3];
  switch (perm) {
    default:
    case 0: o[0] = p0; o[1] = p1; o[2] = p2; break;
    case 1: o[0] = p1; o[1] = p2; o[2] = p0; break;  // Gbr: out[1,2,0] = in[0,1,2]
    case 2: o[0] = p2; o[1] = p0; o[2] = p1; break;  // Brg
    case 3: o[0] = p0; o[1] = p2; o[2] = p1; break;  // Rbg
    case 4: o[0] = p1; o[1] = p0; o[2] = p2; break;  // Grb
    case 5: o[0] = p2; o[1] = p1; o[2] = p0; break;  // Bgr
  }
  const size_t nvec = n / 4;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    const int4 a = reinterpret_cast<const int4*>(p0)[i];
    const int4 b = reinterpret_cast<const int4*>(p1)[i];
    const int4 c = reinterpret_cast<const int4*>(p2)[i];
    int4 x, y, z;
    rct_op<OP>(a.x, b.x, c.x, x.x, y.x, z.x);
    rct_op<OP>(a.y, b.y, c.y, x.y, y.y, z.y);
    rct_op<OP>(a.z, b.z, c.z, x.z, y.z, z.z);
    rct_op<OP>(a.w, b.w, c.w, x.w, y.w, z.w);
    reinterpret_cast<int4*>(o[0])[i] = x;
    reinterpret_cast<int4*>(o[1])[i] = y;
    reinterpret_cast<int4*>(o[2])[i] = z;
  }
  // tail
  for (size_t i = nvec * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    int32_t x, y, z;
    rct_op<OP>(p0[i], p1[i], p2[i], x, y, z);
    o[0][i] = x;
    o[1][i] = y;
    o[2][i] = z;
  }
}

__constant__ int16_t kDeltaPalette[72][3] = {
#include "delta_palette.inc"
};

// get_palette_value (palette.rs:39-163)
__device__ __forceinline__ int32_t palette_value(const int32_t* __restrict__ palette, size_t pstride, int32_t index,
                                                 int c, int palette_size, int bit_depth) {
  if (index < 0) {
    if (c >= 3) return 0;
    uint32_t i = (uint32_t)(-(index + 1));
    i %= 1 + 2 * (72 - 1);
    int32_t r = kDeltaPalette[(i + 1) >> 1][c];
    if ((i & 1) == 0) r = -r;
    if (bit_depth > 8) r *= 1 << (bit_depth - 8);
    return r;
  }
  uint32_t i = (uint32_t)index;
  const uint32_t ps = (uint32_t)palette_size;
  if (i >= ps && i < ps + 64) {
    if (c >= 3) return 0;
    i -= ps;
    i >>= c * 2;
    const int sh = bit_depth > 3 ? bit_depth - 3 : 0;
    return (int32_t)(((uint64_t)(i % 4) * (uint64_t)((1u << bit_depth) - 1)) >> 2) + (1 << sh);
  } else if (i >= ps + 64) {
    if (c >= 3) return 0;
    i -= ps + 64;
    if (c == 1) i /= 5;
    if (c == 2) i /= 25;
    return (int32_t)(((uint64_t)(i % 5) * (uint64_t)((1u << bit_depth) - 1)) >> 2);
  }
  return palette[(size_t)c * pstride + i];
}

__global__ void k5_palette(const int32_t* __restrict__ index, size_t n, const int32_t* __restrict__ palette,
                           int num_colors, size_t pstride, int nb_channels, int bit_depth,
                           int32_t* __restrict__ out) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int32_t idx = index[i];
    for (int c = 0; c < nb_channels; c++)
      out[(size_t)c * n + i] = palette_value(palette, pstride, idx, c, num_colors, bit_depth);
  }
}

// smooth_tendency_impl (squeeze.rs:107-141), a = prev, b = avg, c = next_avg
__device__ __forceinline__ int32_t smooth_tendency(int32_t a, int32_t b, int32_t c) {
  const int32_t a_b = wsub(a, b), b_c = wsub(b, c), a_c = wsub(a, c);
  const int32_t abs_a_b = a_b < 0 ? wsub(0, a_b) : a_b;
  const int32_t abs_b_c = b_c < 0 ? wsub(0, b_c) : b_c;
  const int32_t abs_a_c = a_c < 0 ? wsub(0, a_c) : a_c;
  const bool non_monotonic = (a_b ^ b_c) < 0;
  const bool skip = (b_c != 0) && (a_b != 0) && non_monotonic;
  const int32_t abs_a_b_3 = __mulhi(abs_a_b, 0x55555556);
  int32_t x = wadd(wadd(2, abs_a_c), abs_a_b_3) >> 2;
  const int32_t two_ab = (int32_t)((uint32_t)abs_a_b << 1);
  if (x > wadd(two_ab, x & 1)) x = wadd(two_ab, 1);
  const int32_t two_bc = (int32_t)((uint32_t)abs_b_c << 1);
  if (wadd(x, x & 1) > two_bc) x = two_bc;
  if (skip) x = 0;
  return a_c < 0 ? wsub(0, x) : x;
}

// unsqueeze_impl (squeeze.rs:171-185)
__device__ __forceinline__ void unsqueeze(int32_t avg, int32_t res, int32_t next_avg, int32_t prev, int32_t& a,
                                          int32_t& b) {
  const int32_t diff = wadd(res, smooth_tendency(prev, avg, next_avg));
  const int32_t sign = (int32_t)((uint32_t)diff >> 31);
  const int32_t diff_2 = wadd(diff, sign) >> 1;
  a = wadd(avg, diff_2);
  b = wsub(a, diff);
}

// One lane per line.  Element i of line l lives at p[l*line_pitch + i*elem_pitch].
// n_res = floor(n_out/2) residuals per line, n_avg = n_out - n_res averages.
__global__ void k6_unsqueeze(const int32_t* __restrict__ avg, size_t avg_lp, size_t avg_ep,
                             const int32_t* __restrict__ res, size_t res_lp, size_t res_ep, int32_t* __restrict__ out,
                             size_t out_lp, size_t out_ep, int n_lines, int n_out) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= n_lines) return;
  const int32_t* __restrict__ a = avg + (size_t)l * avg_lp;
  const int32_t* __restrict__ r = res + (size_t)l * res_lp;
  int32_t* __restrict__ o = out + (size_t)l * out_lp;
  const int w = n_out / 2;
  if (w == 0) {  // single output sample (squeeze.rs:468-476, :672-675)
    o[0] = a[0];
    return;
  }
  const bool has_tail = n_out & 1;
  int32_t cur = a[0];
  int32_t prev_b = cur;  // first `prev` is avg[0] (squeeze.rs:411-414, :591-594)
  constexpr int U = 8;
  int i = 0;
  // main body: next_avg = avg[i+1] exists for i < w-1 (or i < w with a tail)
  const int n_main = has_tail ? w : w - 1;
  for (; i + U <= n_main; i += U) {
    int32_t na[U], rr[U];
#pragma unroll
    for (int k = 0; k < U; k++) {
      na[k] = a[(size_t)(i + k + 1) * avg_ep];
      rr[k] = r[(size_t)(i + k) * res_ep];
    }
#pragma unroll
    for (int k = 0; k < U; k++) {
      int32_t va, vb;
      unsqueeze(cur, rr[k], na[k], prev_b, va, vb);
      o[(size_t)(2 * (i + k)) * out_ep] = va;
      o[(size_t)(2 * (i + k) + 1) * out_ep] = vb;
      prev_b = vb;
      cur = na[k];
    }
  }
  for (; i < n_main; i++) {
    const int32_t na = a[(size_t)(i + 1) * avg_ep];
    int32_t va, vb;
    unsqueeze(cur, r[(size_t)i * res_ep], na, prev_b, va, vb);
    o[(size_t)(2 * i) * out_ep] = va;
    o[(size_t)(2 * i + 1) * out_ep] = vb;
    prev_b = vb;
    cur = na;
  }
  if (!has_tail) {  // last pair: next_avg = avg itself (squeeze.rs:423-433, :608-616)
    int32_t va, vb;
    unsqueeze(cur, r[(size_t)(w - 1) * res_ep], cur, prev_b, va, vb);
    o[(size_t)(2 * w - 2) * out_ep] = va;
    o[(size_t)(2 * w - 1) * out_ep] = vb;
  } else {  // odd size: trailing average is copied (squeeze.rs:434-437, :641-643)
    o[(size_t)(2 * w) * out_ep] = cur;
  }
}

}  // namespace

void launch_rct(hipStream_t s, int32_t* p0, int32_t* p1, int32_t* p2, size_t n, int op, int perm) {
  if (n == 0) return;
  const unsigned grid = (unsigned)min((size_t)8192, (n / 4 + 255) / 256 + 1);
  switch (op) {
    case 0: hipLaunchKernelGGL(k4_rct<0>, dim3(grid), dim3(256), 0, s, p0, p1, p2, n, perm); break;
    case 1: hipLaunchKernelGGL(k4_rct<1>, dim3(grid), dim3(256), 0, s, p0, p1, p2, n, perm); break;
    case 2: hipLaunchKernelGGL(k4_rct<2>, dim3(grid), dim3(256), 0, s, p0, p1, p2, n, perm); break;
    case 3: hipLaunchKernelGGL(k4_rct<3>, dim3(grid), dim3(256), 0, s, p0, p1, p2, n, perm); break;
    case 4: hipLaunchKernelGGL(k4_rct<4>, dim3(grid), dim3(256), 0, s, p0, p1, p2, n, perm); break;
    case 5: hipLaunchKernelGGL(k4_rct<5>, dim3(grid), dim3(256), 0, s, p0, p1, p2, n, perm); break;
    default: hipLaunchKernelGGL(k4_rct<6>, dim3(grid), dim3(256), 0, s, p0, p1, p2, n, perm); break;
  }
}

void launch_palette(hipStream_t s, const int32_t* index, size_t n, const int32_t* palette, int num_colors,
                    size_t palette_stride, int nb_channels, int bit_depth, int32_t* out) {
  if (n == 0) return;
  const unsigned grid = (unsigned)min((size_t)8192, (n + 255) / 256);
  hipLaunchKernelGGL(k5_palette, dim3(grid), dim3(256), 0, s, index, n, palette, num_colors, palette_stride,
                     nb_channels, bit_depth, out);
}

void launch_unsqueeze(hipStream_t s, int horizontal, const int32_t* avg, size_t avg_stride, const int32_t* res,
                      size_t res_stride, uint32_t out_w, uint32_t out_h, int32_t* out, size_t out_stride) {
  if (out_w == 0 || out_h == 0) return;
  if (horizontal) {
    const int n_lines = (int)out_h;
    hipLaunchKernelGGL(k6_unsqueeze, dim3((n_lines + 63) / 64), dim3(64), 0, s, avg, avg_stride, (size_t)1, res,
                       res_stride, (size_t)1, out, out_stride, (size_t)1, n_lines, (int)out_w);
  } else {
    const int n_lines = (int)out_w;
    hipLaunchKernelGGL(k6_unsqueeze, dim3((n_lines + 63) / 64), dim3(64), 0, s, avg, (size_t)1, avg_stride, res,
                       (size_t)1, res_stride, out, (size_t)1, out_stride, n_lines, (int)out_h);
  }
}

}  // namespace jxlh
