// Workgroup-cooperative path for the large transforms (DCT64X64 ... DCT256X256,
// transform types 18..26).  Semantics: transform.rs:447-509 (reinterpreting_dct2d_{cy}_{cx}
// then idct2d_R_C), idct_large.rs:251-310 (recursive 1-D IDCT) and :387-501 (2-D drivers).
//
// A 256x256 varblock is 256 KiB per channel -- more than the CU's 160 KiB of LDS -- so the
// two separable passes run slab by slab:
//   pass 1  slab of LV lines (fixed v) x all C horizontal frequencies in LDS, 1-D IDCT_C along
//           u as log2(C/32) decimation sweeps + one register IDCT_32 per (line, leaf) + the
//           butterfly sweeps back up; result parked in the varblock's own output rectangle
//   pass 2  slab of LX pixel columns x all R rows from that rectangle, IDCT_R along v, in place
// The sweeps are elementwise over (pair, line) with `line` the fastest LDS index, so every
// ds access of a wave is contiguous.  The operation order equals the reference recursion
// (even half first, o[i] += o[i-1] on the *unmodified* odd inputs, w_i butterflies), so
// results are bit-identical to the oracle's FMA build.
#pragma once
#include "varblock_core.h"

namespace jxlh {

constexpr int kLargeThreads = 256;
constexpr int kLargeSlab = 4096;  // coefficients per slab

// 1-D IDCT of size N along i for L lines; data at X[i*Lp + line].  Ping-pongs between a and b
// and returns the buffer holding the result.  All threads of the workgroup must call.
template <int N>
__device__ float* lds_idct(float* __restrict__ a, float* __restrict__ b, int L, int Lp, int tid) {
  float* src = a;
  float* dst = b;
  // down-sweep: split every length-n sub-array into even | prefix-summed odd halves
  for (int n = N; n > 32; n >>= 1) {
    const int h = n / 2;
    for (int idx = tid; idx < (N / 2) * L; idx += kLargeThreads) {
      const int line = idx % L, p = idx / L;
      const int s = p / h, i = p % h;
      const int base = s * n;
      const float e = src[(base + 2 * i) * Lp + line];
      float o = src[(base + 2 * i + 1) * Lp + line];
      if (i > 0) {
        o += src[(base + 2 * i - 1) * Lp + line];
      } else {
        o *= kSqrt2;
      }
      dst[(base + i) * Lp + line] = e;
      dst[(base + h + i) * Lp + line] = o;
    }
    __syncthreads();
    float* t = src;
    src = dst;
    dst = t;
  }
  // leaves: register IDCT_32 per (line, leaf), in place
  for (int idx = tid; idx < (N / 32) * L; idx += kLargeThreads) {
    const int line = idx % L, leaf = idx / L;
    float x[32];
    float* p = src + (leaf * 32) * Lp + line;
#pragma unroll
    for (int j = 0; j < 32; j++) x[j] = p[j * Lp];
    idct1d<32, true>(x);
#pragma unroll
    for (int j = 0; j < 32; j++) p[j * Lp] = x[j];
  }
  __syncthreads();
  // up-sweep: out[i] = e[i] + w_i o[i], out[n-1-i] = e[i] - w_i o[i]
  auto sweep = [&](auto n_tag) {
    constexpr int n = decltype(n_tag)::value;
    if constexpr (n <= N) {
      constexpr int h = n / 2;
      for (int idx = tid; idx < (N / 2) * L; idx += kLargeThreads) {
        const int line = idx % L, p = idx / L;
        const int s = p / h, i = p % h;
        const int base = s * n;
        const float e = src[(base + i) * Lp + line];
        const float o = src[(base + h + i) * Lp + line];
        const float w = IdctW<n>::w[i];
        dst[(base + i) * Lp + line] = __builtin_fmaf(o, w, e);
        dst[(base + n - 1 - i) * Lp + line] = __builtin_fmaf(-o, w, e);
      }
      __syncthreads();
      float* t = src;
      src = dst;
      dst = t;
    }
  };
  sweep(std::integral_constant<int, 64>{});
  sweep(std::integral_constant<int, 128>{});
  sweep(std::integral_constant<int, 256>{});
  return src;
}

__device__ inline float* lds_idct_dyn(int n, float* a, float* b, int L, int Lp, int tid) {
  switch (n) {
    case 32: return lds_idct<32>(a, b, L, Lp, tid);
    case 64: return lds_idct<64>(a, b, L, Lp, tid);
    case 128: return lds_idct<128>(a, b, L, Lp, tid);
    default: return lds_idct<256>(a, b, L, Lp, tid);
  }
}

// One line (n samples at stride `st`) through the reinterpreting DCT, n in {4,8,16,32}.
__device__ inline void rdct_line(float* p, int n, int st, bool fused) {
  auto run = [&](auto n_tag, auto f_tag) {
    constexpr int NN = decltype(n_tag)::value;
    constexpr bool FF = decltype(f_tag)::value;
    float x[NN];
#pragma unroll
    for (int j = 0; j < NN; j++) x[j] = p[j * st];
    rdct1d<NN, FF>(x);
#pragma unroll
    for (int j = 0; j < NN; j++) p[j * st] = x[j];
  };
  using T = std::true_type;
  using F = std::false_type;
  switch (n) {
    case 4: fused ? run(std::integral_constant<int, 4>{}, T{}) : run(std::integral_constant<int, 4>{}, F{}); break;
    case 8: fused ? run(std::integral_constant<int, 8>{}, T{}) : run(std::integral_constant<int, 8>{}, F{}); break;
    case 16: fused ? run(std::integral_constant<int, 16>{}, T{}) : run(std::integral_constant<int, 16>{}, F{}); break;
    default: fused ? run(std::integral_constant<int, 32>{}, T{}) : run(std::integral_constant<int, 32>{}, F{}); break;
  }
}

// LLF-from-LF for a cy x cx patch (4..32 each) -> llf[r*mx + q], mn x mx.  scratch >= cy*cx.
__device__ inline void large_llf(const float* __restrict__ lf, int lf_stride, int cy, int cx, float* scratch,
                                 float* __restrict__ llf, int tid) {
  const bool fused = min(cy, cx) > 4;  // reinterpreting_dct2d.rs:584-600
  for (int i = tid; i < cy * cx; i += kLargeThreads) scratch[i] = lf[(i / cx) * lf_stride + (i % cx)];
  __syncthreads();
  if (cy < cx) {
    if (tid < cy) rdct_line(scratch + tid * cx, cx, 1, fused);
    __syncthreads();
    if (tid < cx) rdct_line(scratch + tid, cy, cx, fused);
    __syncthreads();
    for (int i = tid; i < cy * cx; i += kLargeThreads) llf[i] = scratch[i];
  } else {
    if (tid < cx) rdct_line(scratch + tid, cy, cx, fused);   // vertical, per column
    __syncthreads();
    if (tid < cy) rdct_line(scratch + tid * cx, cx, 1, fused);  // then along x, per row v
    __syncthreads();
    // transposed output: llf[u*cy + v] = scratch[v*cx + u]
    for (int i = tid; i < cy * cx; i += kLargeThreads) {
      const int u = i / cy, v = i % cy;
      llf[i] = scratch[v * cx + u];
    }
  }
  __syncthreads();
}

__device__ __forceinline__ float adjust_quant_bias_s(int q, float bias_c, float bias3) {  // group.rs:85-96
  const float quant = (float)q;
  const float adjusted = quant - bias3 / quant;
  return (q > -2 && q < 2) ? quant * bias_c : adjusted;
}

// idx -> (x, y) inside a W x H pixel region (W = 1 << wlog, both multiples of 8) such that
// consecutive threads touch consecutive memory: raster planes are x-major, the 8x8-tiled layout
// stores a block as x*8 + y (see FrameDev::tiled), so there the walk goes down the 8 rows of a
// block column first.
__device__ __forceinline__ void region_xy(bool tiled, int wlog, int idx, int& x, int& y) {
  if (tiled) {
    const int j = idx & 63, blk = idx >> 6;
    x = ((blk & ((1 << (wlog - 3)) - 1)) << 3) + (j >> 3);
    y = ((blk >> (wlog - 3)) << 3) + (j & 7);
  } else {
    x = idx & ((1 << wlog) - 1);
    y = idx >> wlog;
  }
}

// One channel of one large varblock.  coef(k) returns the dequantised coefficient at stored
// index k; lf points at the cy x cx LF patch (row pitch lf_stride); plane at the top-left
// output pixel (addressing given by `lay`).  lds: >= 2*(kLargeSlab + 256) + 1024 floats.
// All threads of the (256-thread) workgroup must call with identical arguments.
template <class CoefFn>
__device__ void large_varblock_channel(int type, CoefFn coef, const float* __restrict__ lf, int lf_stride,
                                       float* __restrict__ plane, const PixLayout lay, float* lds, int tid) {
  const int cx = covered_x(type), cy = covered_y(type);
  const int R = cy * 8, C = cx * 8;
  const bool wide = R < C;
  const int mxRC = max(R, C);
  const int mn = min(cy, cx), mx = max(cy, cx);
  float* bufA = lds;
  float* bufB = lds + (kLargeSlab + 256);
  float* llf = lds + 2 * (kLargeSlab + 256);
  large_llf(lf, lf_stride, cy, cx, bufA, llf, tid);
  // ---------------- pass 1: along u (size C) for lines v
  {
    const int LV = min(R, kLargeSlab / C);
    const int Lp = LV + 1;
    for (int v0 = 0; v0 < R; v0 += LV) {
      for (int idx = tid; idx < C * LV; idx += kLargeThreads) {
        int u, line;
        if (wide) {  // stored in[v*C + u]: u fastest in memory
          u = idx % C;
          line = idx / C;
        } else {     // stored in[u*R + v]: v fastest
          line = idx % LV;
          u = idx / LV;
        }
        const int v = v0 + line;
        const int k = wide ? v * C + u : u * R + v;
        const int kr = k / mxRC, kq = k % mxRC;
        // LLF overwrites the HF-decoded corner (transform.rs:450)
        bufA[u * Lp + line] = (kr < mn && kq < mx) ? llf[kr * mx + kq] : coef(k);
      }
      __syncthreads();
      float* res = lds_idct_dyn(C, bufA, bufB, LV, Lp, tid);
      // park tmp[x][v] at pixel (row v, col x) of the output rectangle
      const int clog = 31 - __clz(C);
      for (int idx = tid; idx < C * LV; idx += kLargeThreads) {
        int x, line;
        region_xy(lay.tiled, clog, idx, x, line);
        plane[lay.at(x, v0 + line)] = res[x * Lp + line];
      }
      __syncthreads();
    }
  }
  __threadfence_block();
  __syncthreads();
  // ---------------- pass 2: along v (size R) for pixel columns x
  {
    const int LX = min(C, kLargeSlab / R);
    const int Lp = LX + 1;
    for (int x0 = 0; x0 < C; x0 += LX) {
      const int xlog = 31 - __clz(LX);
      for (int idx = tid; idx < R * LX; idx += kLargeThreads) {
        int line, v;
        region_xy(lay.tiled, xlog, idx, line, v);
        bufA[v * Lp + line] = plane[lay.at(x0 + line, v)];
      }
      __syncthreads();
      float* res = lds_idct_dyn(R, bufA, bufB, LX, Lp, tid);
      for (int idx = tid; idx < R * LX; idx += kLargeThreads) {
        int line, y;
        region_xy(lay.tiled, xlog, idx, line, y);
        plane[lay.at(x0 + line, y)] = res[y * Lp + line];
      }
      __syncthreads();
    }
  }
  __threadfence_block();
  __syncthreads();
}

}  // namespace jxlh
