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
constexpr int kSlabIters = kLargeSlab / kLargeThreads;      // samples per thread and slab
constexpr int kPairIters = kLargeSlab / 2 / kLargeThreads;  // butterfly pairs per thread and sweep
constexpr int kQuadIters = kLargeSlab / 4 / kLargeThreads;  // quads per thread and two-level sweep
// Pairs (samples: twice as many) a thread has in flight at a time: every LDS / global read of a chunk is issued before
// the first dependent operation.  The whole slab at once (8) costs registers the IDCT_32 leaves need.
#ifndef JXLH_LARGE_CHUNK
#define JXLH_LARGE_CHUNK 2
#endif
constexpr int kChunk = JXLH_LARGE_CHUNK;
static_assert(kPairIters % kChunk == 0 && kQuadIters % kChunk == 0, "chunking");

// 1-D IDCT of size N along i for L = 1 << lL lines; data at X[i*Lp + line].  Ping-pongs between a and b
// and returns the buffer holding the result.  All threads of the workgroup must call.
// Every size here is a power of two: the index arithmetic is shifts and masks (with run-time divisors it was the
// bulk of the kernel's instructions -- ten integer divisions per sample and pass).
template <int N>
__device__ float* lds_idct(float* __restrict__ a, float* __restrict__ b, int lL, int Lp, int tid) {
  float* src = a;
  float* dst = b;
  // L <= 128 divides the workgroup size, so a thread keeps ONE line through a whole sweep and walks the pairs
  // p = p0, p0 + pstep, ...: addresses are affine in the iteration (the generic idx -> (line, pair) arithmetic was
  // most of this kernel's ~100 vector instructions per sample and pass)
  const int line = tid & ((1 << lL) - 1), p0 = tid >> lL, pstep = kLargeThreads >> lL;
  constexpr int P = N / 2;  // pairs per line
  // down-sweep: split every length-n sub-array into even | prefix-summed odd halves.  With n = 2h and p = s*h + i:
  // source rows 2p, 2p + 1 (and 2p - 1), destination rows p + s*h and p + s*h + h
  auto down = [&](auto n_tag) {
    constexpr int n = decltype(n_tag)::value;
    if constexpr (n <= N && n > 32) {
      constexpr int h = n / 2;
#pragma unroll 1
      for (int c0 = 0; c0 < kPairIters; c0 += kChunk) {
        if (p0 + c0 * pstep >= P) break;
        float e[kChunk], o[kChunk], om[kChunk];
#pragma unroll
        for (int it = 0; it < kChunk; it++) {
          const int p = p0 + (c0 + it) * pstep;
          const bool on = p < P;
          const float* q = src + (2 * p) * Lp + line;
          e[it] = on ? q[0] : 0.f;
          o[it] = on ? q[Lp] : 0.f;
          om[it] = (on && (p & (h - 1)) != 0) ? q[-Lp] : 0.f;
        }
#pragma unroll
        for (int it = 0; it < kChunk; it++) {
          const int p = p0 + (c0 + it) * pstep;
          if (p < P) {
            float* d = dst + (p + (p & ~(h - 1))) * Lp + line;  // row p + s*h
            d[0] = e[it];
            d[h * Lp] = (p & (h - 1)) != 0 ? o[it] + om[it] : o[it] * kSqrt2;
          }
        }
      }
      __syncthreads();
      float* t = src;
      src = dst;
      dst = t;
    }
  };
  // Two recursion levels in one LDS round trip (lengths n and n/2; q = n/4, quad j of sub-array s):
  //   E[2j] = x[4j], E[2j+1] = x[4j+2], O'[k] = x[2k+1] + x[2k-1] (k > 0) or x[1]*sqrt2   -- level n
  //   ee = E[2j], eo = E[2j+1] + E[2j-1] (j > 0) or E[1]*sqrt2, likewise oe / oo from O'   -- level n/2
  // i.e. exactly the operations of the two single sweeps (the level-n sums are rounded before level n/2 adds them),
  // with 7 reads + 4 writes per quad instead of 12 + 8, and one barrier instead of two.
  auto down2 = [&](auto n_tag) {
    constexpr int n = decltype(n_tag)::value, h = n / 2, q = n / 4;
    constexpr int Q = N / 4;  // quads per line
    const int g0 = tid >> lL, gstep = kLargeThreads >> lL;
#pragma unroll 1
    for (int c0 = 0; c0 < kQuadIters; c0 += kChunk) {
      if (g0 + c0 * gstep >= Q) break;
      float ee[kChunk], eo[kChunk], oe[kChunk], oo[kChunk];
#pragma unroll
      for (int it = 0; it < kChunk; it++) {
        const int g = g0 + (c0 + it) * gstep;
        const bool on = g < Q;
        const int j = g & (q - 1);
        const float* x = src + (4 * g) * Lp + line;  // row base + 4j of sub-array s: 4g = s*n + 4j
        const bool first = j == 0;
        const float x0 = on ? x[0] : 0.f, x1 = on ? x[Lp] : 0.f, x2 = on ? x[2 * Lp] : 0.f, x3 = on ? x[3 * Lp] : 0.f;
        const float xm1 = (on && !first) ? x[-Lp] : 0.f, xm2 = (on && !first) ? x[-2 * Lp] : 0.f,
                    xm3 = (on && !first) ? x[-3 * Lp] : 0.f;
        const float o_2j = first ? x1 * kSqrt2 : x1 + xm1;   // O'[2j]
        const float o_2j1 = x3 + x1;                          // O'[2j+1]
        const float o_2jm1 = xm1 + xm3;                       // O'[2j-1] (j > 0)
        ee[it] = x0;
        eo[it] = first ? x2 * kSqrt2 : x2 + xm2;
        oe[it] = o_2j;
        oo[it] = first ? o_2j1 * kSqrt2 : o_2j1 + o_2jm1;
      }
#pragma unroll
      for (int it = 0; it < kChunk; it++) {
        const int g = g0 + (c0 + it) * gstep;
        if (g < Q) {
          const int j = g & (q - 1), base = (g & ~(q - 1)) * 4;  // s * n
          float* d = dst + (base + j) * Lp + line;
          d[0] = ee[it];
          d[q * Lp] = eo[it];
          d[h * Lp] = oe[it];
          d[(h + q) * Lp] = oo[it];
        }
      }
    }
    __syncthreads();
    float* t = src;
    src = dst;
    dst = t;
  };
  if constexpr (N == 256) {
    down2(std::integral_constant<int, 256>{});
    down(std::integral_constant<int, 64>{});
  } else if constexpr (N == 128) {
    down2(std::integral_constant<int, 128>{});
  } else {
    down(std::integral_constant<int, 64>{});
  }
  // leaves: register IDCT_32 per (line, leaf), in place
  for (int leaf = p0; leaf < N / 32; leaf += pstep) {
    float x[32];
    float* p = src + (leaf * 32) * Lp + line;
#pragma unroll
    for (int j = 0; j < 32; j++) x[j] = p[j * Lp];
    idct1d<32, true>(x);
#pragma unroll
    for (int j = 0; j < 32; j++) p[j * Lp] = x[j];
  }
  __syncthreads();
  // up-sweep: out[i] = e[i] + w_i o[i], out[n-1-i] = e[i] - w_i o[i]; source rows p + s*h and p + s*h + h,
  // destination rows p + s*h and (s + 1)*n - 1 - i
  auto sweep = [&](auto n_tag) {
    constexpr int n = decltype(n_tag)::value;
    if constexpr (n <= N) {
      constexpr int h = n / 2;
#pragma unroll 1
      for (int c0 = 0; c0 < kPairIters; c0 += kChunk) {
        if (p0 + c0 * pstep >= P) break;
        float e[kChunk], o[kChunk];
#pragma unroll
        for (int it = 0; it < kChunk; it++) {
          const int p = p0 + (c0 + it) * pstep;
          const float* q = src + (p + (p & ~(h - 1))) * Lp + line;
          e[it] = p < P ? q[0] : 0.f;
          o[it] = p < P ? q[h * Lp] : 0.f;
        }
#pragma unroll
        for (int it = 0; it < kChunk; it++) {
          const int p = p0 + (c0 + it) * pstep;
          if (p < P) {
            const int i = p & (h - 1), sn = 2 * (p & ~(h - 1));  // s * n
            const float w = IdctW<n>::w[i];
            dst[(sn + i) * Lp + line] = __builtin_fmaf(o[it], w, e[it]);
            dst[(sn + n - 1 - i) * Lp + line] = __builtin_fmaf(-o[it], w, e[it]);
          }
        }
      }
      __syncthreads();
      float* t = src;
      src = dst;
      dst = t;
    }
  };
  // two butterfly levels (lengths n/2 then n) in one round trip: 4 reads + 4 writes per quad instead of 8 + 8
  auto sweep2 = [&](auto n_tag) {
    constexpr int n = decltype(n_tag)::value, h = n / 2, q = n / 4;
    constexpr int Q = N / 4;
    const int g0 = tid >> lL, gstep = kLargeThreads >> lL;
#pragma unroll 1
    for (int c0 = 0; c0 < kQuadIters; c0 += kChunk) {
      if (g0 + c0 * gstep >= Q) break;
      float ee[kChunk], eo[kChunk], oe[kChunk], oo[kChunk];
#pragma unroll
      for (int it = 0; it < kChunk; it++) {
        const int g = g0 + (c0 + it) * gstep;
        const bool on = g < Q;
        const int j = g & (q - 1), base = (g & ~(q - 1)) * 4;
        const float* x = src + (base + j) * Lp + line;
        ee[it] = on ? x[0] : 0.f;
        eo[it] = on ? x[q * Lp] : 0.f;
        oe[it] = on ? x[h * Lp] : 0.f;
        oo[it] = on ? x[(h + q) * Lp] : 0.f;
      }
#pragma unroll
      for (int it = 0; it < kChunk; it++) {
        const int g = g0 + (c0 + it) * gstep;
        if (g < Q) {
          const int j = g & (q - 1), base = (g & ~(q - 1)) * 4;
          const float wq = IdctW<h>::w[j];
          const float e_lo = __builtin_fmaf(eo[it], wq, ee[it]), e_hi = __builtin_fmaf(-eo[it], wq, ee[it]);  // E[j], E[h-1-j]
          const float o_lo = __builtin_fmaf(oo[it], wq, oe[it]), o_hi = __builtin_fmaf(-oo[it], wq, oe[it]);  // O[j], O[h-1-j]
          const float w_lo = IdctW<n>::w[j], w_hi = IdctW<n>::w[h - 1 - j];
          float* d = dst + base * Lp + line;
          d[j * Lp] = __builtin_fmaf(o_lo, w_lo, e_lo);
          d[(n - 1 - j) * Lp] = __builtin_fmaf(-o_lo, w_lo, e_lo);
          d[(h - 1 - j) * Lp] = __builtin_fmaf(o_hi, w_hi, e_hi);
          d[(h + j) * Lp] = __builtin_fmaf(-o_hi, w_hi, e_hi);
        }
      }
    }
    __syncthreads();
    float* t = src;
    src = dst;
    dst = t;
  };
  if constexpr (N == 256) {
    sweep(std::integral_constant<int, 64>{});
    sweep2(std::integral_constant<int, 256>{});
  } else if constexpr (N == 128) {
    sweep2(std::integral_constant<int, 128>{});
  } else {
    sweep(std::integral_constant<int, 64>{});
  }
  return src;
}

__device__ inline float* lds_idct_dyn(int n, float* a, float* b, int lL, int Lp, int tid) {
  switch (n) {
    case 32: return lds_idct<32>(a, b, lL, Lp, tid);
    case 64: return lds_idct<64>(a, b, lL, Lp, tid);
    case 128: return lds_idct<128>(a, b, lL, Lp, tid);
    default: return lds_idct<256>(a, b, lL, Lp, tid);
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

// Geometry of a large varblock's two passes: pass 1 works on slabs of LV lines (fixed v) x all C horizontal
// frequencies, pass 2 on slabs of LX pixel columns x all R rows; a slab is kLargeSlab samples (a 64x64 varblock is
// one slab per pass, a 256x256 one sixteen).
struct LargeGeom {
  int R, C, cx, cy, mn, mx, mxRC, lc, lr, lm;
  bool wide;
  int LV, llv, LX, xlog;
  __device__ explicit LargeGeom(int type) {
    cx = covered_x(type);
    cy = covered_y(type);
    R = cy * 8;
    C = cx * 8;
    wide = R < C;
    mxRC = max(R, C);
    mn = min(cy, cx);
    mx = max(cy, cx);
    lc = 31 - __clz(C);
    lr = 31 - __clz(R);
    lm = max(lc, lr);
    LV = min(R, kLargeSlab / C);
    llv = 31 - __clz(LV);
    LX = min(C, kLargeSlab / R);
    xlog = 31 - __clz(LX);
  }
  __device__ int slabs_per_pass() const { return R / LV; }  // == C / LX; 1 for the 64x32 / 32x64 half slabs
  // does the slab of lines [v0, v0 + LV) hold coefficients the LLF-from-LF corner overwrites (transform.rs:450)?
  __device__ bool slab_needs_llf(int v0) const { return v0 < (wide ? mn : mx); }
};

// Pass 1 of ONE slab: lines v0 .. v0 + LV of the horizontal IDCT, result parked at pixel (row v, col x) of the
// varblock's output rectangle.  coef(k) returns the dequantised coefficient at stored index k; llf (LDS) holds the
// mn x mx LLF corner when the slab needs it.  lds: >= 2 * (kLargeSlab + 256) floats.  All threads must call.
// Pass 1 of ONE slab: lines v0 .. v0 + LV of the horizontal IDCT, result parked at pixel (row v, col x) of the
// varblock's output rectangle.  coef(k) returns the dequantised coefficient at stored index k; llf (LDS) holds the
// mn x mx LLF corner when the slab needs it.  lds: >= 2 * (kLargeSlab + 256) floats.  All threads must call.
// (Compile-time specialisation on (transform length, lines per slab) was measured: 233-256 VGPRs with spills and
// no faster than this run-time geometry form at 128-181 VGPRs -- profiles/r02_e_large_path.txt.)
template <class CoefFn>
__device__ __forceinline__ void large_pass1_slab(const LargeGeom& g, int v0, CoefFn coef, const float* __restrict__ llf,
                                                 float* __restrict__ plane, const PixLayout lay, float* lds, int tid) {
  float* bufA = lds;
  float* bufB = lds + (kLargeSlab + 256);
  const int Lp = g.LV + 1, total = g.C << g.llv;
#pragma unroll 1
  for (int c0 = 0; c0 < kSlabIters; c0 += 2 * kChunk) {
    if (c0 * kLargeThreads >= total) break;  // half slabs
    float val[2 * kChunk];
    int at[2 * kChunk];
#pragma unroll
    for (int it = 0; it < 2 * kChunk; it++) {  // unrolled: the chunk's coefficient loads are all in flight together
      const int idx = (c0 + it) * kLargeThreads + tid;
      // wide: stored in[v*C + u], u fastest in memory; otherwise in[u*R + v], v fastest
      const int u = g.wide ? (idx & (g.C - 1)) : (idx >> g.llv), line = g.wide ? (idx >> g.lc) : (idx & (g.LV - 1));
      const int v = v0 + line;
      const int k = g.wide ? (v << g.lc) + u : (u << g.lr) + v;
      const int kr = k >> g.lm, kq = k & (g.mxRC - 1);
      at[it] = u * Lp + line;
      // LLF overwrites the HF-decoded corner (transform.rs:450)
      val[it] = (kr < g.mn && kq < g.mx) ? llf[kr * g.mx + kq] : coef(k);
    }
#pragma unroll
    for (int it = 0; it < 2 * kChunk; it++) bufA[at[it]] = val[it];
  }
  __syncthreads();
  float* res = lds_idct_dyn(g.C, bufA, bufB, g.llv, Lp, tid);
#pragma unroll 1
  for (int c0 = 0; c0 < kSlabIters; c0 += 2 * kChunk) {
    if (c0 * kLargeThreads >= total) break;
#pragma unroll
    for (int it = 0; it < 2 * kChunk; it++) {
      const int idx = (c0 + it) * kLargeThreads + tid;
      int x, line;
      region_xy(lay.tiled, g.lc, idx, x, line);
      plane[lay.at(x, v0 + line)] = res[x * Lp + line];
    }
  }
  __syncthreads();
}

// Pass 2 of ONE slab: pixel columns x0 .. x0 + LX, vertical IDCT in place in the output rectangle.
__device__ __forceinline__ void large_pass2_slab(const LargeGeom& g, int x0, float* __restrict__ plane,
                                                 const PixLayout lay, float* lds, int tid) {
  float* bufA = lds;
  float* bufB = lds + (kLargeSlab + 256);
  const int Lp = g.LX + 1, total = g.R << g.xlog;
#pragma unroll 1
  for (int c0 = 0; c0 < kSlabIters; c0 += 2 * kChunk) {
    if (c0 * kLargeThreads >= total) break;
    float val[2 * kChunk];
#pragma unroll
    for (int it = 0; it < 2 * kChunk; it++) {
      const int idx = (c0 + it) * kLargeThreads + tid;
      int line, v;
      region_xy(lay.tiled, g.xlog, idx, line, v);
      val[it] = plane[lay.at(x0 + line, v)];
    }
#pragma unroll
    for (int it = 0; it < 2 * kChunk; it++) {
      const int idx = (c0 + it) * kLargeThreads + tid;
      int line, v;
      region_xy(lay.tiled, g.xlog, idx, line, v);
      bufA[v * Lp + line] = val[it];
    }
  }
  __syncthreads();
  float* res = lds_idct_dyn(g.R, bufA, bufB, g.xlog, Lp, tid);
#pragma unroll 1
  for (int c0 = 0; c0 < kSlabIters; c0 += 2 * kChunk) {
    if (c0 * kLargeThreads >= total) break;
#pragma unroll
    for (int it = 0; it < 2 * kChunk; it++) {
      const int idx = (c0 + it) * kLargeThreads + tid;
      int line, y;
      region_xy(lay.tiled, g.xlog, idx, line, y);
      plane[lay.at(x0 + line, y)] = res[y * Lp + line];
    }
  }
  __syncthreads();
}

// One channel of one large varblock, both passes by one workgroup (stage hook; the frame path runs the passes as
// separate launches over slab units, k_vardct.hip).  lf points at the cy x cx LF patch (row pitch lf_stride); plane
// at the top-left output pixel (addressing given by `lay`).  lds: >= 2*(kLargeSlab + 256) + 1024 floats.
// All threads of the (256-thread) workgroup must call with identical arguments.
template <class CoefFn>
__device__ void large_varblock_channel(int type, CoefFn coef, const float* __restrict__ lf, int lf_stride,
                                       float* __restrict__ plane, const PixLayout lay, float* lds, int tid) {
  const LargeGeom g(type);
  float* llf = lds + 2 * (kLargeSlab + 256);
  large_llf(lf, lf_stride, g.cy, g.cx, lds, llf, tid);
  for (int v0 = 0; v0 < g.R; v0 += g.LV) large_pass1_slab(g, v0, coef, llf, plane, lay, lds, tid);
  __threadfence_block();
  __syncthreads();
  for (int x0 = 0; x0 < g.C; x0 += g.LX) large_pass2_slab(g, x0, plane, lay, lds, tid);
  __threadfence_block();
  __syncthreads();
}

}  // namespace jxlh
