// Wave-level path for the large transforms (DCT64X64 ... DCT256X256 and the 64x32 / 32x64 pair, transform types
// 18..26).  Semantics: transform.rs:447-509 (reinterpreting_dct2d_{cy}_{cx} then idct2d_R_C), idct_large.rs:251-310
// (recursive 1-D IDCT) and :387-501 (2-D drivers).
//
// A 256x256 varblock is 256 KiB per channel -- more than a CU's LDS -- so the two separable passes run slab by slab,
// a slab being 4096 samples = what ONE WAVEFRONT holds as 64 registers per lane:
//   pass 1  slab = LV lines (fixed v) x all C horizontal frequencies; 1-D IDCT_C along u; result parked in the
//           varblock's own output rectangle (it stays in L2 / Infinity Cache until pass 2)
//   pass 2  slab = LX pixel columns x all R rows from that rectangle; IDCT_R along v; in place
// Types below 256 pixels fit a workgroup's four wave tiles: there pass 2 reads the columns from LDS
// (wave_large_pass2_lds, round 4) and the rectangle is written once.
// Round 3 rewrite (profiles/r03_d_large_path.txt).  Round 2 ran the 1-D transforms as LDS sweeps of a 256-thread
// workgroup: ~13 LDS accesses and 7 barriers per sample and pass, ~100 vector instructions per sample.  Now the
// recursion idct_N = butterfly(idct_{N/2}(even), idct_{N/2}(prefix-summed odd)) is cut at length 64:
//   * a lane owns one LEAF of one line: a length-64 sub-transform whose inputs it gathers straight from the staged
//     line -- the decimation levels above the leaf are index arithmetic plus the odd halves' neighbour sums
//     (EE[j] = t[4j], EO[j] = t[4j+2] + t[4j-2], OE[j] = t[4j+1] + t[4j-1], OO[j] = (t[4j+3] + t[4j+1]) + (t[4j-1] +
//     t[4j-3]) for N = 256, the first element of an odd half times sqrt 2 instead) -- and transforms it in registers
//     (idct1d<64>, the same code the 8..32 sizes use);
//   * the N / 64 leaves of a line sit in adjacent lanes, so the butterflies back up (out[i] = e[i] + w_i o[i],
//     out[n-1-i] = e[i] - w_i o[i]) are lane exchanges inside a quad (DPP quad_perm), no memory;
//   * one LDS tile per wave (the staged lines; results written back in place for the coalesced store), only
//     wave-scope synchronisation: 1 LDS write + ~2.3 reads + 1 write + 1 read per sample and pass, no barrier.
// The operation order equals the reference recursion (even half first, o[i] += o[i-1] on the *unmodified* odd inputs,
// level-n sums rounded before level n/2 adds them, w_i butterflies with FMA), so results are bit-identical to the
// oracle's FMA build.
#pragma once
#include "varblock_core.h"

namespace jxlh {

constexpr int kLargeThreads = 256;                 // 4 independent waves per workgroup
constexpr int kLargeWaves = kLargeThreads / 64;
constexpr int kLargeSlab = 4096;                   // samples per slab = per wave
constexpr int kLargeTile = 64 * 65;
#ifndef JXLH_LARGE_BULK
#define JXLH_LARGE_BULK 8  // rounds of raw coefficient loads in flight per lane in pass 1 (a slab is 16, a half slab 8)
#endif                // floats of LDS per wave: lines x (N + N/64) for every N >= 64

__device__ __forceinline__ float dpp_xor1(float v) {  // value of lane ^ 1
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_xor2(float v) {  // value of lane ^ 2
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
}

// line pitch of the tile for transform length N (floats): N + N/64 (>= 64), N + 1 (32).  With K = N/64 lanes per
// line the 32 lanes of a ds_read_b32 group (32/K lines x K leaves, leaf offsets distinct mod K) hit 32 banks.
__host__ __device__ constexpr int large_pitch(int n) { return n >= 64 ? n + n / 64 : n + 1; }
// where sample x of a transformed line sits: every run of 64 results is shifted by one word, so that the K leaves of
// a line (which write x = base_leaf + r, base_leaf a multiple of 64, in the same instruction) use different banks
__host__ __device__ constexpr int large_pos(int x) { return x + (x >> 6); }

// One butterfly level across lane pairs.  in: x[r] = this lane's half-transform H_b[r] (b = 0: the even half E, b = 1:
// the odd half O, `odd` says which this lane holds), partner(v) = the other half's lane.  With W = IdctW<2*64*...>:
//   lane E keeps out[i]       = E[i] + w_i O[i]                     at register i
//   lane O keeps out[n-1-i]   = E[i] - w_i O[i]  (n - 1 - i = n/2 + (63 - i) relative to its run)  at register 63 - i
// so both lanes end with 64 results in ascending order.  woff: the pair's first weight index (lane dependent for the
// second level of N = 256, where the two pairs of a quad cover i in [0, 64) and [64, 128)).
template <int WN, class Partner>
__device__ __forceinline__ void leaf_butterfly(float (&x)[64], bool odd, bool whi, Partner partner) {
  // in place, registers r and 63 - r together (each is the other's source in the odd lane): two temporaries live
#pragma unroll
  for (int r = 0; r < 32; r++) {
    constexpr int kHalf = WN / 2;
    const int m = 63 - r;
    float w_r = IdctW<WN>::w[r], w_m = IdctW<WN>::w[m];
    if constexpr (WN == 256) {  // the quad's second pair works on i in [64, 128)
      w_r = whi ? IdctW<WN>::w[(64 + r) % kHalf] : w_r;
      w_m = whi ? IdctW<WN>::w[(64 + m) % kHalf] : w_m;
    }
    const float xr = x[r], xm = x[m];
    const float oth_r = partner(xr), oth_m = partner(xm);
    // lane E: out[i] = e + w o with e = own, o = partner's, i = its register; lane O: register t receives
    // out[n - 1 - (63 - t)] = e - w o computed from source index 63 - t (e = partner's, o = own)
    const float lo_r = __builtin_fmaf(oth_r, w_r, xr), lo_m = __builtin_fmaf(oth_m, w_m, xm);
    const float hi_r = __builtin_fmaf(-xm, w_m, oth_m), hi_m = __builtin_fmaf(-xr, w_r, oth_r);
    x[r] = odd ? hi_r : lo_r;
    x[m] = odd ? hi_m : lo_m;
  }
}

// 1-D IDCT of length N of the lines staged in `tile` (line l at tile + l * large_pitch(N), natural order), all 64
// lanes: lane = line * K + leaf.  Results replace the lines in place, sample x at large_pos(x).  Lines the caller did
// not stage compute on whatever the tile holds (the caller does not store them).
template <int N>
__device__ __forceinline__ void wave_idct_lines(float* __restrict__ tile, int lane) {
  constexpr int P = large_pitch(N);
  if constexpr (N == 32) {
    float x[32];
    float* t = tile + lane * P;
#pragma unroll
    for (int j = 0; j < 32; j++) x[j] = t[j];
    idct1d<32, true>(x);
    wave_sync();
#pragma unroll
    for (int j = 0; j < 32; j++) t[j] = x[j];
  } else {
    constexpr int K = N / 64;
    const int leaf = lane & (K - 1), line = lane / K;
    float* t = tile + line * P;
    float x[64];
    if constexpr (K == 1) {
#pragma unroll
      for (int j = 0; j < 64; j++) x[j] = t[j];
    } else if constexpr (K == 2) {
      // E[j] = t[2j];  O'[j] = t[2j+1] + t[2j-1], O'[0] = t[1] * sqrt2      (idct_large.rs:284-296)
      const bool odd = leaf != 0;
      const float* p = t + leaf;
#pragma unroll
      for (int j = 0; j < 64; j++) {
        const float a = p[2 * j];
        const float b = t[j ? 2 * j - 1 : 1];
        x[j] = odd ? (j ? a + b : a * kSqrt2) : a;
      }
    } else {
      // two decimation levels at once (see the header comment); leaf = 2 * (odd at length 256) + (odd at length 128)
      const bool o128 = (leaf & 1) != 0, o256 = (leaf & 2) != 0;
      const int c0 = o256 ? (o128 ? 3 : 1) : (o128 ? 2 : 0);  // EE 0, EO 2, OE 1, OO 3
      const int c1 = o256 ? -1 : -2;                          // EO: t[4j-2]; OE, OO: t[4j-1]
      const float* p0 = t + c0;
      const float* p1 = t + c1;
      const bool single = leaf == 0, quad = leaf == 3;
#pragma unroll
      for (int j = 0; j < 64; j++) {
        const float a = p0[4 * j];  // EE t[4j], EO t[4j+2], OE t[4j+1], OO t[4j+3]
        if (j == 0) {
          const float b = t[1];     // OO: (t[3] + t[1]) * sqrt2
          x[0] = single ? a : (quad ? (a + b) * kSqrt2 : a * kSqrt2);
        } else {
          const float b = p1[4 * j];      // EO t[4j-2], OE / OO t[4j-1]
          const float c = t[4 * j + 1];   // OO only
          const float d = t[4 * j - 3];   // OO only
          x[j] = single ? a : (quad ? (a + c) + (b + d) : a + b);
        }
      }
    }
    idct1d<64, true>(x);
    int base = 0;
    if constexpr (K == 2) {
      leaf_butterfly<128>(x, leaf != 0, false, dpp_xor1);
      base = leaf * 64;
    } else if constexpr (K == 4) {
      leaf_butterfly<128>(x, (leaf & 1) != 0, false, dpp_xor1);
      // now: lanes with leaf bit 0 clear hold H[0..64), set hold H[64..128), H = E (bit 1 clear) or O (bit 1 set)
      leaf_butterfly<256>(x, (leaf & 2) != 0, (leaf & 1) != 0, dpp_xor2);
      // EE: out[0..64)  EO: out[64..128)  OE: out[192..256)  OO: out[128..192)
      base = leaf == 0 ? 0 : leaf == 1 ? 64 : leaf == 2 ? 192 : 128;
    }
    wave_sync();  // every lane has its inputs in registers before any result lands on top of them
    float* d = t + large_pos(base);
#pragma unroll
    for (int r = 0; r < 64; r++) d[r] = x[r];
  }
  wave_sync();
}

__device__ __forceinline__ void wave_idct_lines_dyn(int n, float* tile, int lane) {
  switch (n) {
    case 32: wave_idct_lines<32>(tile, lane); break;
    case 64: wave_idct_lines<64>(tile, lane); break;
    case 128: wave_idct_lines<128>(tile, lane); break;
    default: wave_idct_lines<256>(tile, lane); break;
  }
}

// One line (n samples at stride `st`) through the reinterpreting DCT, n in {4,8,16,32}.
__device__ __forceinline__ void rdct_line(float* p, int n, int st, bool fused) {
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

// LLF-from-LF of a cy x cx patch (4..32 each) by ONE wave: out[r * mx + q], mn x mx.  scratch: >= cy * cx floats of
// LDS; `out` may be LDS or global memory.
__device__ __forceinline__ void wave_large_llf(const float* __restrict__ lf, int lf_stride, int cy, int cx, float* scratch,
                                      float* __restrict__ out, int lane) {
  const bool fused = min(cy, cx) > 4;  // reinterpreting_dct2d.rs:584-600
  for (int i = lane; i < cy * cx; i += 64) scratch[i] = lf[(i / cx) * lf_stride + (i % cx)];
  wave_sync();
  if (cy < cx) {
    if (lane < cy) rdct_line(scratch + lane * cx, cx, 1, fused);
    wave_sync();
    if (lane < cx) rdct_line(scratch + lane, cy, cx, fused);
    wave_sync();
    for (int i = lane; i < cy * cx; i += 64) out[i] = scratch[i];
  } else {
    if (lane < cx) rdct_line(scratch + lane, cy, cx, fused);   // vertical, per column
    wave_sync();
    if (lane < cy) rdct_line(scratch + lane * cx, cx, 1, fused);  // then along x, per row v
    wave_sync();
    // transposed output: out[u * cy + v] = scratch[v * cx + u]
    for (int i = lane; i < cy * cx; i += 64) {
      const int u = i / cy, v = i % cy;
      out[i] = scratch[v * cx + u];
    }
  }
  wave_sync();
}

// Geometry of a large varblock's two passes: pass 1 works on slabs of LV lines (fixed v) x all C horizontal
// frequencies, pass 2 on slabs of LX pixel columns x all R rows; a slab is kLargeSlab samples (a 64x64 varblock is one
// slab per pass, a 256x256 one sixteen; 64x32 / 32x64 are half a slab).
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

// ---- tile <-> pixel rectangle (rows y0 .. y0 + nrows, columns x0 .. x0 + ncols of the varblock's rectangle, both
// multiples of 8).  ROWS_ARE_LINES: the tile's lines are pixel rows (pass 1: line = v, sample = x), otherwise pixel
// columns (pass 2: line = x, sample = y).  POS: the tile holds transformed lines (sample s at large_pos(s)).
template <bool ROWS_ARE_LINES, bool POS>
__device__ __forceinline__ int tile_at(int P, int x, int y) {
  const int line = ROWS_ARE_LINES ? y : x, s = ROWS_ARE_LINES ? x : y;
  return line * P + (POS ? large_pos(s) : s);
}

template <bool ROWS_ARE_LINES>
__device__ __forceinline__ void wave_tile_store(const float* __restrict__ tile, int P, float* __restrict__ plane,
                                                const PixLayout lay, int x0, int y0, int ncols, int nrows, int lane) {
  if (lay.tiled) {
    // a 16-byte piece = 4 rows of one pixel column of an 8x8 block (memory order inside a block: x * 8 + y)
    const int nbx = ncols >> 3, total = (ncols * nrows) >> 2;
    for (int f = lane; f < total; f += 64) {
      const int blk = f >> 4, x = (f & 15) >> 1, yq = f & 1;
      const int bx = blk % nbx, by = blk / nbx;
      const int col = bx * 8 + x, row = by * 8 + yq * 4;
      float4 v;
      v.x = tile[tile_at<ROWS_ARE_LINES, true>(P, col, row)];
      v.y = tile[tile_at<ROWS_ARE_LINES, true>(P, col, row + 1)];
      v.z = tile[tile_at<ROWS_ARE_LINES, true>(P, col, row + 2)];
      v.w = tile[tile_at<ROWS_ARE_LINES, true>(P, col, row + 3)];
      *reinterpret_cast<float4*>(plane + lay.at(x0 + col, y0 + row)) = v;
    }
  } else {
    const int total = ncols * nrows;
    for (int f = lane; f < total; f += 64) {
      const int col = f % ncols, row = f / ncols;
      plane[lay.at(x0 + col, y0 + row)] = tile[tile_at<ROWS_ARE_LINES, true>(P, col, row)];
    }
  }
}

// pass-2 staging: pixel columns x0 .. x0 + ncols, all nrows rows -> tile lines = columns, natural order
__device__ __forceinline__ void wave_tile_load_columns(float* __restrict__ tile, int P, const float* __restrict__ plane,
                                                       const PixLayout lay, int x0, int ncols, int nrows, int lane) {
  if (lay.tiled) {
    const int nbx = ncols >> 3, total = (ncols * nrows) >> 2;
    constexpr int kInFlight = 8;  // 16-byte loads in flight per lane (a full slab is 16)
    for (int f0 = 0; f0 < total; f0 += kInFlight * 64) {
      float4 v[kInFlight];
#pragma unroll
      for (int k = 0; k < kInFlight; k++) {
        const int f = f0 + k * 64 + lane;
        const int blk = f >> 4, x = (f & 15) >> 1, yq = f & 1;
        const int bx = blk % nbx, by = blk / nbx;
        v[k] = f < total ? *reinterpret_cast<const float4*>(plane + lay.at(x0 + bx * 8 + x, by * 8 + yq * 4))
                         : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int k = 0; k < kInFlight; k++) {
        const int f = f0 + k * 64 + lane;
        if (f < total) {
          const int blk = f >> 4, x = (f & 15) >> 1, yq = f & 1;
          const int bx = blk % nbx, by = blk / nbx;
          float* d = tile + (bx * 8 + x) * P + by * 8 + yq * 4;
          d[0] = v[k].x;
          d[1] = v[k].y;
          d[2] = v[k].z;
          d[3] = v[k].w;
        }
      }
    }
  } else {
    const int total = ncols * nrows;
    for (int f = lane; f < total; f += 64) {
      const int col = f % ncols, row = f / ncols;
      tile[col * P + row] = plane[lay.at(x0 + col, row)];
    }
  }
}

// Pass 1 of ONE slab by ONE wave: lines v0 .. v0 + LV of the horizontal IDCT, result parked at pixel (row v, col x) of
// the varblock's output rectangle.  coef4(k) returns the four dequantised coefficients at stored indices k .. k + 3
// (k a multiple of 4); llf_at(i) the LLF corner value i = kr * mx + kq (only called where the corner applies).
// tile: kLargeTile floats of LDS owned by this wave.
template <class Coef4, class LlfAt>
__device__ __forceinline__ void wave_large_pass1_stage(const LargeGeom& g, int v0, Coef4 coef4, LlfAt llf_at, float* tile,
                                                       int lane) {
  const int P = large_pitch(g.C), total4 = (g.C << g.llv) >> 2;
  const bool corner = g.slab_needs_llf(v0);
#pragma unroll 1
  for (int f0 = 0; f0 < total4; f0 += 4 * 64) {
    float4 val[4];
    int at[4], step[4];
#pragma unroll
    for (int it = 0; it < 4; it++) {  // unrolled: the chunk's coefficient loads are all in flight together
      const int idx = (f0 + it * 64 + lane) * 4;
      // wide: stored in[v*C + u], u fastest in memory; otherwise in[u*R + v], v fastest
      const int u = g.wide ? (idx & (g.C - 1)) : (idx >> g.llv), line = g.wide ? (idx >> g.lc) : (idx & (g.LV - 1));
      const int v = v0 + line;
      const int k = g.wide ? (v << g.lc) + u : (u << g.lr) + v;
      at[it] = line * P + u;
      step[it] = g.wide ? 1 : P;  // the four values: consecutive u (wide) or consecutive lines
      float4 c = coef4(k);
      if (corner) {  // LLF overwrites the HF-decoded corner (transform.rs:450)
        const int kr = k >> g.lm, kq = k & (g.mxRC - 1);
        if (kr < g.mn && kq < g.mx) {
          c.x = llf_at(kr * g.mx + kq);
          if (kq + 1 < g.mx) c.y = llf_at(kr * g.mx + kq + 1);
          if (kq + 2 < g.mx) c.z = llf_at(kr * g.mx + kq + 2);
          if (kq + 3 < g.mx) c.w = llf_at(kr * g.mx + kq + 3);
        }
      }
      val[it] = c;
    }
#pragma unroll
    for (int it = 0; it < 4; it++) {
      if (f0 + it * 64 + lane < total4) {
        float* d = tile + at[it];
        d[0] = val[it].x;
        d[step[it]] = val[it].y;
        d[2 * step[it]] = val[it].z;
        d[3 * step[it]] = val[it].w;
      }
    }
  }
  wave_sync();
  wave_idct_lines_dyn(g.C, tile, lane);
}

// The same with the slab's raw coefficient loads ALL in flight before the first is used (16 x 16 bytes per lane and
// source array): at two waves per SIMD -- what the LDS tiles allow -- the wave's own loads are the only latency cover it
// has, and the registers are there (256 per wave).  coef.load(k) issues the loads of coefficients k .. k + 3,
// coef.finish(k, raw) dequantises them (its table reads hit L2).
template <class Coef>
__device__ __forceinline__ void wave_large_pass1_stage_bulk(const LargeGeom& g, int v0, const Coef& coef, float* tile, int lane) {
  const int P = large_pitch(g.C), total4 = (g.C << g.llv) >> 2;  // 1024, or 512 for the 64x32 / 32x64 half slabs
  const bool corner = g.slab_needs_llf(v0);
  auto stored_index = [&](int r, int* at) {
    const int idx = (r * 64 + lane) * 4;
    // wide: stored in[v*C + u], u fastest in memory; otherwise in[u*R + v], v fastest
    const int u = g.wide ? (idx & (g.C - 1)) : (idx >> g.llv), line = g.wide ? (idx >> g.lc) : (idx & (g.LV - 1));
    const int v = v0 + line;
    *at = line * P + u;
    return g.wide ? (v << g.lc) + u : (u << g.lr) + v;
  };
  const int step = g.wide ? 1 : P;  // the four values: consecutive u (wide) or consecutive lines
  const int rounds = total4 >> 6;   // 16, or 8 for the half slabs
#pragma unroll 1
  for (int r0 = 0; r0 < rounds; r0 += JXLH_LARGE_BULK) {
    typename Coef::Raw raw[JXLH_LARGE_BULK];
#pragma unroll
    for (int r = 0; r < JXLH_LARGE_BULK; r++) {
      int at;
      raw[r] = coef.load(stored_index(r0 + r, &at));
    }
#pragma unroll
    for (int r = 0; r < JXLH_LARGE_BULK; r++) {
      // the table reads of finish() stay behind the raw loads, 4 rounds of them at a time
      if (r % 4 == 0) __builtin_amdgcn_sched_barrier(0);
      int at;
      const int k = stored_index(r0 + r, &at);
      float4 c = coef.finish(k, raw[r]);
      if (corner) {  // LLF overwrites the HF-decoded corner (transform.rs:450)
        const int kr = k >> g.lm, kq = k & (g.mxRC - 1);
        if (kr < g.mn && kq < g.mx) {
          c.x = coef.llf_at(kr * g.mx + kq);
          if (kq + 1 < g.mx) c.y = coef.llf_at(kr * g.mx + kq + 1);
          if (kq + 2 < g.mx) c.z = coef.llf_at(kr * g.mx + kq + 2);
          if (kq + 3 < g.mx) c.w = coef.llf_at(kr * g.mx + kq + 3);
        }
      }
      float* d = tile + at;
      d[0] = c.x;
      d[step] = c.y;
      d[2 * step] = c.z;
      d[3 * step] = c.w;
    }
  }
  wave_sync();
  wave_idct_lines_dyn(g.C, tile, lane);
}

template <class Coef4, class LlfAt>
__device__ __forceinline__ void wave_large_pass1(const LargeGeom& g, int v0, Coef4 coef4, LlfAt llf_at,
                                                 float* __restrict__ plane, const PixLayout lay, float* tile, int lane) {
  wave_large_pass1_stage(g, v0, coef4, llf_at, tile, lane);
  wave_tile_store<true>(tile, large_pitch(g.C), plane, lay, 0, v0, g.C, g.LV, lane);
  wave_sync();
}

// Pass 2 of one column slab straight from LDS, for varblocks whose whole channel fits the workgroup's tiles (R * C <=
// kLargeWaves * kLargeSlab: everything below 256 pixels): `tiles` = the first of the R / LV consecutive wave tiles
// that hold the pass-1 result (tile s: lines v in [s * LV, (s + 1) * LV), sample x at large_pos(x), line pitch
// large_pitch(C)).  A lane owns one pixel column (or one length-64 leaf of it, R = 128) exactly as in
// wave_idct_lines; the results go from registers to the output rectangle -- four consecutive rows of a column are one
// 16-byte piece of the tiled plane layout -- so the intermediate never leaves the CU.
template <int N>
__device__ __forceinline__ void wave_large_pass2_lds(const LargeGeom& g, int x0, const float* __restrict__ tiles,
                                                     float* __restrict__ plane, const PixLayout lay, int lane) {
  static_assert(N == 32 || N == 64 || N == 128, "the 256-point columns do not fit a workgroup's LDS");
  constexpr int K = N == 128 ? 2 : 1, NR = N == 32 ? 32 : 64;
  const int P1 = large_pitch(g.C);
  const int leaf = lane & (K - 1), col = lane / K;
  const bool active = col < g.LX;
  const float* t = tiles + large_pos(x0 + min(col, g.LX - 1));
  const int llv = g.llv, lmask = g.LV - 1;
  // natural sample v of the lane's column (v is a constant after unrolling: the offset is wave-uniform)
  auto at = [&](int v) -> float { return t[(v >> llv) * kLargeTile + (v & lmask) * P1]; };
  float x[NR];
  if constexpr (K == 1) {
#pragma unroll
    for (int j = 0; j < NR; j++) x[j] = at(j);
    idct1d<NR, true>(x);
  } else {
    // E[j] = t[2j];  O'[j] = t[2j+1] + t[2j-1], O'[0] = t[1] * sqrt2      (idct_large.rs:284-296)
    const bool odd = leaf != 0;
    const float* tl = t + leaf * P1;  // samples 2j and 2j + 1 share a tile (LV is even)
#pragma unroll
    for (int j = 0; j < 64; j++) {
      const float a = tl[((2 * j) >> llv) * kLargeTile + ((2 * j) & lmask) * P1];
      const float b = at(j ? 2 * j - 1 : 1);
      x[j] = odd ? (j ? a + b : a * kSqrt2) : a;
    }
    idct1d<64, true>(x);
    leaf_butterfly<128>(x, odd, false, dpp_xor1);
  }
  if (!active) return;
  const int base = leaf * 64;
  float* d = plane + lay.xoff(x0 + col);
  if (lay.tiled) {
#pragma unroll
    for (int q = 0; q < NR / 4; q++)
      *reinterpret_cast<float4*>(d + lay.at(0, base + 4 * q)) = make_float4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
  } else {
#pragma unroll
    for (int r = 0; r < NR; r++) d[lay.at(0, base + r)] = x[r];
  }
}

__device__ __forceinline__ void wave_large_pass2_lds_dyn(const LargeGeom& g, int x0, const float* tiles, float* plane,
                                                         const PixLayout lay, int lane) {
  switch (g.R) {
    case 32: wave_large_pass2_lds<32>(g, x0, tiles, plane, lay, lane); break;
    case 64: wave_large_pass2_lds<64>(g, x0, tiles, plane, lay, lane); break;
    default: wave_large_pass2_lds<128>(g, x0, tiles, plane, lay, lane); break;
  }
}

// Pass 2 of ONE slab by ONE wave: pixel columns x0 .. x0 + LX, vertical IDCT in place in the output rectangle.
__device__ __forceinline__ void wave_large_pass2(const LargeGeom& g, int x0, float* __restrict__ plane,
                                                 const PixLayout lay, float* tile, int lane) {
  const int P = large_pitch(g.R);
  wave_tile_load_columns(tile, P, plane, lay, x0, g.LX, g.R, lane);
  wave_sync();
  wave_idct_lines_dyn(g.R, tile, lane);
  wave_tile_store<false>(tile, P, plane, lay, x0, 0, g.LX, g.R, lane);
  wave_sync();
}

// One channel of one large varblock, both passes by one 256-thread workgroup (stage hook; the frame path runs the
// passes as separate launches over slab units, k_vardct.hip).  lf points at the cy x cx LF patch (row pitch
// lf_stride); plane at the top-left output pixel (addressing given by `lay`).  lds: kLargeWaves * kLargeTile + 2048
// floats.  coef(k): the dequantised coefficient at stored index k.  All threads must call with identical arguments.
template <class CoefFn>
__device__ void large_varblock_channel(int type, CoefFn coef, const float* __restrict__ lf, int lf_stride,
                                       float* __restrict__ plane, const PixLayout lay, float* lds, int tid) {
  const LargeGeom g(type);
  const int wave = tid >> 6, lane = tid & 63;
  float* tile = lds + wave * kLargeTile;
  float* llf = lds + kLargeWaves * kLargeTile;  // 1024 floats of result + 1024 of scratch
  if (wave == 0) wave_large_llf(lf, lf_stride, g.cy, g.cx, llf + 1024, llf, lane);
  __syncthreads();
  auto coef4 = [&](int k) { return make_float4(coef(k), coef(k + 1), coef(k + 2), coef(k + 3)); };
  for (int s = wave; s < g.slabs_per_pass(); s += kLargeWaves)
    wave_large_pass1(g, s * g.LV, coef4, [&](int i) { return llf[i]; }, plane, lay, tile, lane);
  __threadfence_block();
  __syncthreads();
  for (int s = wave; s < g.slabs_per_pass(); s += kLargeWaves) wave_large_pass2(g, s * g.LX, plane, lay, tile, lane);
  __threadfence_block();
  __syncthreads();
}

}  // namespace jxlh
