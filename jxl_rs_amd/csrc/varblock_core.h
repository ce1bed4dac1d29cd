// Wave-level varblock reconstruction cores (LLF-from-LF + 2-D IDCT + the 8x8 special
// transforms) operating on coefficient tiles staged in LDS.
//
// Semantics: transform_to_pixels_impl (jxl_transforms/src/transform.rs:377-664).
// Design (MI355X): one wavefront owns one LDS tile and works on a *batch* of NB
// varblocks of one shape at a time; each lane runs whole 1-D transforms in VGPRs
// (dct_device.h) and the two passes exchange data through the tile:
//
//   stage   M[b][u][v]   u = horizontal frequency (pass-1 axis), v contiguous
//   pass 1  lane (b, v): column u=0..C-1 of M  -> IDCT_C -> row v of T   (T[b][v][x])
//   pass 2  lane (b, x): column v=0..R-1 of T  -> IDCT_R -> pixels out[y][x]
//
// i.e. horizontal first, then vertical -- the reference's order for every shape
// (idct2d.rs:111-131, idct_large.rs:387-501).  Column reads are conflict-free
// (consecutive lanes -> consecutive LDS words; block strides padded so the 32-lane
// halves of a ds_read_b32 hit 32 distinct banks), row writes are 16-byte
// ds_write_b128 with odd 16-B-slot row pitch.  No wave needs another wave: only
// wave-scope synchronisation is used, so the 4 waves of a workgroup stream
// independent batches.
#pragma once
#include "dct_device.h"
#include "jxlh_internal.h"

namespace jxlh {

__device__ __forceinline__ void wave_sync() {
  // LDS operations of one wavefront execute in program order; this only has to stop
  // the compiler from moving LDS accesses across the phase boundary.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int cmin(int a, int b) { return a < b ? a : b; }
constexpr int cmax(int a, int b) { return a < b ? b : a; }

// LDS tile geometry for an R x C (pixels) DCT shape, R, C in {8, 16, 32}.
template <int R_, int C_, int NB_ = 64 / (R_ < C_ ? R_ : C_)>
struct Shape {
  static constexpr int R = R_, C = C_;
  static constexpr bool kWide = R < C;
  static constexpr int kMin = cmin(R, C), kMax = cmax(R, C);
  static constexpr int NB = NB_;                // varblocks per batch
  static constexpr int N = R * C;               // coefficients per varblock
  static constexpr int E = NB * N / 64;         // coefficients per lane per channel
  static constexpr int LM = kWide ? R + 1 : R;  // M row pitch (words)
  static constexpr int SM_raw = C * LM;
  static constexpr int SM = SM_raw + ((R % 32) - (SM_raw % 32) + 32) % 32;
  static constexpr int LT = C + 4;              // T row pitch: odd number of 16-B slots
  static constexpr int ST_raw = R * LT;
  static constexpr int ST = ST_raw + ((C % 32) - (ST_raw % 32) + 32) % 32;
  static constexpr int I1 = (NB * R + 63) / 64;  // pass-1 iterations (tasks = NB*R columns)
  static constexpr int I2 = (NB * C + 63) / 64;  // pass-2 iterations (tasks = NB*C columns)
  static constexpr int kTile = NB * cmax(SM, ST);
  static_assert(SM % 4 == 0 || kWide, "float4 staging needs 16-byte aligned blocks");
  static_assert(ST % 4 == 0 && LT % 4 == 0, "float4 row writes");
};

// LDS word address of stored coefficient k of block b in M.
template <class S>
__device__ __forceinline__ int m_addr(int b, int k) {
  if constexpr (S::kWide) {
    const int v = k / S::C, u = k % S::C;  // stored in[v*C + u]
    return b * S::SM + u * S::LM + v;
  } else {
    return b * S::SM + k;                  // stored in[u*R + v], LM == R
  }
}

// LLF-from-LF for a CY x CX LF patch held in registers (row-major cy x cx in; min x max
// row-major out).  reinterpreting_dct2d.rs: wide -> rows then columns (:110-136);
// square/thin -> columns then rows, result transposed (:140-213).  Shapes with
// min(cy,cx) <= 4 run unfused there (ScalarDescriptor / 128-bit, :535-600).
template <int CY, int CX>
__device__ __forceinline__ void llf_from_lf(float (&a)[CY * CX]) {
  constexpr bool kF = cmin(CY, CX) > 4;
  if constexpr (CY == 1 && CX == 1) {
    return;
  } else if constexpr (CY < CX) {
#pragma unroll
    for (int y = 0; y < CY; y++) {
      float t[CX];
#pragma unroll
      for (int x = 0; x < CX; x++) t[x] = a[y * CX + x];
      rdct1d<CX, kF>(t);
#pragma unroll
      for (int x = 0; x < CX; x++) a[y * CX + x] = t[x];
    }
#pragma unroll
    for (int x = 0; x < CX; x++) {
      float t[CY];
#pragma unroll
      for (int y = 0; y < CY; y++) t[y] = a[y * CX + x];
      rdct1d<CY, kF>(t);
#pragma unroll
      for (int y = 0; y < CY; y++) a[y * CX + x] = t[y];
    }
  } else {
    float o[CY * CX];
#pragma unroll
    for (int x = 0; x < CX; x++) {
      float t[CY];
#pragma unroll
      for (int y = 0; y < CY; y++) t[y] = a[y * CX + x];
      rdct1d<CY, kF>(t);
#pragma unroll
      for (int y = 0; y < CY; y++) a[y * CX + x] = t[y];
    }
#pragma unroll
    for (int v = 0; v < CY; v++) {
      float t[CX];
#pragma unroll
      for (int x = 0; x < CX; x++) t[x] = a[v * CX + x];
      rdct1d<CX, kF>(t);
#pragma unroll
      for (int u = 0; u < CX; u++) o[u * CY + v] = t[u];
    }
#pragma unroll
    for (int i = 0; i < CY * CX; i++) a[i] = o[i];
  }
}

// Runs LLF + both IDCT passes on a staged batch.
//   lf_load(b, y, x) -> LF sample of block b at (row y, col x) of its cy x cx patch
//   store8(b, x, yb, v)     pixels (rows 8*yb .. 8*yb+7, col x) of block b
// Lanes of blocks b >= nb compute on whatever is in the tile and are masked at the store.
template <class S, class LfLoad, class Store>
__device__ __forceinline__ void idct_batch(float* __restrict__ buf, int nb, int lane, LfLoad lf_load,
                                           Store store) {
  constexpr int R = S::R, C = S::C;
  constexpr int CY = R / 8, CX = C / 8;
  // ---- LLF: one lane per block (tiny: <= 16 samples)
  if (lane < nb) {
    float a[CY * CX];
#pragma unroll
    for (int y = 0; y < CY; y++)
#pragma unroll
      for (int x = 0; x < CX; x++) a[y * CX + x] = lf_load(lane, y, x);
    llf_from_lf<CY, CX>(a);
    constexpr int MN = cmin(CY, CX), MX = cmax(CY, CX);
#pragma unroll
    for (int r = 0; r < MN; r++)
#pragma unroll
      for (int q = 0; q < MX; q++) buf[m_addr<S>(lane, r * S::kMax + q)] = a[r * MX + q];
  }
  wave_sync();
  // ---- pass 1: IDCT_C along u for every (block, v)
  float p1[S::I1][C];
#pragma unroll
  for (int it = 0; it < S::I1; it++) {
    const int t = it * 64 + lane;
    const int b = min(t / R, S::NB - 1), v = t % R;
    const float* src = buf + b * S::SM + v;
#pragma unroll
    for (int u = 0; u < C; u++) p1[it][u] = src[u * S::LM];
    idct1d<C, true>(p1[it]);
  }
  wave_sync();  // every column is in registers before any row of T lands on top of M
#pragma unroll
  for (int it = 0; it < S::I1; it++) {
    const int t = it * 64 + lane;
    const int b = t / R, v = t % R;
    if (b < S::NB) {
      float4* dst = reinterpret_cast<float4*>(buf + b * S::ST + v * S::LT);
#pragma unroll
      for (int x = 0; x < C; x += 4) dst[x / 4] = make_float4(p1[it][x], p1[it][x + 1], p1[it][x + 2], p1[it][x + 3]);
    }
  }
  wave_sync();
  // ---- pass 2: IDCT_R along v for every (block, x); results go straight out
#pragma unroll
  for (int it = 0; it < S::I2; it++) {
    const int t = it * 64 + lane;
    const int b = t / C, x = t % C;
    float col[R];
    const float* src = buf + min(b, S::NB - 1) * S::ST + x;
#pragma unroll
    for (int v = 0; v < R; v++) col[v] = src[v * S::LT];
    idct1d<R, true>(col);
    if (b < nb) {
#pragma unroll
      for (int yb = 0; yb < R / 8; yb++) {
        const float v8[8] = {col[yb * 8],     col[yb * 8 + 1], col[yb * 8 + 2], col[yb * 8 + 3],
                             col[yb * 8 + 4], col[yb * 8 + 5], col[yb * 8 + 6], col[yb * 8 + 7]};
        store(b, x, yb, v8);
      }
    }
  }
  wave_sync();  // tile may be restaged
}

// ------------------------------------------------------------------------------------------
// 8x8 special transforms: one lane per block, coefficients read from an LDS tile
// in[b*kSpecPitch + k], pixels written to out[b*kSpecPitch + p] (p = y*8+x).
// transform.rs:14-32, :295-374, :510-662.  These 4-wide paths are unfused in the
// reference's x86 build (idct2d.rs:348-366) and plain scalar code otherwise.
constexpr int kSpecPitch = 65;  // odd: lane b walks its own block conflict-free
constexpr int kSpecNB = 32;     // blocks per batch: in + out tiles of one wave = 16.6 KB of LDS

__constant__ float kAfvBasisDev[256] = {
#include "afv_basis.inc"
};

// 2-D IDCT of a small block in registers, reference layout contract (tests.rs:119-132).
template <int R, int C>
__device__ __forceinline__ void idct2d_small(float (&d)[R * C]) {
  constexpr bool kF = cmin(R, C) > 4;
  if constexpr (R < C) {
#pragma unroll
    for (int v = 0; v < R; v++) {
      float t[C];
#pragma unroll
      for (int u = 0; u < C; u++) t[u] = d[v * C + u];
      idct1d<C, kF>(t);
#pragma unroll
      for (int u = 0; u < C; u++) d[v * C + u] = t[u];
    }
#pragma unroll
    for (int x = 0; x < C; x++) {
      float t[R];
#pragma unroll
      for (int v = 0; v < R; v++) t[v] = d[v * C + x];
      idct1d<R, kF>(t);
#pragma unroll
      for (int v = 0; v < R; v++) d[v * C + x] = t[v];
    }
  } else {
    float o[R * C];
#pragma unroll
    for (int v = 0; v < R; v++) {
      float t[C];
#pragma unroll
      for (int u = 0; u < C; u++) t[u] = d[u * R + v];
      idct1d<C, kF>(t);
#pragma unroll
      for (int x = 0; x < C; x++) o[v * C + x] = t[x];  // o[v][x] = tmp[x][v]
    }
#pragma unroll
    for (int x = 0; x < C; x++) {
      float t[R];
#pragma unroll
      for (int v = 0; v < R; v++) t[v] = o[v * C + x];
      idct1d<R, kF>(t);
#pragma unroll
      for (int y = 0; y < R; y++) d[y * C + x] = t[y];
    }
  }
}

template <int S>
__device__ __forceinline__ void idct2_top_block(const float* in, float* out) {
  constexpr int num = S / 2;
#pragma unroll
  for (int y = 0; y < num; y++) {
#pragma unroll
    for (int x = 0; x < num; x++) {
      const float c00 = in[y * 8 + x];
      const float c01 = in[y * 8 + num + x];
      const float c10 = in[(y + num) * 8 + x];
      const float c11 = in[(y + num) * 8 + num + x];
      out[y * 2 * 8 + x * 2] = c00 + c01 + c10 + c11;
      out[y * 2 * 8 + x * 2 + 1] = c00 + c01 - c10 - c11;
      out[(y * 2 + 1) * 8 + x * 2] = c00 - c01 + c10 - c11;
      out[(y * 2 + 1) * 8 + x * 2 + 1] = c00 - c01 - c10 + c11;
    }
  }
}

// TYPE in {1,2,3,12,13,14..17}; c = this block's 64 coefficients (c[0] already = lf),
// o = this block's 64 output pixels.  c may be clobbered.  Every index is a compile-time
// constant after unrolling, so c/o may be register arrays (k1_special) as well as LDS rows.
// afv_basis: where the AFV bodies read the 16 x 16 basis from -- the constant table (scalar loads: 256 values against
// ~100 scalar registers, the compiler spills 350 of them in k1_special) or a copy in LDS (broadcast reads, round 5)
template <int TYPE>
__device__ __forceinline__ void special_8x8_t(float* __restrict__ c, float* __restrict__ o,
                                              const float* __restrict__ afv_basis = kAfvBasisDev) {
  if constexpr (TYPE == 1) {  // IDENTITY (Hornuss)
    const float b00 = c[0], b01 = c[1], b10 = c[8], b11 = c[9];
    const float dcs[4] = {b00 + b01 + b10 + b11, b00 + b01 - b10 - b11, b00 - b01 + b10 - b11,
                          b00 - b01 - b10 + b11};
#pragma unroll
    for (int y = 0; y < 2; y++) {
#pragma unroll
      for (int x = 0; x < 2; x++) {
        float residual_sum = 0.0f;
#pragma unroll
        for (int iy = 0; iy < 4; iy++)
#pragma unroll
          for (int ix = 0; ix < 4; ix++) {
            if (ix == 0 && iy == 0) continue;
            residual_sum += c[(y + iy * 2) * 8 + x + ix * 2];
          }
        const float pivot = dcs[y * 2 + x] - residual_sum * (1.0f / 16.0f);
#pragma unroll
        for (int iy = 0; iy < 4; iy++)
#pragma unroll
          for (int ix = 0; ix < 4; ix++) {
            if (ix == 1 && iy == 1) continue;
            o[(y * 4 + iy) * 8 + x * 4 + ix] = c[(y + iy * 2) * 8 + x + ix * 2] + pivot;
          }
        o[(4 * y + 1) * 8 + 4 * x + 1] = pivot;
        o[y * 4 * 8 + x * 4] = c[(y + 2) * 8 + x + 2] + pivot;
      }
    }
  } else if constexpr (TYPE == 2) {  // DCT2X2: three Hadamard levels, ping-pong between the tiles
    idct2_top_block<2>(c, o);
    // levels read the full previous buffer outside the top block too: copy it over
#pragma unroll
    for (int i = 0; i < 64; i++) {
      const int y = i / 8, x = i % 8;
      if (y >= 2 || x >= 2) o[i] = c[i];
    }
    idct2_top_block<4>(o, c);
#pragma unroll
    for (int i = 0; i < 64; i++) {
      const int y = i / 8, x = i % 8;
      if (y >= 4 || x >= 4) c[i] = o[i];
    }
    idct2_top_block<8>(c, o);
  } else if constexpr (TYPE == 3) {  // DCT4X4
    const float b00 = c[0], b01 = c[1], b10 = c[8], b11 = c[9];
    const float dcs[4] = {b00 + b01 + b10 + b11, b00 + b01 - b10 - b11, b00 - b01 + b10 - b11,
                          b00 - b01 - b10 + b11};
#pragma unroll
    for (int y = 0; y < 2; y++)
#pragma unroll
      for (int x = 0; x < 2; x++) {
        float blk[16];
#pragma unroll
        for (int iy = 0; iy < 4; iy++)
#pragma unroll
          for (int ix = 0; ix < 4; ix++)
            blk[iy * 4 + ix] = (ix == 0 && iy == 0) ? dcs[y * 2 + x] : c[(y + iy * 2) * 8 + x + ix * 2];
        idct2d_small<4, 4>(blk);
#pragma unroll
        for (int iy = 0; iy < 4; iy++)
#pragma unroll
          for (int ix = 0; ix < 4; ix++) o[(y * 4 + iy) * 8 + x * 4 + ix] = blk[iy * 4 + ix];
      }
  } else if constexpr (TYPE == 13) {  // DCT8X4
    const float dcs[2] = {c[0] + c[8], c[0] - c[8]};
#pragma unroll
    for (int x = 0; x < 2; x++) {
      float blk[32];
#pragma unroll
      for (int iy = 0; iy < 4; iy++)
#pragma unroll
        for (int ix = 0; ix < 8; ix++)
          blk[iy * 8 + ix] = (ix == 0 && iy == 0) ? dcs[x] : c[(x + iy * 2) * 8 + ix];
      idct2d_small<8, 4>(blk);
#pragma unroll
      for (int iy = 0; iy < 8; iy++)
#pragma unroll
        for (int ix = 0; ix < 4; ix++) o[iy * 8 + x * 4 + ix] = blk[iy * 4 + ix];
    }
  } else if constexpr (TYPE == 12) {  // DCT4X8
    const float dcs[2] = {c[0] + c[8], c[0] - c[8]};
#pragma unroll
    for (int y = 0; y < 2; y++) {
      float blk[32];
#pragma unroll
      for (int iy = 0; iy < 4; iy++)
#pragma unroll
        for (int ix = 0; ix < 8; ix++)
          blk[iy * 8 + ix] = (ix == 0 && iy == 0) ? dcs[y] : c[(y + iy * 2) * 8 + ix];
      idct2d_small<4, 8>(blk);
#pragma unroll
      for (int iy = 0; iy < 4; iy++)
#pragma unroll
        for (int ix = 0; ix < 8; ix++) o[(y * 4 + iy) * 8 + ix] = blk[iy * 8 + ix];
    }
  } else {  // AFV0..3
    static_assert(TYPE >= 14 && TYPE <= 17, "special transform type");
    constexpr int kind = TYPE - 14;
    constexpr int afv_x = kind & 1, afv_y = kind / 2;
    const float b00 = c[0], b01 = c[1], b10 = c[8];
    const float dcs[3] = {(b00 + b10 + b01) * 4.0f, b00 + b10 - b01, b00 - b10};
    {
      float coeff[16];
#pragma unroll
      for (int iy = 0; iy < 4; iy++)
#pragma unroll
        for (int ix = 0; ix < 4; ix++)
          coeff[iy * 4 + ix] = (ix == 0 && iy == 0) ? dcs[0] : c[iy * 2 * 8 + ix * 2];
#pragma unroll
      for (int i = 0; i < 16; i++) {
        // (keeps the 16 basis values of output i next to their use: hoisted, the 256 of them take every register)
        asm volatile("" ::: "memory");
        float pixel = 0.0f;
#pragma unroll
        for (int j = 0; j < 16; j++) pixel += coeff[j] * afv_basis[j * 16 + i];
        const int iy = i / 4, ix = i % 4;
        const int py = afv_y == 1 ? 3 - iy : iy;
        const int px = afv_x == 1 ? 3 - ix : ix;
        // pixels[(iy' + afv_y*4)*8 + afv_x*4 + ix'] = block[by*4 + bx] with (by,bx) flipped:
        // block index i=(iy,ix) lands at iy' = flip(iy), ix' = flip(ix) (the flip is an involution)
        o[(py + afv_y * 4) * 8 + afv_x * 4 + px] = pixel;
      }
    }
    {
      float blk[16];
#pragma unroll
      for (int iy = 0; iy < 4; iy++)
#pragma unroll
        for (int ix = 0; ix < 4; ix++)
          blk[iy * 4 + ix] = (ix == 0 && iy == 0) ? dcs[1] : c[iy * 2 * 8 + ix * 2 + 1];
      idct2d_small<4, 4>(blk);
#pragma unroll
      for (int iy = 0; iy < 4; iy++)
#pragma unroll
        for (int ix = 0; ix < 4; ix++) o[(iy + afv_y * 4) * 8 + (1 - afv_x) * 4 + ix] = blk[iy * 4 + ix];
    }
    {
      float blk[32];
#pragma unroll
      for (int iy = 0; iy < 4; iy++)
#pragma unroll
        for (int ix = 0; ix < 8; ix++)
          blk[iy * 8 + ix] = (ix == 0 && iy == 0) ? dcs[2] : c[(1 + iy * 2) * 8 + ix];
      idct2d_small<4, 8>(blk);
#pragma unroll
      for (int iy = 0; iy < 4; iy++)
#pragma unroll
        for (int ix = 0; ix < 8; ix++) o[(iy + (1 - afv_y) * 4) * 8 + ix] = blk[iy * 8 + ix];
    }
  }
}

// Runtime-type front end on memory-resident blocks (stage hook, generic callers).
__device__ inline void special_8x8(int type, float* c, float* o) {
  switch (type) {
    case 1: special_8x8_t<1>(c, o); return;
    case 2: special_8x8_t<2>(c, o); return;
    case 3: special_8x8_t<3>(c, o); return;
    case 12: special_8x8_t<12>(c, o); return;
    case 13: special_8x8_t<13>(c, o); return;
    case 14: special_8x8_t<14>(c, o); return;
    case 15: special_8x8_t<15>(c, o); return;
    case 16: special_8x8_t<16>(c, o); return;
    default: special_8x8_t<17>(c, o); return;
  }
}

// The same with the block held in registers between an LDS row read and an LDS row write.
template <int TYPE>
__device__ __forceinline__ void special_8x8_regs(const float* __restrict__ c_row, float* __restrict__ o_row) {
  float c[64], o[64];
#pragma unroll
  for (int i = 0; i < 64; i++) c[i] = c_row[i];
  special_8x8_t<TYPE>(c, o);
#pragma unroll
  for (int i = 0; i < 64; i++) o_row[i] = o[i];
}

}  // namespace jxlh
