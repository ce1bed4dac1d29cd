// K2+K3 fused: Gaborish -> EPF1 -> EPF2 in ONE pass over HBM (12 B/px in, 12 B/px out).
//
// One 512-thread workgroup owns a 56x32 output tile.  The three XYB channels of the tile plus
// a 4-pixel halo (1 Gaborish + 2 EPF1 + 1 EPF2, jxl/src/render/mod.rs:28-36) are staged once
// in LDS with 16-byte coalesced loads; every stage then runs LDS -> LDS on a region that
// shrinks by its own border, and only the last stage writes to HBM.  Each thread produces
// 4-pixel row strips; a staged row is exactly 16 strips = one 16-lane DPP row, so the strip
// itself comes in as one conflict-free ds_read_b128 and its left/right neighbour taps are
// DPP row shifts of the adjacent lanes' registers (no second trip to LDS).  For EPF1 the 16
// absolute differences per pixel and channel collapse to two shared difference maps
//   V(x,y) = |P(x,y) - P(x,y+1)|,  H(x,y) = |P(x,y) - P(x+1,y)|
// (every |a-b| of epf1.rs:100-115 is one of them), summed in the reference's order, so the
// result stays bit-identical to the per-stage kernels / the oracle.
//
// Edge semantics (jxl/src/render/simple_pipeline/run_stage.rs:129-146): each stage sees ITS
// OWN input mirrored at the frame border.  The staged input is loaded with mirrored
// coordinates; after each intermediate stage the out-of-frame part of its output region is
// overwritten with the mirrored in-frame values (they are inside the same tile), so the
// next stage reads exactly what the reference's pipeline would hand it.
//
// Stage subsets (gab on/off, epf_iters 0..2) are compile-time variants of the same kernel;
// epf_iters == 3 (EPF0, 7-pixel halo) uses the per-stage kernels of k_filters.hip.
#include "jxlh_internal.h"

namespace jxlh {
namespace {

constexpr int kTW = 56, kTH = 32, kB = 4;
constexpr int kBW = kTW + 2 * kB;  // 64 floats = 16 strips = one DPP row of lanes
constexpr int kBH = kTH + 2 * kB;  // 40
constexpr int kStrips = kBW / 4;   // 16
constexpr int kPlane = kBW * kBH;  // floats per channel
constexpr int kFusedThreads = 512;
static_assert(kStrips == 16, "lane <-> strip mapping relies on 16-lane DPP rows");

struct FusedArgs {
  const float* in[3];
  float* out[3];
  const float* inv_sigma;
  size_t stride, sigma_stride;
  int w, h;
  int y0, y1;  // output rows [y0, y1) (band sharding); tiles start at y0
  float gab_k[3][3];
  float scale[3];
  float sm1, bsm1, sm2, bsm2;
  int tiled_in, xblocks;  // input layout (see FrameDev::tiled)
};

// float offset of frame pixel (fx, fy) in an input plane
__device__ __forceinline__ size_t in_offset(const FusedArgs& a, int fx, int fy) {
  return a.tiled_in ? ((size_t)((fy >> 3) * a.xblocks + (fx >> 3)) * 64 + (size_t)((fx & 7) * 8 + (fy & 7)))
                    : ((size_t)fy * a.stride + (size_t)fx);
}

// DPP row shifts: lane i takes the value of lane i-1 / i+1 of its 16-lane row (the edge lanes
// of a row get 0; those taps only feed edge strips nobody consumes).
__device__ __forceinline__ float dpp_from_left(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_from_right(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, true));
}

// 8 consecutive values of a tile row around a strip: v[0..1] = cols bx0-2,-1; v[2..5] = strip;
// v[6..7] = cols bx0+4,+5.  Lane l of a 16-lane row holds strip l of one tile row, so the
// neighbours are the adjacent lanes' strip registers.  Must be called by all 64 lanes.
// LDS tiles are 16-byte aligned and every strip starts on a 4-float boundary; say so, or the
// compiler splits the access into ds_read2_b32 pairs (2-way bank conflicts).
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 lds_load4(const float* p) {
  f32x4 v = *reinterpret_cast<const f32x4*>(__builtin_assume_aligned(p, 16));
  // keep the 128-bit load whole: without this the optimiser scalarises it and the backend
  // re-pairs the pieces as ds_read2_b32 {0,3},{1,2} -> 4-way bank conflicts
  asm volatile("" : "+v"(v));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void lds_store4(float* p, float4 v) {
  *reinterpret_cast<float4*>(__builtin_assume_aligned(p, 16)) = v;
}

__device__ __forceinline__ void load8(const float* __restrict__ row, int bx0, float (&v)[8]) {
  const float4 c = lds_load4(row + bx0);
  v[0] = dpp_from_left(c.z);
  v[1] = dpp_from_left(c.w);
  v[2] = c.x; v[3] = c.y; v[4] = c.z; v[5] = c.w;
  v[6] = dpp_from_right(c.x);
  v[7] = dpp_from_right(c.y);
}
__device__ __forceinline__ void load4(const float* __restrict__ row, int bx0, float (&v)[4]) {
  const float4 c = lds_load4(row + bx0);
  v[0] = c.x; v[1] = c.y; v[2] = c.z; v[3] = c.w;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

#define FAD(a, b) __builtin_fabsf((a) - (b))

// ---- Gaborish on one 4-px strip of one channel (gaborish.rs:83-85)
__device__ __forceinline__ float4 gab_strip(const float* __restrict__ src, int by, int bx0, float k0, float k1,
                                            float k2) {
  float t[8], m[8], b[8];
  load8(src + clampi(by - 1, 0, kBH - 1) * kBW, bx0, t);
  load8(src + by * kBW, bx0, m);
  load8(src + clampi(by + 1, 0, kBH - 1) * kBW, bx0, b);
  float o[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int j = i + 2;
    float sum = m[j] * k0;
    sum = __builtin_fmaf(k1, t[j] + m[j - 1] + b[j] + m[j + 1], sum);
    sum = __builtin_fmaf(k2, t[j - 1] + t[j + 1] + b[j - 1] + b[j + 1], sum);
    o[i] = sum;
  }
  return make_float4(o[0], o[1], o[2], o[3]);
}

__device__ __forceinline__ float sad_mul_px(int fx, int fy, float sm, float bsm) {
  const int xm = fx & 7, ym = fy & 7;
  return (xm == 0 || xm == 7 || ym == 0 || ym == 7) ? bsm : sm;
}

// ---- EPF1 on one strip, all three channels (epf1.rs:84-146)
__device__ __forceinline__ void epf1_strip(const float* __restrict__ src, int by, int bx0, int fx0, int fy,
                                           float sigma, const FusedArgs& a, float4 (&out)[3]) {
  const int rm2 = clampi(by - 2, 0, kBH - 1), rm1 = clampi(by - 1, 0, kBH - 1);
  const int rp1 = clampi(by + 1, 0, kBH - 1), rp2 = clampi(by + 2, 0, kBH - 1);
  float sads[4][4];  // [neighbour][pixel]
#pragma unroll
  for (int k = 0; k < 4; k++)
#pragma unroll
    for (int i = 0; i < 4; i++) sads[k][i] = 0.0f;
  float ctr[3][4], pn[3][4], ps[3][4], pw[3][4], pe[3][4];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const float* p = src + c * kPlane;
    float r0[4], r1[8], r2[8], r3[8], r4[4];
    load4(p + rm2 * kBW, bx0, r0);
    load8(p + rm1 * kBW, bx0, r1);
    load8(p + by * kBW, bx0, r2);
    load8(p + rp1 * kBW, bx0, r3);
    load4(p + rp2 * kBW, bx0, r4);
    // vertical difference map V(x, r) = |P(x,r) - P(x,r+1)|, x in -1..4 (index x+1)
    float vm2[4], vm1[6], v0[6], vp1[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      vm2[i] = FAD(r0[i], r1[i + 2]);
      vp1[i] = FAD(r3[i + 2], r4[i]);
    }
#pragma unroll
    for (int x = 0; x < 6; x++) {
      vm1[x] = FAD(r1[x + 1], r2[x + 1]);
      v0[x] = FAD(r2[x + 1], r3[x + 1]);
    }
    // horizontal difference map H(x, r) = |P(x,r) - P(x+1,r)|
    float hm1[5], h0[7], hp1[5];  // x from -1 (hm1, hp1) / -2 (h0)
#pragma unroll
    for (int x = 0; x < 5; x++) {
      hm1[x] = FAD(r1[x + 1], r1[x + 2]);
      hp1[x] = FAD(r3[x + 1], r3[x + 2]);
    }
#pragma unroll
    for (int x = 0; x < 7; x++) h0[x] = FAD(r2[x], r2[x + 1]);
    const float scale = a.scale[c];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      // order of the five terms == epf1.rs:116-119
      const float sN = vm2[i] + vm1[i] + vm1[i + 1] + vm1[i + 2] + v0[i + 1];
      const float sW = hm1[i] + h0[i] + h0[i + 1] + h0[i + 2] + hp1[i];
      const float sE = hm1[i + 1] + h0[i + 1] + h0[i + 2] + h0[i + 3] + hp1[i + 1];
      const float sS = vm1[i + 1] + v0[i] + v0[i + 1] + v0[i + 2] + vp1[i];
      sads[0][i] = __builtin_fmaf(sN, scale, sads[0][i]);
      sads[1][i] = __builtin_fmaf(sW, scale, sads[1][i]);
      sads[2][i] = __builtin_fmaf(sE, scale, sads[2][i]);
      sads[3][i] = __builtin_fmaf(sS, scale, sads[3][i]);
      ctr[c][i] = r2[i + 2];
      pn[c][i] = r1[i + 2];
      ps[c][i] = r3[i + 2];
      pw[c][i] = r2[i + 1];
      pe[c][i] = r2[i + 3];
    }
  }
  float o[3][4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    if (sigma < kMinSigma) {
#pragma unroll
      for (int c = 0; c < 3; c++) o[c][i] = ctr[c][i];
      continue;
    }
    const float inv_sigma = sigma * sad_mul_px(fx0 + i, fy, a.sm1, a.bsm1);
    float wsum = 1.0f, wgt[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      wgt[k] = fmaxf(__builtin_fmaf(sads[k][i], inv_sigma, 1.0f), 0.0f);
      wsum += wgt[k];
    }
    const float inv_w = 1.0f / wsum;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      float acc = ctr[c][i];
      acc = __builtin_fmaf(ps[c][i], wgt[3], acc);
      acc = __builtin_fmaf(pe[c][i], wgt[2], acc);
      acc = __builtin_fmaf(pw[c][i], wgt[1], acc);
      acc = __builtin_fmaf(pn[c][i], wgt[0], acc);
      o[c][i] = acc * inv_w;
    }
  }
#pragma unroll
  for (int c = 0; c < 3; c++) out[c] = make_float4(o[c][0], o[c][1], o[c][2], o[c][3]);
}

// ---- EPF2 on one strip (epf2.rs:84-136)
__device__ __forceinline__ void epf2_strip(const float* __restrict__ src, int by, int bx0, int fx0, int fy,
                                           float sigma, const FusedArgs& a, float4 (&out)[3]) {
  const int rm1 = clampi(by - 1, 0, kBH - 1), rp1 = clampi(by + 1, 0, kBH - 1);
  float t[3][4], m[3][8], b[3][4];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const float* p = src + c * kPlane;
    load4(p + rm1 * kBW, bx0, t[c]);
    load8(p + by * kBW, bx0, m[c]);
    load4(p + rp1 * kBW, bx0, b[c]);
  }
  float o[3][4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const float xc = m[0][i + 2], yc = m[1][i + 2], bc = m[2][i + 2];
    if (sigma < kMinSigma) {
      o[0][i] = xc;
      o[1][i] = yc;
      o[2][i] = bc;
      continue;
    }
    const float inv_sigma = sigma * sad_mul_px(fx0 + i, fy, a.sm2, a.bsm2);
    float wacc = 1.0f, xa = xc, ya = yc, ba = bc;
    // neighbour order N, W, E, S (epf2.rs:95)
    const float nx[4] = {t[0][i], m[0][i + 1], m[0][i + 3], b[0][i]};
    const float ny[4] = {t[1][i], m[1][i + 1], m[1][i + 3], b[1][i]};
    const float nb[4] = {t[2][i], m[2][i + 1], m[2][i + 3], b[2][i]};
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const float sad = __builtin_fmaf(FAD(nx[k], xc), a.scale[0],
                                       __builtin_fmaf(FAD(ny[k], yc), a.scale[1], FAD(nb[k], bc) * a.scale[2]));
      const float wgt = fmaxf(__builtin_fmaf(sad, inv_sigma, 1.0f), 0.0f);
      wacc += wgt;
      xa = __builtin_fmaf(wgt, nx[k], xa);
      ya = __builtin_fmaf(wgt, ny[k], ya);
      ba = __builtin_fmaf(wgt, nb[k], ba);
    }
    const float inv_w = 1.0f / wacc;
    o[0][i] = xa * inv_w;
    o[1][i] = ya * inv_w;
    o[2][i] = ba * inv_w;
  }
#pragma unroll
  for (int c = 0; c < 3; c++) out[c] = make_float4(o[c][0], o[c][1], o[c][2], o[c][3]);
}

// Overwrites out-of-frame positions of a stage's output region [B-m, B+T+m) with the values
// at their mirrored in-frame coordinates.  Only called by tiles that touch the frame border.
__device__ __forceinline__ void mirror_fill(float* __restrict__ buf, int m, int tx0, int ty0, int w, int h, int tid) {
  const int rw = kTW + 2 * m, rh = kTH + 2 * m;
  for (int idx = tid; idx < rw * rh; idx += kFusedThreads) {
    const int bx = kB - m + idx % rw, by = kB - m + idx / rw;
    const int fx = tx0 - kB + bx, fy = ty0 - kB + by;
    if (fx >= 0 && fx < w && fy >= 0 && fy < h) continue;
    const int sx = mirror(fx, w) - (tx0 - kB), sy = mirror(fy, h) - (ty0 - kB);
    if (sx < 0 || sx >= kBW || sy < 0 || sy >= kBH) continue;  // outside this tile: never consumed
#pragma unroll
    for (int c = 0; c < 3; c++) buf[c * kPlane + by * kBW + bx] = buf[c * kPlane + sy * kBW + sx];
  }
}

template <bool GAB, bool E1, bool E2>
__global__ __launch_bounds__(kFusedThreads, 4) void k23_fused_filters(const FusedArgs a) {
  __shared__ __attribute__((aligned(16))) float s_a[3 * kPlane];
  __shared__ __attribute__((aligned(16))) float s_b[3 * kPlane];
  // 1/sigma of the 8x8 blocks this tile touches (block columns/rows relative to the tile's first block)
  constexpr int kSigW = kBW / 8 + 2, kSigH = kBH / 8 + 2;
  __shared__ float s_sigma[kSigH * kSigW];
  const int tid = threadIdx.x;
  // blockIdx.x enumerates tiles so that the workgroups one XCD receives (ids congruent mod 8)
  // walk along a tile row: neighbouring tiles share 128-byte output lines and halo input
  // lines, which then meet in the same (non-coherent) L2.
  const int tiles_x = (a.w + kTW - 1) / kTW;
  const int tiles_y = (a.y1 - a.y0 + kTH - 1) / kTH;
  int tile_x, tile_y;
  {
    const int b = blockIdx.x, k = b & 7, j = b >> 3;
    tile_y = (j / tiles_x) * 8 + k;
    tile_x = j % tiles_x;
  }
  if (tile_y >= tiles_y) return;
  const int tx0 = tile_x * kTW, ty0 = a.y0 + tile_y * kTH;
  const bool edge = tx0 - kB < 0 || ty0 - kB < 0 || tx0 + kTW + kB > a.w || ty0 + kTH + kB > a.h;
  constexpr int kBorder = (GAB ? 1 : 0) + (E1 ? 2 : 0) + (E2 ? 1 : 0);
  static_assert(kBorder >= 1 && kBorder <= kB, "at least one stage");

  const int sbx0 = max(tx0 - kB, 0) >> 3, sby0 = max(ty0 - kB, 0) >> 3;
  if constexpr (E1 || E2) {
    if (tid < kSigH * kSigW) {
      const int sx = min(sbx0 + tid % kSigW, (a.w - 1) >> 3), sy = min(sby0 + tid / kSigW, (a.h - 1) >> 3);
      s_sigma[tid] = a.inv_sigma[(size_t)sy * a.sigma_stride + sx];
    }
  }
  // ---- stage the input tile (region margin = kBorder) with mirrored coordinates
  {
    constexpr int m = kBorder;
    constexpr int rows = kTH + 2 * m;
    if (!edge && a.tiled_in) {
      // interior tile, 8x8-tiled column-major input: a lane fetches 4 rows of one pixel column
      // (16 contiguous bytes); a wave covers 32 columns x 8 rows = four whole 256-byte blocks
      // per channel, and scatters into the raster LDS tile with conflict-free ds_write_b32.
      constexpr int yg0 = (kB - m) / 4, ygn = (rows + 2 * ((kB - m) % 4) + 3) / 4;  // 4-row groups touched
      for (int idx = tid; idx < kBW * ygn; idx += kFusedThreads) {
        // idx -> (pair of row groups, half of the columns): lanes 0-31 / 32-63 = the two 4-row
        // halves of the same 32 columns
        const int w64 = idx >> 6, l = idx & 63;
        const int xh = w64 % 2, ypair = w64 / 2;
        const int bx = xh * 32 + (l & 31), yg = yg0 + ypair * 2 + (l >> 5);
        if (yg * 4 >= kBH) continue;
        const int by = yg * 4;
        const size_t off = in_offset(a, tx0 - kB + bx, ty0 - kB + by);
#pragma unroll
        for (int c = 0; c < 3; c++) {
          const float4 v = *reinterpret_cast<const float4*>(a.in[c] + off);
          float* d = s_a + c * kPlane + by * kBW + bx;
          d[0] = v.x;
          d[kBW] = v.y;
          d[2 * kBW] = v.z;
          d[3 * kBW] = v.w;
        }
      }
    } else if (!edge) {  // interior tile, raster input: pure 16-byte coalesced rows
      for (int idx = tid; idx < kStrips * rows; idx += kFusedThreads) {
        const int by = kB - m + idx / kStrips, bx0 = (idx % kStrips) * 4;
        const size_t off = in_offset(a, tx0 - kB + bx0, ty0 - kB + by);
#pragma unroll
        for (int c = 0; c < 3; c++)
          lds_store4(s_a + c * kPlane + by * kBW + bx0, *reinterpret_cast<const float4*>(a.in[c] + off));
      }
    } else {
      for (int idx = tid; idx < kStrips * rows; idx += kFusedThreads) {
        const int by = kB - m + idx / kStrips, bx0 = (idx % kStrips) * 4;
        const int fy = mirror(ty0 - kB + by, a.h);
        const int fx0 = tx0 - kB + bx0;
        const int x0 = mirror(fx0, a.w), x1 = mirror(fx0 + 1, a.w), x2 = mirror(fx0 + 2, a.w),
                  x3 = mirror(fx0 + 3, a.w);
        const size_t o0 = in_offset(a, x0, fy), o1 = in_offset(a, x1, fy), o2 = in_offset(a, x2, fy),
                     o3 = in_offset(a, x3, fy);
#pragma unroll
        for (int c = 0; c < 3; c++) {
          const float* __restrict__ pl = a.in[c];
          lds_store4(s_a + c * kPlane + by * kBW + bx0, make_float4(pl[o0], pl[o1], pl[o2], pl[o3]));
        }
      }
    }
  }
  __syncthreads();
  float* src = s_a;
  float* dst = s_b;
  int margin = kBorder;

  auto run_stage = [&](auto stage_tag, int border) {
    constexpr int STAGE = decltype(stage_tag)::value;  // 0 gaborish, 1 epf1, 2 epf2
    margin -= border;
    const bool last = margin == 0;
    const int rows = kTH + 2 * margin;
    // every row is 16 strips; whole waves run (DPP needs all lanes), stores are guarded
    for (int t0 = 0; t0 < rows * kStrips; t0 += kFusedThreads) {
      if (t0 + (tid & ~63) >= rows * kStrips) break;  // wave-uniform: nothing left for this wave
      const int t = t0 + tid;
      const bool live = t < rows * kStrips;
      const int by = kB - margin + (live ? t / kStrips : 0), bx0 = (t % kStrips) * 4;
      const int fy = ty0 - kB + by, fx0 = tx0 - kB + bx0;
      float4 o[3];
      if constexpr (STAGE == 0) {
#pragma unroll
        for (int c = 0; c < 3; c++)
          o[c] = gab_strip(src + c * kPlane, by, bx0, a.gab_k[c][0], a.gab_k[c][1], a.gab_k[c][2]);
      } else {
        const int sy = (clampi(fy, 0, a.h - 1) >> 3) - sby0, sx = (clampi(fx0, 0, a.w - 1) >> 3) - sbx0;
        const float sigma = s_sigma[sy * kSigW + sx];
        if (__all(sigma < kMinSigma)) {
          // every strip of this wave is below MIN_SIGMA: the stage is the identity here (the
          // reference takes the same shortcut per SIMD vector, epf1.rs:72-78)
#pragma unroll
          for (int c = 0; c < 3; c++) o[c] = lds_load4(src + c * kPlane + by * kBW + bx0);
        } else if constexpr (STAGE == 1) {
          epf1_strip(src, by, bx0, fx0, fy, sigma, a, o);
        } else {
          epf2_strip(src, by, bx0, fx0, fy, sigma, a, o);
        }
      }
      if (!live) continue;
      if (last) {
        if (bx0 >= kB && bx0 < kB + kTW && fy < a.y1 && fy < a.h && fx0 < a.w) {
#pragma unroll
          for (int c = 0; c < 3; c++) *reinterpret_cast<float4*>(a.out[c] + (size_t)fy * a.stride + fx0) = o[c];
        }
      } else {
#pragma unroll
        for (int c = 0; c < 3; c++) lds_store4(dst + c * kPlane + by * kBW + bx0, o[c]);
      }
    }
    if (!last) {
      __syncthreads();
      if (edge) {
        mirror_fill(dst, margin, tx0, ty0, a.w, a.h, tid);
        __syncthreads();
      }
      float* t = src;
      src = dst;
      dst = t;
    }
  };
  if constexpr (GAB) run_stage(std::integral_constant<int, 0>{}, 1);
  if constexpr (E1) run_stage(std::integral_constant<int, 1>{}, 2);
  if constexpr (E2) run_stage(std::integral_constant<int, 2>{}, 1);
}

template <bool GAB, bool E1, bool E2>
void launch_variant(hipStream_t s, const FusedArgs& a) {
  const int tiles_x = (a.w + kTW - 1) / kTW, tiles_y = (a.y1 - a.y0 + kTH - 1) / kTH;
  const dim3 grid(tiles_x * ((tiles_y + 7) / 8) * 8);
  hipLaunchKernelGGL((k23_fused_filters<GAB, E1, E2>), grid, dim3(kFusedThreads), 0, s, a);
}

}  // namespace

// Runs the frame's stage list (gab?, epf1?, epf2?) fused; planes -> tmp.  Returns false if the
// combination is not covered (epf_iters == 3 or nothing to do).
bool launch_fused_filters(hipStream_t s, const FrameDev& f, int y0, int y1) {
  if (f.epf_iters >= 3) return false;
  const bool gab = f.gab != 0, e1 = f.epf_iters >= 1, e2 = f.epf_iters >= 2;
  if (!gab && !e1 && !e2) return false;
  if (y1 <= y0) return true;
  FusedArgs a;
  for (int c = 0; c < 3; c++) {
    a.in[c] = f.planes[c];
    a.out[c] = f.tmp[c];
    a.scale[c] = f.epf_channel_scale[c];
    for (int k = 0; k < 3; k++) a.gab_k[c][k] = f.gab_k[c][k];
  }
  a.inv_sigma = f.inv_sigma;
  a.stride = f.plane_stride;
  a.sigma_stride = (size_t)f.xblocks;
  a.w = f.xsize;
  a.h = f.ysize;
  a.y0 = y0;
  a.y1 = y1;
  a.sm1 = f.epf_sm[1];
  a.bsm1 = f.epf_bsm[1];
  a.sm2 = f.epf_sm[2];
  a.bsm2 = f.epf_bsm[2];
  a.tiled_in = f.tiled;
  a.xblocks = f.xblocks;
  if (gab && e1 && e2) launch_variant<true, true, true>(s, a);
  else if (gab && e1) launch_variant<true, true, false>(s, a);
  else if (gab) launch_variant<true, false, false>(s, a);
  else if (e1 && e2) launch_variant<false, true, true>(s, a);
  else launch_variant<false, true, false>(s, a);
  return true;
}

}  // namespace jxlh
