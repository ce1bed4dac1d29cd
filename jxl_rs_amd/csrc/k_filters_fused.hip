// K2+K3 fused: Gaborish -> EPF1 -> EPF2 in ONE pass over HBM (12 B/px in, 12 B/px out).
//
// One workgroup owns a 56 x kTH output tile.  The three XYB channels of the tile plus a 4-pixel
// halo (1 Gaborish + 2 EPF1 + 1 EPF2, jxl/src/render/mod.rs:28-36) are staged once in LDS with
// 16-byte coalesced loads; every stage then runs LDS -> LDS *in place* on a region that shrinks by
// its own border (all threads compute their outputs into registers, barrier, write back), and
// only the last stage writes to HBM.  One buffer instead of a ping-pong pair halves the LDS per
// pixel, which pays for tall tiles (less halo work) at the same number of waves per CU.
//
// Work items are 4x2 micro-tiles: one 4-pixel strip of two consecutive rows.  A staged row is 16
// strips = one 16-lane DPP row, so a strip comes in as one conflict-free ds_read_b128 and its
// left/right neighbour taps are DPP row shifts of the adjacent lanes' registers.  Two rows per
// item let the EPF stages share work the reference repeats per pixel while keeping its order of
// operations, hence its bits:
//   EPF1 (epf1.rs:100-119): every |a-b| is an entry of one of two difference maps
//       V(x,y) = |P(x,y) - P(x,y+1)|,   H(x,y) = |P(x,y) - P(x+1,y)|
//     and the four 5-term sums are plus-shaped sums over those maps taken in the order
//     (top, left, centre, right, bottom):  PV(x,y) = V(x,y-1)+V(x-1,y)+V(x,y)+V(x+1,y)+V(x,y+1).
//     SAD_N(x,y) = sum_c scale_c*PV_c(x,y-1), SAD_S(x,y) = sum_c scale_c*PV_c(x,y) = SAD_N(x,y+1);
//     SAD_E(x,y) = sum_c scale_c*PH_c(x,y),   SAD_W(x,y) = SAD_E(x-1,y)  -- bit for bit, because
//     the reference adds the same five numbers in the same order for both.
//   EPF2 (epf2.rs:95-109): the single-pixel SAD between two pixels is symmetric, so the S term
//     of (x,y) is the N term of (x,y+1) and the E term of (x,y) the W term of (x+1,y).
// 1/(1+sum w) uses rcp + two FMA refinement steps that round exactly like IEEE division on the
// weights' range [1,16) (checked exhaustively on the device by jxlh_selftest_recip).
//
// Edge semantics (jxl/src/render/simple_pipeline/run_stage.rs:129-146): each stage sees ITS OWN
// input mirrored at the frame border.  The staged input is loaded with mirrored coordinates; after
// each intermediate stage the out-of-frame part of its output region is overwritten with the
// mirrored in-frame values (they are inside the same tile), so the next stage reads exactly what
// the reference's pipeline would hand it.
//
// Stage subsets (gab on/off, epf_iters 0..2) are compile-time variants of the same kernel;
// epf_iters == 3 runs as two launches: Gaborish + EPF0 (EPF0 always closes its kernel: its 7x10
// register window leaves no room to hold outputs across an in-place barrier), then EPF1 + EPF2.
#include "jxlh_internal.h"

#ifndef JXLH_FUSED_TH
#define JXLH_FUSED_TH 56
#endif
#ifndef JXLH_FUSED_THREADS
#define JXLH_FUSED_THREADS 512
#endif
#ifndef JXLH_FUSED_WAVES_PER_EU
#define JXLH_FUSED_WAVES_PER_EU 6
#endif
#ifndef JXLH_FUSED_E0_WPE
#define JXLH_FUSED_E0_WPE 3
#endif
#ifndef JXLH_FUSED_E0_THREADS
#define JXLH_FUSED_E0_THREADS 256
#endif
#ifndef JXLH_FAST_RECIP
#define JXLH_FAST_RECIP 1
#endif
#ifndef JXLH_E1_ROLLED
#define JXLH_E1_ROLLED 1
#endif
#ifndef JXLH_E2_STRIPS
#define JXLH_E2_STRIPS 1
#endif
// An EPF wavefront (128 strip rows) with at most this many active strips hands them to the compacted pass
// (<= 64: a wavefront that compacts takes 64 strips back; 0 = only all-identity wavefronts skip work, the round-1
// behaviour).  EPF2 as the last stage compacts wavefronts with fewer than JXLH_DENSE_ITEMS active strips of 64.
#ifndef JXLH_SPARSE_MAX
#define JXLH_SPARSE_MAX 64
#endif
#ifndef JXLH_DENSE_ITEMS
#define JXLH_DENSE_ITEMS 40
#endif
#ifndef JXLH_DENSE_ITEMS0  // the same threshold for EPF0 as the last stage of its kernel
#define JXLH_DENSE_ITEMS0 40
#endif

#ifndef JXLH_NT_FLOAD
#define JXLH_NT_FLOAD false
#endif
#ifndef JXLH_NT_FSTORE
#define JXLH_NT_FSTORE false
#endif
namespace jxlh {
namespace {

constexpr int kTW = 56, kTH = JXLH_FUSED_TH, kB = 4;
constexpr int kBW = kTW + 2 * kB;  // 64 floats = 16 strips = one DPP row of lanes
constexpr int kBH = kTH + 2 * kB;
constexpr int kStrips = kBW / 4;   // 16
constexpr int kPlane = kBW * kBH;  // floats per channel
constexpr int kThreadsDefault = JXLH_FUSED_THREADS;
static_assert(kStrips == 16, "lane <-> strip mapping relies on 16-lane DPP rows");
static_assert(kTH % 4 == 0 && kThreadsDefault % 64 == 0, "tiled staging fetches 4-row groups");
constexpr int kSigW = kBW / 8 + 2, kSigH = kBH / 8 + 2;
constexpr int kDenseItems = JXLH_DENSE_ITEMS, kDenseItems0 = JXLH_DENSE_ITEMS0, kSparseMax = JXLH_SPARSE_MAX;
static_assert(kSparseMax <= 64, "a compacting wavefront takes 64 strips");

struct FusedArgs {
  const float* in[3];
  float* out[3];
  const float* inv_sigma;
  uint32_t stride, sigma_stride;  // floats; a plane is < 4 GiB (jxlh_frame_begin bounds the frame)
  int w, h;
  int y0, y1;  // output rows [y0, y1) (band sharding); tiles start at y0
  float gab_k[3][3];
  float scale[3];
  float sm0, bsm0, sm1, bsm1, sm2, bsm2;
  int tiled_in, xblocks;  // input layout (see FrameDev::tiled)
};

// BYTE offset of frame pixel (fx, fy) in an input plane.  32-bit on purpose: with a uniform base
// pointer the access becomes "SGPR base + 32-bit VGPR offset" instead of a 64-bit address pair
// per lane and channel (the kernel is register-bound at 80 VGPRs).
__device__ __forceinline__ uint32_t in_offset(const FusedArgs& a, int fx, int fy) {
  return 4u * (a.tiled_in ? ((uint32_t)((fy >> 3) * a.xblocks + (fx >> 3)) * 64u +
                             (uint32_t)((fy & 4) * 8 + (fx & 7) * 4 + (fy & 3)))
                          : ((uint32_t)fy * a.stride + (uint32_t)fx));
}
template <class T>
__device__ __forceinline__ const T& at_bytes(const float* base, uint32_t byte_off) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}
template <class T>
__device__ __forceinline__ T& at_bytes(float* base, uint32_t byte_off) {
  return *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off);
}

// DPP row shifts: lane i takes the value of lane i-1 / i+1 of its 16-lane row (the edge lanes
// of a row get 0; those taps only feed edge strips nobody consumes).
__device__ __forceinline__ float dpp_from_left(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_from_right(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, true));
}

#include "filters_core.inc"

// Overwrites out-of-frame positions of a stage's output region [B-m, B+T+m) with the values
// at their mirrored in-frame coordinates.  Only called by tiles that touch the frame border.
template <int kT>
__device__ __forceinline__ void mirror_fill(float* __restrict__ buf, int m, int tx0, int ty0, int w, int h, int tid) {
  const int rw = kTW + 2 * m, rh = kTH + 2 * m;
  for (int idx = tid; idx < rw * rh; idx += kT) {
    const int bx = kB - m + idx % rw, by = kB - m + idx / rw;
    const int fx = tx0 - kB + bx, fy = ty0 - kB + by;
    if (fx >= 0 && fx < w && fy >= 0 && fy < h) continue;
    const int sx = mirror(fx, w) - (tx0 - kB), sy = mirror(fy, h) - (ty0 - kB);
    if (sx < 0 || sx >= kBW || sy < 0 || sy >= kBH) continue;  // outside this tile: never consumed
#pragma unroll
    for (int c = 0; c < 3; c++) buf[c * kPlane + by * kBW + bx] = buf[c * kPlane + sy * kBW + sx];
  }
}

// Threads per workgroup: 512 (three workgroups of 80 VGPRs per CU), except the EPF0 variants: their 147 VGPRs allow three
// wavefronts per SIMD, i.e. ONE 512-thread workgroup per CU with every barrier and the staging loads exposed; as
// 256-thread workgroups three fit (LDS 51 KB each) and overlap each other: 16K epf_iters = 3, 2.97 -> see profiles/r03_g.
template <bool E0>
constexpr int fused_threads() { return E0 ? JXLH_FUSED_E0_THREADS : kThreadsDefault; }

template <bool GAB, bool E0, bool E1, bool E2>
__global__ __launch_bounds__(fused_threads<E0>(), E0 ? JXLH_FUSED_E0_WPE : JXLH_FUSED_WAVES_PER_EU) void k23_fused_filters(const FusedArgs a) {
  constexpr int kT = fused_threads<E0>();
  __shared__ __attribute__((aligned(16))) float s_buf[3 * kPlane];
  // 1/sigma of the 8x8 blocks this tile touches (block columns/rows relative to the tile's first block)
  __shared__ float s_sigma[kSigH * kSigW];
  // compacted EPF items of the current stage (see run_stage) and the number of wavefronts free to take them
  __shared__ uint16_t s_list[(kTH + 2 * kB) * kStrips];
  __shared__ int s_cnt, s_nsw;
  const int tid_kernel = threadIdx.x, tid = tid_kernel;
  if (tid_kernel == 0) {
    s_cnt = 0;
    s_nsw = 0;
  }
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
  constexpr int kBorder = (GAB ? 1 : 0) + (E0 ? 3 : 0) + (E1 ? 2 : 0) + (E2 ? 1 : 0);
  static_assert(kBorder >= 1 && kBorder <= kB, "at least one stage");
  static_assert(!(E0 && (E1 || E2)), "EPF0 closes its kernel; EPF1/EPF2 follow in a second launch");

  const int sbx0 = max(tx0 - kB, 0) >> 3, sby0 = max(ty0 - kB, 0) >> 3;
  if constexpr (E0 || E1 || E2) {
    if (tid < kSigH * kSigW) {
      const int sx = min(sbx0 + tid % kSigW, (a.w - 1) >> 3), sy = min(sby0 + tid / kSigW, (a.h - 1) >> 3);
      s_sigma[tid] = at_bytes<float>(a.inv_sigma, 4u * ((uint32_t)sy * a.sigma_stride + (uint32_t)sx));
    }
  }
  // ---- stage the input tile (region margin = kBorder) with mirrored coordinates
  {
    constexpr int m = kBorder;
    constexpr int rows = kTH + 2 * m;
    if (!edge && a.tiled_in) {
      // interior tile, 8x8-tiled input (FrameDev::tiled): a lane fetches 4 rows of one pixel column
      // (16 contiguous bytes); the 32 lanes of a half-wave read the same 4-row half of four blocks
      // (4 x 128 contiguous bytes), the two halves together four whole 256-byte blocks per channel;
      // the values scatter into the raster LDS tile with conflict-free ds_write_b32.
      constexpr int yg0 = (kB - m) / 4, ygn = (rows + 2 * ((kB - m) % 4) + 3) / 4;  // 4-row groups touched
      constexpr int items = kBW * ((ygn + 1) / 2) * 2;
#pragma unroll
      for (int it = 0; it < (items + kT - 1) / kT; it++) {
        // idx -> (pair of row groups, half of the columns): lanes 0-31 / 32-63 = the two 4-row
        // halves of the same 32 columns
        const int idx = it * kT + tid;
        const int w64 = idx >> 6, l = idx & 63;
        const int xh = w64 % 2, ypair = w64 / 2;
        const int bx = xh * 32 + (l & 31), yg = yg0 + ypair * 2 + (l >> 5);
        if (idx >= items || yg * 4 >= kBH) continue;
        const int by = yg * 4;
        const uint32_t off = in_offset(a, tx0 - kB + bx, ty0 - kB + by);
#pragma unroll
        for (int c = 0; c < 3; c++) {
          const float4 v = gload_f4<JXLH_NT_FLOAD>(&at_bytes<float>(a.in[c], off));
          float* d = s_buf + c * kPlane + by * kBW + bx;
          d[0] = v.x;
          d[kBW] = v.y;
          d[2 * kBW] = v.z;
          d[3 * kBW] = v.w;
        }
      }
    } else if (!edge) {  // interior tile, raster input: pure 16-byte coalesced rows
      for (int idx = tid; idx < kStrips * rows; idx += kT) {
        const int by = kB - m + idx / kStrips, bx0 = (idx % kStrips) * 4;
        const uint32_t off = in_offset(a, tx0 - kB + bx0, ty0 - kB + by);
#pragma unroll
        for (int c = 0; c < 3; c++)
          lds_store4(s_buf + c * kPlane + by * kBW + bx0, gload_f4<JXLH_NT_FLOAD>(&at_bytes<float>(a.in[c], off)));
      }
    } else {
      for (int idx = tid; idx < kStrips * rows; idx += kT) {
        const int by = kB - m + idx / kStrips, bx0 = (idx % kStrips) * 4;
        const int fy = mirror(ty0 - kB + by, a.h);
        const int fx0 = tx0 - kB + bx0;
        const int x0 = mirror(fx0, a.w), x1 = mirror(fx0 + 1, a.w), x2 = mirror(fx0 + 2, a.w),
                  x3 = mirror(fx0 + 3, a.w);
        const uint32_t o0 = in_offset(a, x0, fy), o1 = in_offset(a, x1, fy), o2 = in_offset(a, x2, fy),
                       o3 = in_offset(a, x3, fy);
#pragma unroll
        for (int c = 0; c < 3; c++) {
          const float* __restrict__ pl = a.in[c];
          lds_store4(s_buf + c * kPlane + by * kBW + bx0, make_float4(at_bytes<float>(pl, o0), at_bytes<float>(pl, o1), at_bytes<float>(pl, o2), at_bytes<float>(pl, o3)));
        }
      }
    }
  }
  __syncthreads();

  // One stage, in place: margin = the output region's margin around the tile (the input region's
  // minus the stage's border).  Every 4x2 item of the region is computed into registers, then
  // (after a barrier: all reads done) written back over the input.
  //
  // EPF stages and sparse sigma maps.  A block whose sigma is below MIN_SIGMA passes through (epf1.rs:72-78), and
  // on d1-like content most blocks do (93 % of the SURVEY population) -- yet a wavefront spans 8 blocks, so nearly
  // half of the wavefronts would run the whole stage for a handful of live lanes.  A wavefront therefore counts
  // its active items: none -> nothing to do (the stage is the identity in place); many (>= kDenseItems) -> the
  // dense form (one strip per lane, side taps through DPP); few -> the active items go to a workgroup-wide list
  // and are processed after a barrier as compacted wavefronts by the GENERIC form of the stage (side taps from
  // LDS).  Both forms add the same terms in the same order.
  auto run_stage = [&](auto stage_tag, auto margin_tag) {
    constexpr int STAGE = decltype(stage_tag)::value;  // 0 gaborish, 1 epf1, 2 epf2
    constexpr int margin = decltype(margin_tag)::value;
    constexpr bool last = margin == 0;
    constexpr int rows = kTH + 2 * margin;
    constexpr int n = (rows / 2) * kStrips;
    constexpr int kPasses = (n + kT - 1) / kT;
    static_assert(rows % 2 == 0, "4x2 items");
    // Opaque copy of the thread id: everything a stage derives from it (item coordinates, sigma
    // indices, store offsets) is then computed inside the stage instead of being hoisted above
    // the previous stage, where it would sit in registers the 80-VGPR budget does not have.
    int tid = tid_kernel;
    asm volatile("" : "+v"(tid));
    static_assert(STAGE != 3 || last, "EPF0 runs as the last stage");
    if constexpr ((STAGE == 2 && last && JXLH_E2_STRIPS) || STAGE == 3) {
      constexpr int ns = rows * kStrips;
      auto strip_geom = [&](int t, int& by, int& bx0, int& fy, int& fx0, float& sigma) {
        by = kB + t / kStrips;
        bx0 = (t % kStrips) * 4;
        fy = ty0 - kB + by;
        fx0 = tx0 - kB + bx0;
        const int sx = (min(max(fx0, 0), a.w - 1) >> 3) - sbx0, sy = (min(max(fy, 0), a.h - 1) >> 3) - sby0;
        sigma = s_sigma[sy * kSigW + sx];
      };
      auto store = [&](int bx0, int fy, int fx0, int c, float4 o) {
        if (bx0 >= kB && bx0 < kB + kTW && fy < a.y1 && fy < a.h && fx0 < a.w)
          gstore_f4<JXLH_NT_FSTORE>(&at_bytes<float>(a.out[c], 4u * ((uint32_t)fy * a.stride + (uint32_t)fx0)), o);
      };
#pragma unroll 1
      for (int t0 = 0; t0 < ns; t0 += kT) {
        if (t0 + (tid & ~63) >= ns) break;
        const int t = t0 + tid;
        const bool live = t < ns;
        int by, bx0, fy, fx0;
        float sigma;
        strip_geom(live ? t : 0, by, bx0, fy, fx0, sigma);
        const float* p = s_buf + by * kBW + bx0;
        auto put = [&](int c, float4 o) {
          if (live) store(bx0, fy, fx0, c, o);
        };
        const bool act = live && !(sigma < kMinSigma);
        const int cnt = __popcll(__ballot(act));
        if (cnt == 0) {
#pragma unroll
          for (int c = 0; c < 3; c++) put(c, lds_load4(p + c * kPlane));
        } else if (cnt >= (STAGE == 3 ? kDenseItems0 : kDenseItems)) {
          if constexpr (STAGE == 3) epf0_strip<false>(p, fx0, fy, sigma, a, put);
          else epf2_strip<false>(p, fx0, fy, sigma, a, put);
        } else {
          // few active strips: the others leave now, the active ones are queued
          if (!act) {
#pragma unroll
            for (int c = 0; c < 3; c++) put(c, lds_load4(p + c * kPlane));
          } else {
            s_list[atomicAdd(&s_cnt, 1)] = (uint16_t)t;
          }
        }
      }
      {
        __syncthreads();
        const int cnt = s_cnt;
#pragma unroll 1
        for (int i0 = 0; i0 < cnt; i0 += kT) {
          if (i0 + (tid & ~63) >= cnt) break;  // wave-uniform
          const int i = i0 + tid;
          const bool on = i < cnt;
          const int t = s_list[min(i, cnt - 1)];
          int by, bx0, fy, fx0;
          float sigma;
          strip_geom(t, by, bx0, fy, fx0, sigma);
          const float* p = s_buf + by * kBW + bx0;
          auto put = [&](int c, float4 o) {
            if (on) store(bx0, fy, fx0, c, o);
          };
          if constexpr (STAGE == 3) epf0_strip<true>(p, fx0, fy, sigma, a, put, bx0 == 0, bx0 == kBW - 4);
          else epf2_strip<true>(p, fx0, fy, sigma, a, put, bx0 == 0, bx0 == kBW - 4);
        }
      }
      return;
    }
    static_assert(kPasses == 1 || STAGE == 0 || last, "held registers of one pass");
    auto geom_of = [&](int t) -> Geom {
      Geom g;
      g.live = t < n;
      const int by = kB - margin + (g.live ? (t / kStrips) * 2 : 0);
      g.bx0 = (t % kStrips) * 4;
      g.fy = ty0 - kB + by;
      g.fx0 = tx0 - kB + g.bx0;
      g.p = s_buf + by * kBW + g.bx0;
      if constexpr (STAGE != 0) {
        const int sx = (min(max(g.fx0, 0), a.w - 1) >> 3) - sbx0;
        const int sy0 = (min(max(g.fy, 0), a.h - 1) >> 3) - sby0, sy1 = (min(max(g.fy + 1, 0), a.h - 1) >> 3) - sby0;
        g.sigma0 = s_sigma[sy0 * kSigW + sx];
        g.sigma1 = s_sigma[sy1 * kSigW + sx];
      } else {
        g.sigma0 = g.sigma1 = 0.0f;
      }
      return g;
    };
    auto put_global = [&](const Geom& g, int r, int c, float4 o) {
      const int fyr = g.fy + r;
      if (g.live && g.bx0 >= kB && g.bx0 < kB + kTW && fyr < a.y1 && fyr < a.h && g.fx0 < a.w)
        gstore_f4<JXLH_NT_FSTORE>(&at_bytes<float>(a.out[c], 4u * ((uint32_t)fyr * a.stride + (uint32_t)g.fx0)), o);
    };
    float4 held[last ? 1 : kPasses][2][3];
    // what `held` carries (EPF stages: at most one entry per thread): -1 nothing, otherwise a 4x2 item t (dense
    // form, both rows) or, with bit 15 set, a compacted strip entry t * 2 + r (row r only, in held[0][0])
    int held_e = -1;
    if constexpr (STAGE == 0) {
#pragma unroll
      for (int pass = 0; pass < kPasses; pass++) {
        if (pass * kT + (tid & ~63) >= n) continue;  // wave-uniform: nothing left for this wave
        const Geom g = geom_of(pass * kT + tid);
        auto put_g = [&](int r, int c, float4 o) {
          if constexpr (last) put_global(g, r, c, o);
          else held[pass][r][c] = o;
        };
#pragma unroll
        for (int c = 0; c < 3; c++) gab_pair(g.p + c * kPlane, c, a.gab_k[c][0], a.gab_k[c][1], a.gab_k[c][2], put_g);
      }
    } else {
      static_assert(STAGE == 0 || STAGE == 3 || kPasses == 1, "one EPF item per thread (EPF0 returned above)");
      // ---- phase A: classify.  dense: this wavefront runs its own 64 items in the dense form; otherwise its
      // active strips go to the list and the wavefront is free to take 64 compacted strips
      auto geom = [&]() -> Geom {
        int tl = tid_kernel;
        asm volatile("" : "+v"(tl));  // recomputed where needed, not kept (see epf1_pair)
        return geom_of(tl);
      };
      bool dense = false;
      int my_slot = -1;
      if ((tid & ~63) < n) {
        const Geom g = geom();
        const bool act0 = g.live && !(g.sigma0 < kMinSigma), act1 = g.live && !(g.sigma1 < kMinSigma);
        const unsigned long long m0 = __ballot(act0), m1 = __ballot(act1);
        const int cnt = __popcll(m0) + __popcll(m1);  // active strips of this wavefront
        // EPF2 as an in-place stage has no compacted form: any active strip makes the wavefront dense
        dense = STAGE == 1 ? cnt > kSparseMax : cnt > 0;
        if (!dense) {
          int slot = 0, base = 0;
          if ((tid & 63) == 0) {
            slot = atomicAdd(&s_nsw, 1);
            if (cnt) base = atomicAdd(&s_cnt, cnt);
          }
          my_slot = __builtin_amdgcn_readfirstlane(slot);
          base = __builtin_amdgcn_readfirstlane(base);
          const unsigned long long below = (1ull << (tid & 63)) - 1ull;
          if (act0) s_list[base + __popcll(m0 & below)] = (uint16_t)(tid * 2);
          if (act1) s_list[base + __popcll(m0) + __popcll(m1 & below)] = (uint16_t)(tid * 2 + 1);
          // rows below MIN_SIGMA: the stage is the identity there (in place: nothing to do; as the last stage:
          // copy out) -- the reference takes the same shortcut per SIMD vector, epf1.rs:72-78
          if constexpr (last) {
#pragma unroll
            for (int r = 0; r < 2; r++) {
              if (r ? act1 : act0) continue;
#pragma unroll
              for (int c = 0; c < 3; c++) put_global(g, r, c, lds_load4(g.p + c * kPlane + r * kBW));
            }
          }
        }
      }
      __syncthreads();  // the list is complete
      // ---- phase B: a wavefront is EITHER dense (its own items) OR takes compacted strips, so `held` has one
      // definition per thread and no live range crosses the other form's code
      if (dense) {
        const Geom g = geom();
        auto put = [&](const Geom& gg, int r, int c, float4 o) {
          if constexpr (last) put_global(gg, r, c, o);
          else held[0][r][c] = o;
        };
        if constexpr (STAGE == 1) epf1_pair(g.p, geom, a, put);
        else epf2_pair(g.p, g.fx0, g.fy, g.sigma0, g.sigma1, a, [&](int r, int c, float4 o) { put(g, r, c, o); });
        held_e = tid;
      } else if constexpr (STAGE == 1) {
        const int cnt = s_cnt;
        const int i = my_slot >= 0 ? my_slot * 64 + (tid & 63) : cnt;
        if (__any(i < cnt)) {
          const bool on = i < cnt;
          const int e = (int)s_list[min(i, cnt - 1)];
          const Geom g = geom_of(e >> 1);
          const int r = e & 1;
          epf1_strip_g(g.p + r * kBW, g.fx0, g.fy + r, r ? g.sigma1 : g.sigma0, g.bx0 == 0, g.bx0 == kBW - 4, a,
                       [&](int c, float4 o) {
                         if constexpr (last) {
                           if (on) put_global(g, r, c, o);
                         } else {
                           held[0][0][c] = o;
                         }
                       });
          if (on) held_e = e | 0x8000;
        }
      }
    }
    if constexpr (!last) {
      __syncthreads();  // every read of the stage's input is done
      if constexpr (STAGE == 0) {
#pragma unroll
        for (int pass = 0; pass < kPasses; pass++) {
          const int t = pass * kT + tid;
          if (t >= n) continue;
          float* d = s_buf + (kB - margin + (t / kStrips) * 2) * kBW + (t % kStrips) * 4;
#pragma unroll
          for (int r = 0; r < 2; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) lds_store4(d + c * kPlane + r * kBW, held[pass][r][c]);
        }
      } else if (held_e >= 0) {
        const bool strip = (held_e & 0x8000) != 0;
        const int t = strip ? (held_e & 0x7fff) >> 1 : held_e, r0 = strip ? (held_e & 1) : 0;
        if (t < n) {
          float* d = s_buf + (kB - margin + (t / kStrips) * 2 + r0) * kBW + (t % kStrips) * 4;
#pragma unroll
          for (int c = 0; c < 3; c++) lds_store4(d + c * kPlane, held[0][0][c]);
          if (!strip) {
#pragma unroll
            for (int c = 0; c < 3; c++) lds_store4(d + c * kPlane + kBW, held[0][1][c]);
          }
        }
      }
      if constexpr (STAGE != 0) {
        if (tid == 0) {
          s_cnt = 0;
          s_nsw = 0;
        }
      }
      __syncthreads();
      if (edge) {
        mirror_fill<kT>(s_buf, margin, tx0, ty0, a.w, a.h, tid);
        __syncthreads();
      }
    }
  };
  constexpr int kMg = kBorder - (GAB ? 1 : 0);      // margin after Gaborish
  constexpr int kMe1 = kMg - (E1 ? 2 : 0);          // after EPF1
  if constexpr (GAB) run_stage(std::integral_constant<int, 0>{}, std::integral_constant<int, kMg>{});
  if constexpr (E0) run_stage(std::integral_constant<int, 3>{}, std::integral_constant<int, 0>{});
  if constexpr (E1) run_stage(std::integral_constant<int, 1>{}, std::integral_constant<int, kMe1>{});
  if constexpr (E2) run_stage(std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{});
}

template <bool GAB, bool E0, bool E1, bool E2>
void launch_variant(hipStream_t s, const FusedArgs& a) {
  const int tiles_x = (a.w + kTW - 1) / kTW, tiles_y = (a.y1 - a.y0 + kTH - 1) / kTH;
  const dim3 grid(tiles_x * ((tiles_y + 7) / 8) * 8);
  hipLaunchKernelGGL((k23_fused_filters<GAB, E0, E1, E2>), grid, dim3(fused_threads<E0>()), 0, s, a);
}

__global__ void k_selftest_recip(uint32_t lo_bits, uint32_t hi_bits, unsigned long long* mismatches) {
  const uint32_t stride = gridDim.x * blockDim.x;
  unsigned long long bad = 0;
  for (uint64_t b = (uint64_t)lo_bits + blockIdx.x * blockDim.x + threadIdx.x; b < hi_bits; b += stride) {
    const float w = __builtin_bit_cast(float, (uint32_t)b);
    const float fast = recip_weight_sum(w);
    const float ref = 1.0f / w;
    bad += __builtin_bit_cast(uint32_t, fast) != __builtin_bit_cast(uint32_t, ref);
  }
  if (bad) atomicAdd(mismatches, bad);
}

}  // namespace

void launch_selftest_recip(hipStream_t s, uint32_t lo_bits, uint32_t hi_bits, unsigned long long* mismatches) {
  hipLaunchKernelGGL(k_selftest_recip, dim3(2048), dim3(256), 0, s, lo_bits, hi_bits, mismatches);
}

// Runs the frame's stage list fused.  Returns 0 if there is nothing to do, 1 if the result is in
// f.tmp (one pass: gab?, epf1?, epf2?), 2 if it is in f.planes (epf_iters == 3: Gaborish + EPF0 go
// planes -> tmp, EPF1 + EPF2 come back tmp -> planes, both raster).
int launch_fused_filters(hipStream_t s, const FrameDev& f, int y0, int y1) {
  const bool gab = f.gab != 0, e0 = f.epf_iters >= 3, e1 = f.epf_iters >= 1, e2 = f.epf_iters >= 2;
  if (!gab && !e1 && !e2) return 0;
  if (y1 <= y0) return e0 ? 2 : 1;
  FusedArgs a;
  for (int c = 0; c < 3; c++) {
    a.in[c] = f.planes[c];
    a.out[c] = f.tmp[c];
    a.scale[c] = f.epf_channel_scale[c];
    for (int k = 0; k < 3; k++) a.gab_k[c][k] = f.gab_k[c][k];
  }
  a.inv_sigma = f.inv_sigma;
  a.stride = (uint32_t)f.plane_stride;
  a.sigma_stride = (uint32_t)f.xblocks;
  a.w = f.xsize;
  a.h = f.ysize;
  a.y0 = y0;
  a.y1 = y1;
  a.sm0 = f.epf_sm[0];
  a.bsm0 = f.epf_bsm[0];
  a.sm1 = f.epf_sm[1];
  a.bsm1 = f.epf_bsm[1];
  a.sm2 = f.epf_sm[2];
  a.bsm2 = f.epf_bsm[2];
  a.tiled_in = f.tiled;
  a.xblocks = f.xblocks;
  if (e0) {
    // first pass over the band widened by the second pass's 3-pixel border (kept 4-aligned for
    // the tiled staging path)
    a.y0 = max(0, y0 - 4);
    a.y1 = min(f.ysize, y1 + 3);
    if (gab) launch_variant<true, true, false, false>(s, a);
    else launch_variant<false, true, false, false>(s, a);
    for (int c = 0; c < 3; c++) {
      a.in[c] = f.tmp[c];
      a.out[c] = f.planes[c];
    }
    a.tiled_in = 0;
    a.y0 = y0;
    a.y1 = y1;
    launch_variant<false, false, true, true>(s, a);
    return 2;
  }
  if (gab && e1 && e2) launch_variant<true, false, true, true>(s, a);
  else if (gab && e1) launch_variant<true, false, true, false>(s, a);
  else if (gab) launch_variant<true, false, false, false>(s, a);
  else if (e1 && e2) launch_variant<false, false, true, true>(s, a);
  else launch_variant<false, false, true, false>(s, a);
  return 1;
}

}  // namespace jxlh
