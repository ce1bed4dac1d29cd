// K2+K3 fused: Gaborish -> EPF1 -> EPF2 in ONE pass over HBM (12 B/px in, 12 B/px out).
//
// One workgroup owns a 56 x kTH output tile.  The three XYB channels of the tile plus a 4-pixel
// halo (1 Gaborish + 2 EPF1 + 1 EPF2, jxl/src/render/mod.rs:28-36) are staged once in LDS with
// 16-byte coalesced loads; every stage then runs LDS -> LDS *in place* on a region that shrinks by
// its own border (all threads compute their outputs into registers, barrier, write back), and
// only the last stage writes to HBM.  One buffer instead of a ping-pong pair halves the LDS per
// pixel: more workgroups per CU (see JXLH_FUSED_TH below).
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

// Tile heights and workgroup sizes of the two geometries (one 4x2 item per thread in the EPF stages:
// (TH + 6) / 2 * 16 <= THREADS); which stage list uses which: see the namespaces below.
#ifndef JXLH_FUSED_TH
#define JXLH_FUSED_TH 56
#endif
#ifndef JXLH_FUSED_THREADS
#define JXLH_FUSED_THREADS 512
#endif
#ifndef JXLH_FUSED_TH_LOW
#define JXLH_FUSED_TH_LOW 24
#endif
#ifndef JXLH_FUSED_THREADS_LOW
#define JXLH_FUSED_THREADS_LOW 256
#endif
// Tiles per workgroup of the low geometry.  > 1: a workgroup walks a column strip and keeps the 2 * border input rows two
// consecutive tiles share in LDS (each input row fetched once, like the reference's ring buffers) -- built and measured
// in round 6 and SLOWER: 0.36 (1) / 0.42 (3) / 0.48 (6) / 0.55 ms (12) on the 8K spec population: the kernel lives on
// six short workgroups per CU in different phases (load, stages, store); a strip serialises those phases inside a
// workgroup, needs 6 KB more LDS (five per CU) and 96 VGPRs, and lengthens the launch's tail
// (profiles/r06_e_filter_strips.txt).  Kept as a build option; the halo traffic is attacked through the L2 instead
// (JXLH_FUSED_BAND).
#ifndef JXLH_FUSED_CHUNKS
#define JXLH_FUSED_CHUNKS 1
#endif
// Tile rows of an XCD's band in the blockIdx -> tile map (1 = an XCD walks along one tile row: rounds 1-5).  > 1 puts
// vertically adjacent tiles on the same XCD, walked column by column, so that the halo ROWS (a third of a 24-row tile's
// fetches) can hit in that XCD's L2 -- also built in round 6 and also slower: 0.363 (1) / 0.379 (2) / 0.395 (4) / 0.433
// (8) / 0.449 ms (16): the row walk streams the planes through memory in address order, the column walk strides by
// 256 KB (profiles/r06_e_filter_bands.txt).
#ifndef JXLH_FUSED_BAND
#define JXLH_FUSED_BAND 1
#endif
#ifndef JXLH_FUSED_STRIP_WPE
#define JXLH_FUSED_STRIP_WPE 5  // the strip form's 32 KB of LDS allow five workgroups per CU
#endif
#ifndef JXLH_FUSED_WAVES_PER_EU
#define JXLH_FUSED_WAVES_PER_EU 6
#endif
#ifndef JXLH_FUSED_E0_WPE
#define JXLH_FUSED_E0_WPE 3
#endif
#ifndef JXLH_FUSED_E0_THREADS
#define JXLH_FUSED_E0_THREADS 128
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

// Two tile geometries of the same code (k_filters_fused_tile.inc), chosen per stage list by launch_fused_filters:
//   low   24 output rows, 256 threads: 25 KB of LDS, six workgroups of four waves per CU -- twice as many tiles in
//         different phases (load, stages, store) at any time.  Gaborish + EPF1 (+ EPF2): 8K spec population
//         0.406-0.410 -> 0.364-0.366 ms on one box, 16K all types 1.516 -> 1.416 (profiles/r04_i_filter_tiles.txt)
//   tall  56 output rows, 512 threads: 50 KB, three workgroups per CU, 14 % halo rows instead of 33 %.  Gaborish alone
//         (0.0925 vs 0.0961 ms at 4096^2) is faster with it; the EPF1 + EPF2 launch of epf_iters = 3 (raster input) does
//         not care.  Gaborish + EPF0 takes the low tile with 128-thread workgroups (see launch_fused_filters).
// (16 / 192, 20 / 192, 32 / 320, 40 / 384 and 8 / 128 rows / threads are slower than either.)
namespace tall {
#define JXLH_TILE_TH JXLH_FUSED_TH
#define JXLH_TILE_THREADS JXLH_FUSED_THREADS
#include "k_filters_fused_tile.inc"
#undef JXLH_TILE_TH
#undef JXLH_TILE_THREADS
}  // namespace tall
namespace low {
#define JXLH_TILE_TH JXLH_FUSED_TH_LOW
#define JXLH_TILE_THREADS JXLH_FUSED_THREADS_LOW
#include "k_filters_fused_tile.inc"
#undef JXLH_TILE_TH
#undef JXLH_TILE_THREADS
}  // namespace low
using tall::recip_weight_sum;

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
    // Gaborish + EPF0: the low tile with 128-thread workgroups (six of two waves per CU at its 153 VGPRs): 3.13 vs 3.21-3.23 ms
    // for both launches at 16K (tall / 256 threads); low / 256: 3.53, low / 192: 3.38
    if (gab) low::launch_variant<true, true, false, false>(s, a);
    else low::launch_variant<false, true, false, false>(s, a);
    for (int c = 0; c < 3; c++) {
      a.in[c] = f.tmp[c];
      a.out[c] = f.planes[c];
    }
    a.tiled_in = 0;
    a.y0 = y0;
    a.y1 = y1;
    tall::launch_variant<false, false, true, true>(s, a);
    return 2;
  }
  // (the low tile as column strips of JXLH_FUSED_CHUNKS tiles whose shared input rows stay in LDS: round 6)
  constexpr int NC = JXLH_FUSED_CHUNKS;
  if (gab && e1 && e2) low::launch_variant<true, false, true, true, NC>(s, a);
  else if (gab && e1) low::launch_variant<true, false, true, false, NC>(s, a);
  else if (gab) tall::launch_variant<true, false, false, false>(s, a);
  else if (e1 && e2) low::launch_variant<false, false, true, true, NC>(s, a);
  else low::launch_variant<false, false, true, false, NC>(s, a);
  return 1;
}

}  // namespace jxlh
