// K1, family E: the 64..256-pixel transforms of a frame (see k_vardct.hip for the binning that feeds them).  Its own
// translation unit: the register-resident length-64 leaves want no SLP packing (v_pk_* pairs cost moves and registers
// on gfx950, Makefile), while the special-transform kernel of k_vardct.hip measured better with it.
#include "k_vardct_common.h"
#include "varblock_large.h"

#ifndef JXLH_LLF_GRID
#define JXLH_LLF_GRID 1024  // four 32 KB workgroups of k1_large_llf fit a CU
#endif

namespace jxlh {
namespace {

// family E: DCT64X64 .. DCT256X256 (varblock_large.h).  Work is cut into SLABS: 4096 samples of one channel of one
// varblock, what one wavefront transforms in registers.  Two routes (round 4, profiles/r04_e_large_path.txt):
//   * below 256 pixels (64x64, 64x32, 32x64, 128x64, 64x128, 128x128) a channel is at most four slabs, so ONE launch
//     does both separable passes: k1_large_fused keeps the pass-1 result in the workgroup's four wave tiles and pass
//     2 reads its columns from there and stores from registers.  No intermediate store / reload: the kernels are
//     bound by instruction issue at the two waves per SIMD the LDS tiles allow (VALU busy ~55 %), not by bytes.
//   * 256-pixel types (a channel is up to 256 KiB, more than a CU's LDS): the two passes are separate launches over
//     uniform slab units, one wavefront per unit, four independent wavefronts per workgroup; the varblock's own
//     output rectangle is the inter-pass scratch (it stays in L2 / Infinity Cache between the launches).
//   k1_large_units   one lane per large varblock: appends it to the fused list of its slab count (1 / 2 / 4) or
//                    reserves its slabs in the two-pass unit list (item | slab << 24); one atomic per wave and list,
//                    every list counter on its own 128-byte line (counters sharing a line cost 100 us at 16K)
//   k1_large_llf     one wavefront per (varblock, channel): LLF-from-LF of the cy x cx patch -> llf planes (the corner
//                    pass 1 substitutes, transform.rs:450); the values sit at the linear positions of the varblock's
//                    own block rectangle, so the planes have the LF image's size
//   k1_large_fused   workgroup unit = four slab slots (one 128x128 channel, two 128x64 / 64x128, four smaller ones)
//   k1_large_pass<1> unit = (slab of lines, channel): dequantise + LLF corner + horizontal IDCT -> output rectangle
//   k1_large_pass<2> unit = (slab of pixel columns, channel): vertical IDCT in place
// Every body is instantiated per transform type (the geometry folds to constants); pass 1 keeps eight rounds of raw
// coefficient loads in flight per lane.
// unit lists behind each other in the `units` allocation (launch_vardct_large carves them the same way)
struct LargeLists {
  uint32_t* two_pass;  // item | slab << 24 of the 256-pixel types, counter kNumClasses * kCountPitch
  uint32_t* fused[3];  // items of the types whose channel is 1 (64x64, 64x32, 32x64), 2 (128x64, 64x128) or 4 (128x128)
                       // slabs: one workgroup transforms 4 / 2 / 1 of them at a time; counters on the three lines behind
};

__global__ __launch_bounds__(256) void k1_large_units(const WorkLists wl, const LargeLists ll, int fuse) {
  const int count = wl.counts[(kClsLarge) * kCountPitch];
  const int lane = threadIdx.x & 63;
  int* counters = wl.counts + (kNumClasses) * kCountPitch;
  // one atomic per wave and list: the lanes that append `n` entries to the list behind `counter` get consecutive places
  auto reserve = [&](bool mine, int n, int* counter) {
    const unsigned long long m = __ballot(mine);
    if (m == 0) return 0;
    // exclusive prefix of n over the lanes of m and the total, one step per set lane
    int before = 0, total = 0;
    for (unsigned long long r = m; r; r &= r - 1) {
      const int l = __ffsll((long long)r) - 1;
      const int nl = __shfl(n, l);
      before += l < lane ? nl : 0;
      total += nl;
    }
    const int leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, total);
    return __shfl(base, leader) + before;
  };
  const int stride = gridDim.x * blockDim.x;
  for (int e0 = blockIdx.x * blockDim.x + (threadIdx.x & ~63); e0 < count; e0 += stride) {  // uniform over the wave
    const int e = e0 + lane;
    const bool valid = e < count;
    const int type = valid ? (int)(wl.items[kClsLarge][e].packed >> 20) & 31 : 18;
    const int n = max(1, covered_x(type) * covered_y(type) * 64 / kLargeSlab);  // 64x32 / 32x64: half a slab
    const bool small = fuse && max(covered_x(type), covered_y(type)) < 32;
    // capacities >= what any valid map can need (k1_scan drops overlapping varblocks)
    const int i4 = reserve(valid && small && n == 4, 1, counters + 3 * kCountPitch);
    const int i2 = reserve(valid && small && n == 2, 1, counters + 2 * kCountPitch);
    const int i1 = reserve(valid && small && n == 1, 1, counters + kCountPitch);
    const int base = reserve(valid && !small, n, counters);
    if (!valid) continue;
    if (small) {
      if (n == 4) ll.fused[2][i4] = (uint32_t)e;
      else if (n == 2) ll.fused[1][i2] = (uint32_t)e;
      else ll.fused[0][i1] = (uint32_t)e;
    } else {
      for (int s = 0; s < n; s++) ll.two_pass[base + s] = (uint32_t)e | ((uint32_t)s << 24);
    }
  }
}

// offset of LLF value i of a varblock inside an llf plane: linear position i of its cy x cx block rectangle
__device__ __forceinline__ int llf_offset(int lf_off, int xblocks, int cx, int i) {
  return lf_off + (i / cx) * xblocks + (i % cx);
}

__global__ __launch_bounds__(kLargeThreads) void k1_large_llf(const FrameDev f, const WorkLists wl, float* __restrict__ llf_planes,
                                                              size_t llf_plane_stride) {
  __shared__ float s_scratch[kLargeWaves][2 * 1024];
  const int count = wl.counts[(kClsLarge) * kCountPitch] * 3;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int u = blockIdx.x * kLargeWaves + wave; u < count; u += gridDim.x * kLargeWaves) {
    const int e = u / 3, ch = u % 3;
    const WorkItem it = wl.items[kClsLarge][e];
    BlockInfo bi;
    decode_item(f, it, &bi);
    const int type = (int)(it.packed >> 20) & 31;
    const int cx = covered_x(type), cy = covered_y(type);
    float* out = s_scratch[wave] + 1024;
    wave_large_llf(f.lf[ch] + bi.lf_off[ch], f.xblocks, cy, cx, s_scratch[wave], out, lane);
    float* dst = llf_planes + (size_t)ch * llf_plane_stride;
    for (int i = lane; i < cx * cy; i += 64) dst[llf_offset(bi.lf_off[ch], f.xblocks, cx, i)] = out[i];
    wave_sync();
  }
}

// One dequantiser for the three channels of a large varblock (the channel is uniform over the wavefront): Y alone, or
// the channel's own coefficient plus the chroma-from-luma multiple of the dequantised Y (group.rs:100-133).
struct LargeCoef {
  const int32_t* __restrict__ qy;
  const int32_t* __restrict__ qc;
  const float* __restrict__ ty;
  const float* __restrict__ tc;
  const float* __restrict__ llf;
  float sdy, sdc, cc, bc, b1, b3;
  int lf_off, xblocks, cx;
  bool luma;
  __device__ LargeCoef(const FrameDev& f, const BlockInfo& bi, int type, int ch, const float* llf_planes, size_t llf_plane_stride) {
    const int q = quant_table_for_type(type);
    const float* __restrict__ table = f.tables + f.table_offset[q];
    const int tsize = quant_table_size(q);
    qy = f.coeffs + bi.coef_off + kGroupArea;
    qc = f.coeffs + bi.coef_off + ch * kGroupArea;
    ty = table + tsize;
    tc = table + ch * tsize;
    sdy = bi.sdy;
    sdc = ch == 0 ? bi.sdy * f.x_dm : bi.sdy * f.b_dm;
    cc = ch == 0 ? bi.x_cc : bi.b_cc;
    bc = ch == 0 ? f.quant_biases[0] : f.quant_biases[2];
    b1 = f.quant_biases[1];
    b3 = f.quant_biases[3];
    luma = ch == 1;
    llf = llf_planes + (size_t)ch * llf_plane_stride;
    lf_off = bi.lf_off[ch];
    xblocks = f.xblocks;
    cx = covered_x(type);
  }
  struct Raw {
    int4 y, c;
  };
  __device__ __forceinline__ Raw load(int k) const {
    Raw r;
    r.y = *reinterpret_cast<const int4*>(qy + k);
    r.c = luma ? make_int4(0, 0, 0, 0) : *reinterpret_cast<const int4*>(qc + k);
    return r;
  }
  __device__ __forceinline__ float4 finish(int k, const Raw& raw) const {
    const float4 wy = *reinterpret_cast<const float4*>(ty + k);
    const int vy[4] = {raw.y.x, raw.y.y, raw.y.z, raw.y.w};
    const int vc[4] = {raw.c.x, raw.c.y, raw.c.z, raw.c.w};
    const float fy[4] = {wy.x, wy.y, wy.z, wy.w};
    float fc[4] = {0.f, 0.f, 0.f, 0.f};
    if (!luma) {
      const float4 wc = *reinterpret_cast<const float4*>(tc + k);
      fc[0] = wc.x; fc[1] = wc.y; fc[2] = wc.z; fc[3] = wc.w;
    }
    // adjust_quant_bias divides only for |q| >= 2 (group.rs:91-95); most slabs of a large varblock hold
    // nothing but 0 / +-1 (high frequencies): a wavefront without a larger value skips the divisions
    // (one unsigned maximum over q + 1 instead of a comparison per value)
    auto umax3 = [](unsigned a, unsigned b, unsigned c) { return max(max(a, b), c); };
    unsigned top = max(umax3((unsigned)(vy[0] + 1), (unsigned)(vy[1] + 1), (unsigned)(vy[2] + 1)), (unsigned)(vy[3] + 1));
    if (!luma) top = max(umax3(top, (unsigned)(vc[0] + 1), (unsigned)(vc[1] + 1)), max((unsigned)(vc[2] + 1), (unsigned)(vc[3] + 1)));
    const bool big = top > 2u;
    float r[4];
    if (__builtin_amdgcn_ballot_w64(big) != 0) {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const float y = adjust_quant_bias(vy[i], b1, b3) * (fy[i] * sdy);
        r[i] = luma ? y : __builtin_fmaf(cc, y, adjust_quant_bias(vc[i], bc, b3) * (fc[i] * sdc));
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const float y = ((float)vy[i] * b1) * (fy[i] * sdy);
        r[i] = luma ? y : __builtin_fmaf(cc, y, ((float)vc[i] * bc) * (fc[i] * sdc));
      }
    }
    return make_float4(r[0], r[1], r[2], r[3]);
  }
  __device__ __forceinline__ float llf_at(int i) const { return llf[llf_offset(lf_off, xblocks, cx, i)]; }
};

template <int PASS>
__global__ __launch_bounds__(kLargeThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void k1_large_pass(const FrameDev f, const WorkLists wl,
                                                               const uint32_t* __restrict__ units,
                                                               const float* __restrict__ llf_planes, size_t llf_plane_stride) {
  __shared__ __attribute__((aligned(16))) float s_tile[kLargeWaves * kLargeTile];
  const int total = wl.counts[(kNumClasses) * kCountPitch] * 3;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* tile = s_tile + wave * kLargeTile;
  for (int u = blockIdx.x * kLargeWaves + wave; u < total; u += gridDim.x * kLargeWaves) {
    const uint32_t unit = units[u / 3];
    const int e = (int)(unit & 0xffffffu), slab = (int)(unit >> 24), ch = u % 3;
    const WorkItem it = wl.items[kClsLarge][e];
    BlockInfo bi;
    decode_item(f, it, &bi);
    const int type = (int)(it.packed >> 20) & 31;
    const PixLayout lay = pix_layout(f);
    float* plane = f.planes[ch] + bi.px_off[ch];
    // one specialised body per 256-pixel type (the geometry folds to constants), the generic one for the rest (the
    // smaller types come here only with JXLH_LARGE_FUSED=0)
    auto body = [&](const LargeGeom g, int t) {
      if constexpr (PASS == 2) {
        wave_large_pass2(g, slab * g.LX, plane, lay, tile, lane);
      } else {
        const LargeCoef coef(f, bi, t, ch, llf_planes, llf_plane_stride);
        wave_large_pass1_stage_bulk(g, slab * g.LV, coef, tile, lane);
        wave_tile_store<true>(tile, large_pitch(g.C), plane, lay, 0, slab * g.LV, g.C, g.LV, lane);
        wave_sync();
      }
    };
    switch (type) {
      case 24: body(LargeGeom(24), 24); break;
      case 25: body(LargeGeom(25), 25); break;
      case 26: body(LargeGeom(26), 26); break;
      default: body(LargeGeom(type), type); break;
    }
  }
}

// The transforms below 256 pixels in ONE launch: a channel of such a varblock is at most kLargeWaves slabs, so the
// pass-1 result stays in the workgroup's four wave tiles and pass 2 reads its columns from there.  A workgroup unit is
// four slab slots: one 128x128 channel, two 128x64 / 64x128 channels or four <= 64x64 ones; wave w works on slab
// w % S of sub-unit w / S in both passes, the two barriers are the only workgroup-wide steps.
__global__ __launch_bounds__(kLargeThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void k1_large_fused(const FrameDev f, const WorkLists wl, const LargeLists ll,
                                                                const float* __restrict__ llf_planes, size_t llf_plane_stride) {
  __shared__ __attribute__((aligned(16))) float s_tile[kLargeWaves * kLargeTile];
  const int* cnt = wl.counts + (kNumClasses) * kCountPitch;
  const int n1 = cnt[kCountPitch], n2 = cnt[2 * kCountPitch], n4 = cnt[3 * kCountPitch];
  const int g1 = (n1 + 3) >> 2, g2 = (n2 + 1) >> 1;
  const int total = (g1 + g2 + n4) * 3;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const PixLayout lay = pix_layout(f);
  for (int u = blockIdx.x; u < total; u += gridDim.x) {
    const int gu = u / 3, ch = u % 3;
    // which list, how many slabs per channel there, this wave's item and slab
    int slot, first, n;
    if (gu < n4) slot = 2, first = gu, n = n4;
    else if (gu < n4 + g2) slot = 1, first = (gu - n4) * 2, n = n2;
    else slot = 0, first = (gu - n4 - g2) * 4, n = n1;
    const int S = 1 << slot, sub = wave >> slot, slab = wave & (S - 1);
    const bool live = first + sub < n;
    int type = 18;
    float* plane = nullptr;
    if (live) {
      const int e = (int)ll.fused[slot][first + sub];
      const WorkItem it = wl.items[kClsLarge][e];
      BlockInfo bi;
      decode_item(f, it, &bi);
      type = (int)(it.packed >> 20) & 31;
      plane = f.planes[ch] + bi.px_off[ch];
      float* tile = s_tile + wave * kLargeTile;
      // one specialised body per transform type (the geometry folds to constants); the type is uniform over the wave
      auto pass1 = [&](auto type_tag) {
        constexpr int T = decltype(type_tag)::value;
        const LargeGeom g(T);
        const LargeCoef coef(f, bi, T, ch, llf_planes, llf_plane_stride);
        wave_large_pass1_stage_bulk(g, slab * g.LV, coef, tile, lane);
      };
      switch (type) {
        case 18: pass1(std::integral_constant<int, 18>{}); break;
        case 19: pass1(std::integral_constant<int, 19>{}); break;
        case 20: pass1(std::integral_constant<int, 20>{}); break;
        case 21: pass1(std::integral_constant<int, 21>{}); break;
        case 22: pass1(std::integral_constant<int, 22>{}); break;
        default: pass1(std::integral_constant<int, 23>{}); break;
      }
    }
    __syncthreads();
    if (live) {
      int lane2 = lane;
      asm volatile("" : "+v"(lane2));  // nothing of pass 2 is computed (and kept in registers) ahead of pass 1
      const float* tiles = s_tile + (sub << slot) * kLargeTile;
      auto pass2 = [&](auto type_tag) {
        constexpr int T = decltype(type_tag)::value;
        const LargeGeom g(T);
        wave_large_pass2_lds<covered_y(T) * 8>(g, slab * g.LX, tiles, plane, lay, lane2);
      };
      switch (type) {
        case 18: pass2(std::integral_constant<int, 18>{}); break;
        case 19: pass2(std::integral_constant<int, 19>{}); break;
        case 20: pass2(std::integral_constant<int, 20>{}); break;
        case 21: pass2(std::integral_constant<int, 21>{}); break;
        case 22: pass2(std::integral_constant<int, 22>{}); break;
        default: pass2(std::integral_constant<int, 23>{}); break;
      }
    }
    __syncthreads();  // the tiles are free again
  }
}

}  // namespace

void launch_vardct_large(hipStream_t s, const FrameDev& f, const WorkLists& wl, int nblk, uint32_t* large_units,
                         size_t unit_capacity, size_t nblocks) {
  auto grid_for = [](long work_items, int items_per_wg, int cap) {
    long g = (work_items + items_per_wg - 1) / items_per_wg;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
  };
  // the large class: unit lists, LLF corners, the one-launch path, then one launch per separable pass of the 256-pixel
  // types.  All five exit at once when the class is empty
  const size_t nb = nblocks;
  LargeLists ll;
  ll.two_pass = large_units;
  ll.fused[0] = ll.two_pass + (nb / 32 + 16);
  ll.fused[1] = ll.fused[0] + (nb / 32 + 16);
  ll.fused[2] = ll.fused[1] + (nb / 128 + 16);
  float* llf_planes = reinterpret_cast<float*>(
      (reinterpret_cast<uintptr_t>(large_units + unit_capacity) + 63) & ~(uintptr_t)63);
  static const int fuse = [] {  // development switch: JXLH_LARGE_FUSED=0 sends every large type through the two passes
    const char* e = getenv("JXLH_LARGE_FUSED");
    return e ? atoi(e) : 1;
  }();
  hipLaunchKernelGGL(k1_large_units, dim3(grid_for(nblk / 64 + 1, 256, 64)), dim3(256), 0, s, wl, ll, fuse);
  hipLaunchKernelGGL(k1_large_llf, dim3(grid_for(3L * (nblk / 32 + 1), kLargeWaves, JXLH_LLF_GRID)), dim3(kLargeThreads), 0, s, f, wl,
                     llf_planes, nblocks);
  // two 77 KiB workgroups fit a CU: 512 is the resident capacity
  const dim3 glarge(grid_for(3L * (nblk / 32 + 1), kLargeWaves, 512));
  if (fuse) hipLaunchKernelGGL(k1_large_fused, glarge, dim3(kLargeThreads), 0, s, f, wl, ll, llf_planes, nblocks);
  hipLaunchKernelGGL(k1_large_pass<1>, glarge, dim3(kLargeThreads), 0, s, f, wl, ll.two_pass, llf_planes, nblocks);
  hipLaunchKernelGGL(k1_large_pass<2>, glarge, dim3(kLargeThreads), 0, s, f, wl, ll.two_pass, llf_planes, nblocks);
}

}  // namespace jxlh
