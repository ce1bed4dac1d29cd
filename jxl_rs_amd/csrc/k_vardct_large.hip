// K1, family E: the 64..256-pixel transforms of a frame (see k_vardct.hip for the binning that feeds them).  Its own
// translation unit: the register-resident length-64 leaves want no SLP packing (v_pk_* pairs cost moves and registers
// on gfx950, Makefile), while the special-transform kernel of k_vardct.hip measured better with it.
#include "k_vardct_common.h"
#include "varblock_large.h"

namespace jxlh {
namespace {

// family E: DCT64X64 .. DCT256X256 (varblock_large.h).  The two separable passes are separate launches over uniform
// SLAB units (4096 samples of one channel of one varblock; a 256x256 varblock is 16 slabs per pass and channel, a
// 64x64 one 1), one wavefront per unit, four independent wavefronts per workgroup:
//   k1_large_units   one thread per large varblock: reserves its slabs in the unit list (item | slab << 24)
//   k1_large_llf     one wavefront per (varblock, channel): LLF-from-LF of the cy x cx patch -> llf planes (the corner
//                    pass 1 substitutes, transform.rs:450); the values sit at the linear positions of the varblock's
//                    own block rectangle, so the planes have the LF image's size
//   k1_large_pass<1> unit = (slab of lines, channel): dequantise + LLF corner + horizontal IDCT -> output rectangle
//   k1_large_pass<2> unit = (slab of pixel columns, channel): vertical IDCT in place
// (the varblock's own output rectangle is the inter-pass scratch; it stays in L2 / Infinity Cache between the launches)
__global__ __launch_bounds__(256) void k1_large_units(const WorkLists wl, uint32_t* __restrict__ units) {
  const int count = wl.counts[(kClsLarge) * kCountPitch];
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < count; e += gridDim.x * blockDim.x) {
    const int type = (int)(wl.items[kClsLarge][e].packed >> 20) & 31;
    const int n = max(1, covered_x(type) * covered_y(type) * 64 / kLargeSlab);  // 64x32 / 32x64: half a slab
    const int base = atomicAdd(&wl.counts[(kNumClasses) * kCountPitch], n);
    // capacity = nblocks / 32 + 16 >= the units any valid map can need (k1_scan drops overlapping varblocks)
    for (int s = 0; s < n; s++) units[base + s] = (uint32_t)e | ((uint32_t)s << 24);
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

template <int PASS>
__global__ __launch_bounds__(kLargeThreads) void k1_large_pass(const FrameDev f, const WorkLists wl,
                                                               const uint32_t* __restrict__ units,
                                                               const float* __restrict__ llf_planes, size_t llf_plane_stride) {
  __shared__ __attribute__((aligned(16))) float s_tile[kLargeWaves * kLargeTile];
  const int total = wl.counts[(kNumClasses) * kCountPitch] * 3;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* tile = s_tile + wave * kLargeTile;
  const float b0 = f.quant_biases[0], b1 = f.quant_biases[1], b2 = f.quant_biases[2], b3 = f.quant_biases[3];
  for (int u = blockIdx.x * kLargeWaves + wave; u < total; u += gridDim.x * kLargeWaves) {
    const uint32_t unit = units[u / 3];
    const int e = (int)(unit & 0xffffffu), slab = (int)(unit >> 24), ch = u % 3;
    const WorkItem it = wl.items[kClsLarge][e];
    BlockInfo bi;
    decode_item(f, it, &bi);
    const int type = (int)(it.packed >> 20) & 31;
    const LargeGeom g(type);
    const PixLayout lay = pix_layout(f);
    float* plane = f.planes[ch] + bi.px_off[ch];
    if constexpr (PASS == 2) {
      wave_large_pass2(g, slab * g.LX, plane, lay, tile, lane);
    } else {
      const int q = quant_table_for_type(type);
      const float* __restrict__ table = f.tables + f.table_offset[q];
      const int tsize = quant_table_size(q);
      // one dequantiser for the three channels (the channel is uniform over the wavefront): Y alone, or the
      // channel's own coefficient plus the chroma-from-luma multiple of the dequantised Y (group.rs:100-133)
      const int32_t* __restrict__ qy = f.coeffs + bi.coef_off + kGroupArea;
      const int32_t* __restrict__ qc = f.coeffs + bi.coef_off + ch * kGroupArea;
      const float* __restrict__ ty = table + tsize;
      const float* __restrict__ tc = table + ch * tsize;
      const float sdy = bi.sdy, sdc = ch == 0 ? bi.sdy * f.x_dm : bi.sdy * f.b_dm;
      const float cc = ch == 0 ? bi.x_cc : bi.b_cc, bc = ch == 0 ? b0 : b2;
      const bool luma = ch == 1;
      const float* __restrict__ llf = llf_planes + (size_t)ch * llf_plane_stride;
      const int lf_off = bi.lf_off[ch], xblocks = f.xblocks, cx = g.cx;
      wave_large_pass1(
          g, slab * g.LV,
          [&](int k) {
            const int4 iy = *reinterpret_cast<const int4*>(qy + k);
            const float4 wy = *reinterpret_cast<const float4*>(ty + k);
            const int vy[4] = {iy.x, iy.y, iy.z, iy.w};
            const float fy[4] = {wy.x, wy.y, wy.z, wy.w};
            int vc[4] = {0, 0, 0, 0};
            float fc[4] = {0.f, 0.f, 0.f, 0.f};
            if (!luma) {
              const int4 ic = *reinterpret_cast<const int4*>(qc + k);
              const float4 wc = *reinterpret_cast<const float4*>(tc + k);
              vc[0] = ic.x; vc[1] = ic.y; vc[2] = ic.z; vc[3] = ic.w;
              fc[0] = wc.x; fc[1] = wc.y; fc[2] = wc.z; fc[3] = wc.w;
            }
            // adjust_quant_bias divides only for |q| >= 2 (group.rs:91-95); most slabs of a large varblock hold
            // nothing but 0 / +-1 (high frequencies): a wavefront without a larger value skips the divisions
            bool big = false;
#pragma unroll
            for (int i = 0; i < 4; i++) big |= (unsigned)(vy[i] + 1) > 2u || (unsigned)(vc[i] + 1) > 2u;
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
          },
          [&](int i) { return llf[llf_offset(lf_off, xblocks, cx, i)]; }, plane, lay, tile, lane);
    }
  }
}

}  // namespace

void launch_vardct_large(hipStream_t s, const FrameDev& f, const WorkLists& wl, int nblk, uint32_t* large_units,
                         size_t unit_capacity, size_t nblocks) {
  auto grid_for = [](long work_items, int items_per_wg, int cap) {
    long g = (work_items + items_per_wg - 1) / items_per_wg;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
  };
  // the large class: unit list, LLF corners, then one launch per separable pass.  All four exit at once when the
  // class is empty
  float* llf_planes = reinterpret_cast<float*>(
      (reinterpret_cast<uintptr_t>(large_units + unit_capacity) + 63) & ~(uintptr_t)63);
  hipLaunchKernelGGL(k1_large_units, dim3(grid_for(nblk / 64 + 1, 256, 64)), dim3(256), 0, s, wl, large_units);
  hipLaunchKernelGGL(k1_large_llf, dim3(grid_for(3L * (nblk / 32 + 1), kLargeWaves, 512)), dim3(kLargeThreads), 0, s, f, wl,
                     llf_planes, nblocks);
  // two 66 KB workgroups fit a CU: 512 is the resident capacity
  const dim3 glarge(grid_for(3L * (nblk / 32 + 1), kLargeWaves, 512));
  hipLaunchKernelGGL(k1_large_pass<1>, glarge, dim3(kLargeThreads), 0, s, f, wl, large_units, llf_planes, nblocks);
  hipLaunchKernelGGL(k1_large_pass<2>, glarge, dim3(kLargeThreads), 0, s, f, wl, large_units, llf_planes, nblocks);
}

}  // namespace jxlh
