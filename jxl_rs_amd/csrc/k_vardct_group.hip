// K1 -- per-group dequantisation + chroma-from-luma + LLF-from-LF + variable-size IDCT.
//
// Replaces the `if let Some(pixels)` branch of decode_vardct_group
// (jxl/src/frame/group.rs:579-611): dequant_block (:137-177), dequant_lane (:100-133),
// adjust_quant_bias (:85-96), the LF patch copy (:227-235), transform_to_pixels and
// the copy into the group planes (:237-250).
//
// Launch geometry: `split` workgroups per 256x256 group (split = 1 is the
// one-threadblock-per-group geometry; larger splits only raise occupancy on small
// frames).  Every workgroup
//   1. loads the group's 32x32 transform map into LDS and prefix-scans the varblock
//      sizes in raster order -> coefficient offset of every varblock (the reference
//      lays varblocks back to back in decode order, group.rs:440, :612);
//   2. buckets the varblocks whose top-left block lies in its band of block rows by
//      transform type (LDS atomics; order inside a bucket is irrelevant);
//   3. walks the buckets: each wavefront pulls batches of same-shape varblocks,
//      stages the dequantised coefficients of one channel into its private LDS tile
//      with 16-byte coalesced loads, and runs the wave-level cores of varblock_core.h.
//      Channel order Y, X, B: the dequantised Y stays in VGPRs for the two
//      chroma-from-luma FMAs.
// HBM traffic is the compulsory 12 B/px in + 12 B/px out (+ maps); arithmetic is f32
// with the reference's operation order (bit-exact vs the FMA build of the oracle).
#include "varblock_core.h"
#include "varblock_large.h"

namespace jxlh {
namespace {

constexpr int kWaves = 4;
constexpr int kThreads = kWaves * 64;
constexpr int kWaveBuf = 2624;  // floats; max kTile over the 9 shapes <= 32

// DCT32X8 (tall, T pitch 12 x 32 rows) would need 3136 words at NB = 8; it runs 4 per batch.
using S8x8 = Shape<8, 8>;
using S16x16 = Shape<16, 16>;
using S32x32 = Shape<32, 32>;
using S16x8 = Shape<16, 8>;
using S8x16 = Shape<8, 16>;
using S32x8 = Shape<32, 8, 4>;
using S8x32 = Shape<8, 32>;
using S32x16 = Shape<32, 16>;
using S16x32 = Shape<16, 32>;
static_assert(S8x8::kTile <= kWaveBuf && S16x8::kTile <= kWaveBuf && S8x16::kTile <= kWaveBuf &&
                  S16x16::kTile <= kWaveBuf && S32x8::kTile <= kWaveBuf && S8x32::kTile <= kWaveBuf &&
                  S32x16::kTile <= kWaveBuf && S16x32::kTile <= kWaveBuf && S32x32::kTile <= kWaveBuf,
              "wave tile too small");
static_assert(kWaves * kWaveBuf >= 2 * (kLargeSlab + 256) + 1024, "large-transform scratch");
static_assert(2 * kSpecNB * kSpecPitch <= kWaveBuf, "special tile too small");

struct BlockInfo {
  int coef_off;  // offset of the varblock inside each channel's 65536-coefficient slab
  int px_off;    // y*stride + x of the top-left pixel
  int lf_off;    // by*xblocks + bx of the top-left block
  float sdy;     // inv_global_scale / raw_quant          (group.rs:153)
  float x_cc;    // base_x + ytox / color_factor          (color_correlation_map.rs:76-78)
  float b_cc;
};

struct GroupCtx {
  int group;            // group index
  int bx0, by0;         // group origin in blocks
  int bw, bh;           // group size in blocks (<= 32)
  const int32_t* coef;  // 3 * 65536 slab of this group
};

// entry: bx | by << 5 | off64 << 10 | type << 20
__device__ __forceinline__ uint32_t pack_entry(int bx, int by, int off64, int type) {
  return (uint32_t)bx | ((uint32_t)by << 5) | ((uint32_t)off64 << 10) | ((uint32_t)type << 20);
}

__device__ __forceinline__ void fill_block_info(const FrameDev& f, const GroupCtx& g, uint32_t e, BlockInfo* bi) {
  const int bx = e & 31, by = (e >> 5) & 31, off64 = (e >> 10) & 1023;
  const int gbx = g.bx0 + bx, gby = g.by0 + by;
  bi->coef_off = off64 * 64;
  bi->px_off = (int)((size_t)(gby * 8) * f.plane_stride + (size_t)gbx * 8);
  bi->lf_off = gby * f.xblocks + gbx;
  const int rq = f.raw_quant[bi->lf_off];
  bi->sdy = f.inv_global_scale / (float)(uint32_t)rq;
  const int ci = (gby / kColorTileBlocks) * f.cmap_stride + gbx / kColorTileBlocks;
  bi->x_cc = f.base_x + (float)f.ytox[ci] / f.color_factor;
  bi->b_cc = f.base_b + (float)f.ytob[ci] / f.color_factor;
}

// group.rs:85-96
__device__ __forceinline__ float adjust_quant_bias(int q, float bias_c, float bias3) {
  const float quant = (float)q;
  const float adjusted = quant - bias3 / quant;
  return (q > -2 && q < 2) ? quant * bias_c : adjusted;
}

// Dequantise four consecutive coefficients of channel CH (0 = X, 1 = Y, 2 = B).
// dy: dequantised Y at the same positions (input for X/B, output for Y).
template <int CH>
__device__ __forceinline__ float4 dequant4(const FrameDev& f, const int4 q, const float4 t, const BlockInfo& bi,
                                           float (&dy)[4]) {
  const float bias3 = f.quant_biases[3];
  const float bias = f.quant_biases[CH];
  float sd = bi.sdy;
  if constexpr (CH == 0) sd = bi.sdy * f.x_dm;
  if constexpr (CH == 2) sd = bi.sdy * f.b_dm;
  const int qq[4] = {q.x, q.y, q.z, q.w};
  const float tt[4] = {t.x, t.y, t.z, t.w};
  float r[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const float mul = tt[i] * sd;                              // dequant_lane :114-118
    const float v = adjust_quant_bias(qq[i], bias, bias3) * mul;  // :124-126
    if constexpr (CH == 1) {
      dy[i] = v;
      r[i] = v;
    } else if constexpr (CH == 0) {
      r[i] = __builtin_fmaf(bi.x_cc, dy[i], v);                // :128
    } else {
      r[i] = __builtin_fmaf(bi.b_cc, dy[i], v);                // :129
    }
  }
  return make_float4(r[0], r[1], r[2], r[3]);
}

// One DCT shape (R x C pixels) of transform type TYPE: all batches assigned to this wave.
template <class S>
__device__ void process_dct_class(const FrameDev& f, const GroupCtx& g, int type, const uint32_t* __restrict__ list,
                                  int count, float* __restrict__ buf, BlockInfo* __restrict__ binfo, int wave,
                                  int lane) {
  const int q = quant_table_for_type(type);
  const float* __restrict__ table = f.tables + f.table_offset[q];
  const int tsize = quant_table_size(q);
  const int nbatches = (count + S::NB - 1) / S::NB;
  for (int batch = wave; batch < nbatches; batch += kWaves) {
    const int nb = min(S::NB, count - batch * S::NB);
    if (lane < nb) fill_block_info(f, g, list[batch * S::NB + lane], &binfo[lane]);
    wave_sync();
    float dy[S::E];
    // channel order of the reference: Y, X, B (group.rs:223)
    auto run_channel = [&](auto ch_tag) {
      constexpr int CH = decltype(ch_tag)::value;
      const int32_t* __restrict__ coef = g.coef + CH * kGroupArea;
      const float* __restrict__ tab = table + CH * tsize;
      // ---- stage: coalesced 16-byte loads of raw coefficients + weights -> dequant -> LDS
#pragma unroll
      for (int j = 0; j < S::E / 4; j++) {
        const int fl = (j * 64 + lane) * 4;
        const int b = fl / S::N, k = fl % S::N;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        float d4[4] = {dy[j * 4], dy[j * 4 + 1], dy[j * 4 + 2], dy[j * 4 + 3]};
        if (b < nb) {
          const BlockInfo bi = binfo[b];
          const int4 qv = *reinterpret_cast<const int4*>(coef + bi.coef_off + k);
          const float4 tv = *reinterpret_cast<const float4*>(tab + k);
          v = dequant4<CH>(f, qv, tv, bi, d4);
        }
        if constexpr (CH == 1) {
          dy[j * 4] = d4[0];
          dy[j * 4 + 1] = d4[1];
          dy[j * 4 + 2] = d4[2];
          dy[j * 4 + 3] = d4[3];
        }
        if constexpr (S::kWide) {
          buf[m_addr<S>(b, k)] = v.x;
          buf[m_addr<S>(b, k + 1)] = v.y;
          buf[m_addr<S>(b, k + 2)] = v.z;
          buf[m_addr<S>(b, k + 3)] = v.w;
        } else {
          *reinterpret_cast<float4*>(buf + m_addr<S>(b, k)) = v;
        }
      }
      wave_sync();
      const float* __restrict__ lfp = f.lf[CH];
      float* __restrict__ plane = f.planes[CH];
      const size_t stride = f.plane_stride;
      const int xblocks = f.xblocks;
      idct_batch<S>(
          buf, nb, lane,
          [&](int b, int y, int x) { return lfp[binfo[b].lf_off + y * xblocks + x]; },
          [&](int b, int y, int x, float val) { plane[(size_t)binfo[b].px_off + (size_t)y * stride + x] = val; });
    };
    run_channel(std::integral_constant<int, 1>{});
    run_channel(std::integral_constant<int, 0>{});
    run_channel(std::integral_constant<int, 2>{});
  }
}

// The nine 8x8 special transform types (IDENTITY, DCT2X2, DCT4X4, DCT4X8, DCT8X4, AFV0-3).
__device__ void process_special_class(const FrameDev& f, const GroupCtx& g, int type, const uint32_t* __restrict__ list,
                                      int count, float* __restrict__ buf, BlockInfo* __restrict__ binfo, int wave,
                                      int lane) {
  const int q = quant_table_for_type(type);
  const float* __restrict__ table = f.tables + f.table_offset[q];
  const int tsize = quant_table_size(q);  // 64
  float* __restrict__ tin = buf;
  float* __restrict__ tout = buf + kSpecNB * kSpecPitch;
  const int nbatches = (count + kSpecNB - 1) / kSpecNB;
  for (int batch = wave; batch < nbatches; batch += kWaves) {
    const int nb = min(kSpecNB, count - batch * kSpecNB);
    if (lane < nb) fill_block_info(f, g, list[batch * kSpecNB + lane], &binfo[lane]);
    wave_sync();
    float dy[4 * (kSpecNB * 64 / 256)];
    auto run_channel = [&](auto ch_tag) {
      constexpr int CH = decltype(ch_tag)::value;
      const int32_t* __restrict__ coef = g.coef + CH * kGroupArea;
      const float* __restrict__ tab = table + CH * tsize;
#pragma unroll
      for (int j = 0; j < kSpecNB * 64 / 256; j++) {
        const int fl = (j * 64 + lane) * 4;
        const int b = fl / 64, k = fl % 64;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        float d4[4] = {dy[j * 4], dy[j * 4 + 1], dy[j * 4 + 2], dy[j * 4 + 3]};
        if (b < nb) {
          const BlockInfo bi = binfo[b];
          const int4 qv = *reinterpret_cast<const int4*>(coef + bi.coef_off + k);
          const float4 tv = *reinterpret_cast<const float4*>(tab + k);
          v = dequant4<CH>(f, qv, tv, bi, d4);
        }
        if constexpr (CH == 1) {
          dy[j * 4] = d4[0];
          dy[j * 4 + 1] = d4[1];
          dy[j * 4 + 2] = d4[2];
          dy[j * 4 + 3] = d4[3];
        }
        float* dst = tin + b * kSpecPitch + k;
        dst[0] = v.x;
        dst[1] = v.y;
        dst[2] = v.z;
        dst[3] = v.w;
      }
      wave_sync();
      if (lane < nb) {
        float* c = tin + lane * kSpecPitch;
        c[0] = f.lf[CH][binfo[lane].lf_off];  // transform_buffer[0] = lf[0]
        special_8x8(type, c, tout + lane * kSpecPitch);
      }
      wave_sync();
      float* __restrict__ plane = f.planes[CH];
#pragma unroll
      for (int j = 0; j < kSpecNB * 64 / 256; j++) {
        const int fl = (j * 64 + lane) * 4;
        const int b = fl / 64, p = fl % 64;
        if (b < nb) {
          const float* src = tout + b * kSpecPitch + p;
          float* dst = plane + (size_t)binfo[b].px_off + (size_t)(p / 8) * f.plane_stride + (p % 8);
          *reinterpret_cast<float4*>(dst) = make_float4(src[0], src[1], src[2], src[3]);
        }
      }
      wave_sync();
    };
    run_channel(std::integral_constant<int, 1>{});
    run_channel(std::integral_constant<int, 0>{});
    run_channel(std::integral_constant<int, 2>{});
  }
}

__global__ __launch_bounds__(kThreads) void k1_vardct_group(const FrameDev f, const int group_row0, const int split,
                                                            int* __restrict__ error_flag) {
  __shared__ uint32_t s_list[kGroupBlocks * kGroupBlocks];
  __shared__ int s_count[32], s_base[32], s_cursor[32];
  __shared__ int s_wave_sum[kWaves];
  __shared__ BlockInfo s_binfo[kWaves][kSpecNB];
  __shared__ __attribute__((aligned(16))) float s_buf[kWaves * kWaveBuf];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = blockIdx.x % split;
  const int group = group_row0 * f.xgroups + blockIdx.x / split;
  GroupCtx g;
  g.group = group;
  g.bx0 = (group % f.xgroups) * kGroupBlocks;
  g.by0 = (group / f.xgroups) * kGroupBlocks;
  g.bw = min(kGroupBlocks, f.xblocks - g.bx0);
  g.bh = min(kGroupBlocks, f.yblocks - g.by0);
  g.coef = f.coeffs + (size_t)group * 3 * kGroupArea;

  // ---- 1. transform map -> LDS; per-thread 4 consecutive blocks of the 32x32 raster
  if (tid < 32) {
    s_count[tid] = 0;
    s_cursor[tid] = 0;
  }
  int sizes[4], types[4];
  int local = 0;
  {
    const int by = tid >> 3, bx4 = (tid & 7) * 4;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int bx = bx4 + i;
      uint8_t raw = 0;
      if (bx < g.bw && by < g.bh) raw = f.transform_map[(size_t)(g.by0 + by) * f.xblocks + g.bx0 + bx];
      const int type = raw & 127;
      const bool first = raw >= 128;
      int sz = 0;
      if (first) {
        if (type < JXLH_NUM_TRANSFORMS) {
          sz = covered_x(type) * covered_y(type);
        } else {
          atomicExch(error_flag, JXLH_ERR_INVALID_TRANSFORM);
        }
      }
      sizes[i] = sz;
      types[i] = type;
      local += sz;
    }
  }
  // exclusive scan of `local` over the 256 threads
  int incl = local;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int n = __shfl_up(incl, d, 64);
    if (lane >= d) incl += n;
  }
  if (lane == 63) s_wave_sum[wave] = incl;
  __syncthreads();
  int wave_off = 0;
#pragma unroll
  for (int w = 0; w < kWaves; w++)
    if (w < wave) wave_off += s_wave_sum[w];
  int off64 = wave_off + incl - local;

  // ---- 2. bucket this workgroup's band by type
  const int band = kGroupBlocks / split;
  const int by = tid >> 3;
  const bool mine = (by / band) == sub;
  int offs[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    offs[i] = off64;
    off64 += sizes[i];
    if (mine && sizes[i] > 0) atomicAdd(&s_count[types[i]], 1);
  }
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int t = 0; t < JXLH_NUM_TRANSFORMS; t++) {
      s_base[t] = acc;
      acc += s_count[t];
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; i++) {
    if (mine && sizes[i] > 0) {
      const int slot = s_base[types[i]] + atomicAdd(&s_cursor[types[i]], 1);
      s_list[slot] = pack_entry((tid & 7) * 4 + i, by, offs[i], types[i]);
    }
  }
  __syncthreads();

  // ---- 3. per-type processing
  float* buf = s_buf + wave * kWaveBuf;
  BlockInfo* binfo = s_binfo[wave];
#define JXLH_DCT_CLASS(TYPE, SHAPE)                                                                   \
  if (s_count[TYPE] > 0)                                                                              \
    process_dct_class<SHAPE>(f, g, TYPE, s_list + s_base[TYPE], s_count[TYPE], buf, binfo, wave, lane);
  JXLH_DCT_CLASS(0, S8x8)
  JXLH_DCT_CLASS(4, S16x16)
  JXLH_DCT_CLASS(5, S32x32)
  JXLH_DCT_CLASS(6, S16x8)
  JXLH_DCT_CLASS(7, S8x16)
  JXLH_DCT_CLASS(8, S32x8)
  JXLH_DCT_CLASS(9, S8x32)
  JXLH_DCT_CLASS(10, S32x16)
  JXLH_DCT_CLASS(11, S16x32)
#undef JXLH_DCT_CLASS
  constexpr int kSpecialTypes[9] = {1, 2, 3, 12, 13, 14, 15, 16, 17};
#pragma unroll 1
  for (int i = 0; i < 9; i++) {
    const int t = kSpecialTypes[i];
    if (s_count[t] > 0) process_special_class(f, g, t, s_list + s_base[t], s_count[t], buf, binfo, wave, lane);
  }
  // large classes are workgroup-cooperative (uniform control flow: counts live in LDS)
#pragma unroll 1
  for (int t = 18; t < JXLH_NUM_TRANSFORMS; t++) {
    if (s_count[t] > 0) {
      __syncthreads();
      process_large_class(f, g.bx0, g.by0, g.coef, t, s_list + s_base[t], s_count[t], s_buf, tid);
    }
  }
}

}  // namespace

void launch_vardct_groups(hipStream_t s, const FrameDev& f, int group_row0, int group_row1, int split,
                          int* error_flag) {
  const int ngroups = (group_row1 - group_row0) * f.xgroups;
  if (ngroups <= 0) return;
  hipLaunchKernelGGL(k1_vardct_group, dim3(ngroups * split), dim3(kThreads), 0, s, f, group_row0, split, error_flag);
}

}  // namespace jxlh
