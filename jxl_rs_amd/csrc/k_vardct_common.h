// Shared by the K1 translation units (k_vardct.hip: scan + DCT8..32 classes + special transforms; k_vardct_large.hip:
// the 64..256 transforms): the work-list types k1_scan writes and the per-item decoding.
#pragma once
#include "varblock_core.h"

namespace jxlh {

constexpr int kWaves = 4;
constexpr int kThreads = kWaves * 64;

// class ids of the work lists
enum : int {
  kClsDct8 = 0, kClsDct16x8, kClsDct8x16, kClsDct16x16, kClsDct32x8, kClsDct8x32, kClsDct32x16, kClsDct16x32,
  kClsDct32x32, kClsSpecial, kClsLarge, kNumClasses
};

__host__ __device__ constexpr int class_of_type(int t) {
  constexpr int lut[27] = {kClsDct8,    kClsSpecial, kClsSpecial, kClsSpecial,  kClsDct16x16, kClsDct32x32, kClsDct16x8,
                           kClsDct8x16, kClsDct32x8, kClsDct8x32, kClsDct32x16, kClsDct16x32, kClsSpecial,  kClsSpecial,
                           kClsSpecial, kClsSpecial, kClsSpecial, kClsSpecial,  kClsLarge,    kClsLarge,    kClsLarge,
                           kClsLarge,   kClsLarge,   kClsLarge,   kClsLarge,    kClsLarge,    kClsLarge};
  return lut[t];
}
// The same tables as 4-bit fields of two 64-bit immediates: a lane that looks a table up by a type it has just
// loaded gets shifts instead of a second, dependent memory access (k1_scan's duration is one latency chain).
template <class F>
constexpr uint64_t pack_lut4(F f, int t0) {
  uint64_t v = 0;
  for (int t = t0; t < t0 + 16 && t < JXLH_NUM_TRANSFORMS; t++) v |= (uint64_t)f(t) << (4 * (t - t0));
  return v;
}
constexpr int clog2(int v) { return v <= 1 ? 0 : 1 + clog2(v >> 1); }
constexpr int log2_covered_x(int t) { return clog2(covered_x(t)); }
constexpr int log2_covered_y(int t) { return clog2(covered_y(t)); }
__device__ __forceinline__ int lut4(uint64_t lo, uint64_t hi, int t) {
  return (int)(((t < 16 ? lo : hi) >> (4 * (t & 15))) & 15u);
}
__device__ __forceinline__ int class_of_type_reg(int t) {
  return lut4(pack_lut4(class_of_type, 0), pack_lut4(class_of_type, 16), t);
}
__device__ __forceinline__ int log2_covered_x_reg(int t) {
  return lut4(pack_lut4(log2_covered_x, 0), pack_lut4(log2_covered_x, 16), t);
}
__device__ __forceinline__ int log2_covered_y_reg(int t) {
  return lut4(pack_lut4(log2_covered_y, 0), pack_lut4(log2_covered_y, 16), t);
}
// worst-case number of varblocks of a class per 8x8 block of frame area, as a divisor
__host__ __device__ constexpr int class_min_area(int c) {
  constexpr int lut[kNumClasses] = {1, 2, 2, 4, 4, 4, 8, 8, 16, 1, 32};
  return lut[c];
}

// 16-byte work item: what k1_scan knows about a varblock, raw -- the class kernels derive the dequantisation scale and the
// colour-correlation factors when they decode an item (a few divisions on a handful of lanes per batch), and the
// scan, whose duration is the serial head of K1, writes half the bytes of round 2's 32-byte form.
struct __attribute__((aligned(16))) WorkItem {
  uint32_t packed;  // bx | by << 5 | off64 << 10 | type << 20   (bx, by in blocks inside the group)
  uint32_t group;   // group id (< 2^16: jxlh_frame_begin bounds the frame to 2^31 coefficients = 10922 groups)
                    // | entries of the varblock's B run << 16 when the frame is read in the slot-bucketed form
  int32_t raw_quant;  // HfMetadata::raw_quant_map at the varblock's first block
  uint32_t cc;        // (uint8_t)ytox | (uint8_t)ytob << 8 of the block's colour tile
};
static_assert(sizeof(WorkItem) == 16, "work item layout");

struct BlockInfo {
  int coef_off;  // offset of the varblock inside the frame's coefficient store (channel X)
  // per channel (they differ only in chroma-subsampled frames, K1e / group.rs:223-250, :485-504):
  int px_off[3];  // offset of the top-left pixel in the channel's plane
  int lf_off[3];  // by*xblocks + bx of the channel's first LF sample
  float sdy, x_cc, b_cc;
  int slot_base;  // sparse input: index of the varblock's first slot in sp_slot_start (channel X)
  int first_pos;  // position of the varblock's first coefficient inside the channel slab
  // slot-bucketed entries read in place (FrameDev::se_*): the varblock's entry range per channel, written by k1_scan
  uint32_t e0[3], en[3];
  int cnt_base;   // index of the varblock's first slot count in se_counts (channel X)
};
// Side item of the work lists when the frame is read in the slot-bucketed form (same index as the WorkItem):
// frame-wide index of the varblock's first entry per channel, entries of the X run | Y run << 16 (the B run's count
// rides in WorkItem::group).  16 bits each: a varblock of the DCT classes has at most 16 slots of at most 255 entries
// (positions may repeat: several passes' updates, wide values split into in-range entries).
struct __attribute__((aligned(16))) EntryItem {
  uint32_t e0[3];
  uint32_t nxy;
};
static_assert(sizeof(EntryItem) == 16, "entry item layout");

// every class counter on its own 128-byte line: the 1024 scan workgroups' atomics then meet on nine lines (and L2
// channels) instead of one
constexpr int kCountPitch = 32;
constexpr int kCntFallback0 = kNumClasses + 4;         // + class: batches of the class's fallback list
constexpr int kCntDense0 = kCntFallback0 + kClsSpecial;  // first of the kClsSpecial counters of the dense-route DCT lists
constexpr int kCountLines = kCntDense0 + kClsSpecial;  // the classes, the two-pass slab units, the three fused large
                                              // lists, the entries form's fallback batches (kCntFallback0 ..), the DCT
                                              // classes of the groups that are read from dense slabs (kCntDense0 ..)
constexpr size_t kCountBytes = (size_t)kCountLines * kCountPitch * sizeof(int);
constexpr int kFbAny = 32, kFbAnyPitch = 32;  // summary words of the fallback launch, each on its own 128-byte line
struct WorkLists {
  WorkItem* items[kNumClasses];
  EntryItem* eitems[kClsSpecial];  // the DCT classes only: special / large varblocks read dense slabs
  // Entries form, per-group routing (round 6): the DCT-class varblocks of the groups FrameDev::group_route flags -- groups
  // that arrived as a dense slab, as plain pairs, with a value outside the entries' 10 bits or as an added pass while
  // the rest of the frame is slot-bucketed -- go to these lists, which the dense-slab kernels run (counters at
  // kCntDense0 + class); everything else of the frame keeps reading its entries in place.
  WorkItem* ditems[kClsSpecial];
  uint32_t* fb_any;                 // kFbAny summary words (kFbAnyPitch apart): == fb_epoch = some batch was left
  uint32_t* fallback[kClsSpecial];  // entries form, direct kernels: per class one word per batch; == FrameDev::fb_epoch
                                    // of the launch = left to the fallback launch (varblocks with more entries than a
                                    // lane holds, raw_quant == 0).  Never cleared: see launch_vardct_groups.
  int* counts;  // kCountLines counters at kCountPitch ints, zeroed before k1_scan: the classes, then the large
                // transforms' unit lists (k_vardct_large.hip: two-pass slab units, fused lists of 1 / 2 / 4 slabs)
};

__device__ __forceinline__ void decode_item(const FrameDev& f, const WorkItem& it, BlockInfo* bi) {
  const int bx = it.packed & 31, by = (it.packed >> 5) & 31, off64 = (it.packed >> 10) & 1023;
  const int g = (int)(it.group & 0xffffu);
  const int gbx = (g % f.xgroups) * kGroupBlocks + bx, gby = (g / f.xgroups) * kGroupBlocks + by;
  bi->coef_off = g * 3 * kGroupArea + off64 * 64;  // < 2^31: jxlh_frame_begin bounds the frame
  bi->slot_base = g * 3 * kSlotTable + off64;
  bi->first_pos = off64 * 64;
  bi->cnt_base = g * 3 * kSlotsPerRun + off64;
  if (!f.subsampled) {
    const int px = block_px_offset(f, gbx, gby), lf = gby * f.xblocks + gbx;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      bi->px_off[c] = px;
      bi->lf_off[c] = lf;
    }
  } else {
    // A channel holds only the blocks aligned to its sampling, at the down-sampled position; its LF
    // samples sit in the top-left corner of each LF group's rectangle.  The other blocks are still
    // transformed (their lanes cannot be re-assigned cheaply) and stored into a scrap tile behind the plane.
    const int lfbx = gbx & ~(kLfGroupBlocks - 1), lfby = gby & ~(kLfGroupBlocks - 1);
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const int hs = f.hshift[c], vs = f.vshift[c];
      const bool aligned = ((gbx >> hs) << hs) == gbx && ((gby >> vs) << vs) == gby;
      bi->px_off[c] = aligned ? block_px_offset(f, gbx >> hs, gby >> vs) : f.scrap_off;
      bi->lf_off[c] = (lfby + ((gby - lfby) >> vs)) * f.xblocks + lfbx + ((gbx - lfbx) >> hs);
    }
  }
  bi->sdy = f.inv_global_scale / (float)(uint32_t)it.raw_quant;               // group.rs:153
  bi->x_cc = f.base_x + (float)(int8_t)(it.cc & 0xffu) / f.color_factor;        // color_correlation_map.rs:76-78
  bi->b_cc = f.base_b + (float)(int8_t)((it.cc >> 8) & 0xffu) / f.color_factor;
}

// group.rs:85-96
__device__ __forceinline__ float adjust_quant_bias(int q, float bias_c, float bias3) {
  const float quant = (float)q;
  const float adjusted = quant - bias3 / quant;
  return (q > -2 && q < 2) ? quant * bias_c : adjusted;
}

// Dequantise four consecutive coefficients of channel CH (0 = X, 1 = Y, 2 = B); dy = the
// dequantised Y at the same positions (in for X/B, out for Y).  dequant_lane, group.rs:100-133.
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
    const float mul = tt[i] * sd;
    const float v = adjust_quant_bias(qq[i], bias, bias3) * mul;
    if constexpr (CH == 1) {
      dy[i] = v;
      r[i] = v;
    } else if constexpr (CH == 0) {
      r[i] = __builtin_fmaf(bi.x_cc, dy[i], v);
    } else {
      r[i] = __builtin_fmaf(bi.b_cc, dy[i], v);
    }
  }
  return make_float4(r[0], r[1], r[2], r[3]);
}


// ---- adjust_quant_bias from a table (the division it holds made the dequantisation ~37 vector instructions per
// coefficient, and the transform kernels are bound by instruction issue, not by HBM).  For |q| < kAdjN, tab[ch][|q|]
// holds what the reference computes for +|q| (group.rs:85-96) -- 0 * bias_c, 1 * bias_c, and for |q| >= 2 the device's
// own IEEE (float)|q| - bias3 / (float)|q| -- and the value for -|q| is its negation EXACTLY: IEEE multiplication,
// division and subtraction are sign-symmetric.  The one exception, a table entry that is a zero (then -(+0) is not what
// the reference gets for the negative coefficient), is detected when the table is built and turns the table off.
// Larger magnitudes take the reference's expression behind a wave-uniform branch.
constexpr int kAdjN = 128;
struct AdjTable {
  float v[3][kAdjN];
  int nofast;
};
// every thread of the workgroup calls this once before its first dequantisation; ends with a workgroup barrier
__device__ __forceinline__ void build_adj_table(const FrameDev& f, AdjTable* t, int tid, int nthreads) {
  if (tid == 0) t->nofast = 0;
  __syncthreads();
  for (int i = tid; i < kAdjN; i += nthreads) {
    const float quant = (float)i;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const float v = i < 2 ? quant * f.quant_biases[c] : quant - f.quant_biases[3] / quant;
      t->v[c][i] = v;
      if (i >= 2 && v == 0.0f) t->nofast = 1;
    }
  }
  __syncthreads();
}
// dequant_lane (group.rs:100-133) for four coefficients of channel CH: same operations as dequant4, the adjusted
// value from the table
template <int CH>
__device__ __forceinline__ float4 dequant4t(const FrameDev& f, const int4 q, const float4 t, const BlockInfo& bi,
                                            const AdjTable* __restrict__ at, float (&dy)[4]) {
  const float* __restrict__ tab = at->v[CH];
  const bool nofast = at->nofast != 0;
  float sd = bi.sdy;
  if constexpr (CH == 0) sd = bi.sdy * f.x_dm;
  if constexpr (CH == 2) sd = bi.sdy * f.b_dm;
  const int qq[4] = {q.x, q.y, q.z, q.w};
  const float tt[4] = {t.x, t.y, t.z, t.w};
  int aq[4];
  float am[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    aq[i] = qq[i] < 0 ? -qq[i] : qq[i];
    am[i] = tab[min(aq[i], kAdjN - 1)];
  }
  // kAdjN is a power of two: the OR of the four magnitudes is below it iff each is
  if (__builtin_expect(nofast || __any(((uint32_t)aq[0] | (uint32_t)aq[1] | (uint32_t)aq[2] | (uint32_t)aq[3]) >= (uint32_t)kAdjN), 0)) {
#pragma unroll
    for (int i = 0; i < 4; i++)
      if (nofast || aq[i] >= kAdjN)
        am[i] = __uint_as_float(__float_as_uint(adjust_quant_bias(qq[i], f.quant_biases[CH], f.quant_biases[3])) ^
                                ((uint32_t)qq[i] & 0x80000000u));
  }
  float r[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const float adj = __uint_as_float(__float_as_uint(am[i]) ^ ((uint32_t)qq[i] & 0x80000000u));
    const float mul = tt[i] * sd;
    const float v = adj * mul;
    if constexpr (CH == 1) {
      dy[i] = v;
      r[i] = v;
    } else if constexpr (CH == 0) {
      r[i] = __builtin_fmaf(bi.x_cc, dy[i], v);
    } else {
      r[i] = __builtin_fmaf(bi.b_cc, dy[i], v);
    }
  }
  return make_float4(r[0], r[1], r[2], r[3]);
}

}  // namespace jxlh
