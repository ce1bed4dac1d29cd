// K1 -- dequantisation + chroma-from-luma + LLF-from-LF + variable-size IDCT for a frame.
//
// Replaces the `if let Some(pixels)` branch of decode_vardct_group
// (jxl/src/frame/group.rs:579-611): dequant_block (:137-177), dequant_lane (:100-133),
// adjust_quant_bias (:85-96), the LF patch copy (:227-235), transform_to_pixels and
// the copy into the group planes (:237-250).
//
// Varblock sizes and positions are data dependent, so the work is first *binned*:
//
//   k1_scan   one 256-thread workgroup per 256x256 group: loads the group's 32x32 transform
//             map, prefix-scans the varblock sizes in raster order -> coefficient offset of
//             every varblock (the reference lays varblocks back to back in decode order,
//             group.rs:440, :612), evaluates the per-varblock scalars (inv_global_scale /
//             raw_quant, the two chroma-from-luma multipliers) and appends one 32-byte work
//             item per varblock to a frame-wide list per transform class.
//   k1_dct*   one kernel per class family, sized for that family's registers/LDS: wavefronts
//             stride over batches of same-shape varblocks; a batch's dequantised coefficients
//             are staged channel by channel in the wave's private LDS tile with 16-byte
//             coalesced loads and run through the wave-level cores of varblock_core.h.  For
//             the small shapes all three channels' loads are issued up front and the dequant
//             weights stay in registers across batches.
//   k1_special / k1_large  the 8x8 special transforms and the 64..256 transforms.
//
// HBM traffic is the compulsory 12 B/px in + 12 B/px out (+ maps, + 32 B per varblock of
// work items); arithmetic is f32 in the reference's operation order (bit-exact vs the FMA
// build of the oracle).  Output is independent of the order in which work items land in
// the lists (each varblock is independent).
#include <algorithm>
#include <atomic>

#include "k_vardct_common.h"

// Cache policy of K1's two streams (round 3, profiles/r03_n_k1_policy.txt): the coefficient slabs are read exactly once
// and the pixels are next touched by another kernel after 0.8 GB of other traffic -- `nt` on both keeps them from
// displacing each other in L2: K1 0.448 -> 0.430 ms, pipelined step 0.824 -> 0.806 ms (alternating runs, one box).
// (`nt` on the filter kernel's loads / stores measured 10-40 % SLOWER: its halo rows are re-read by the neighbour tile.)
#ifndef JXLH_NT_COEF
#define JXLH_NT_COEF true
#endif
#ifndef JXLH_NT_K1_STORE
#define JXLH_NT_K1_STORE true
#endif
namespace jxlh {
namespace {

// ------------------------------------------------------------------------------------------
// One workgroup scans kScanGroups groups (one 256-thread quarter each).  Its duration is the serial head of K1 and is
// one latency chain -- map bytes -> two block scans -> a returning global atomic per class -> item stores -- so (round 3)
// every global input is requested up front, the type tables are register immediates, two classes share a 32-bit scan
// word, and the quarters pool their counts: 256 workgroups' atomics meet on a class counter instead of 1024 (same-
// address atomics serialise at ~12 ns each: 13 of the former 36 us).
constexpr int kScanGroups = 4;
constexpr int kScanThreads = kScanGroups * kThreads;
// STRIP (the frame runs through k123_strip, k_strip.hip): the scan also decides per 64x64 tile who reconstructs it -- the
// strip kernel, iff every varblock touching the tile lies inside one 32x32 quadrant of it and is a DCT with sides <= 32
// (what aligning every varblock to its own size gives) -- writes a descriptor
// per block of those tiles ({type | dx << 5 | dy << 7 | off64 << 9 | 1 << 31, raw_quant of the varblock}: everything the
// strip kernel needs to find a block's varblock and its coefficients) and appends work items for the OTHER tiles only.
// ENT (the frame is read in the slot-bucketed form, FrameDev::se_*): the scan also turns the group's slot counts into
// the entry range of every varblock -- an exclusive prefix sum over the 1024 slots of each channel (a thread loads the
// counts of four slots as one word, the three channels ride one 64-bit block scan, 17 bits each), kept in LDS and read
// at the varblock's first slot -- and writes it beside the work item: the class kernels then go from the item straight
// to the entries.  This replaces the unpack pass of round 4 (pair words + 4-byte slot tables: 69 + 12.6 MB written and
// read again per 8K frame).
// ROUTE (ENT only): some groups of the frame are read from their dense slabs (FrameDev::group_route): their DCT-class
// varblocks go to WorkLists::ditems.  A frame without routed groups runs the instantiation without any of it.
template <bool STRIP, bool ENT = false, bool ROUTE = false>
__global__ __launch_bounds__(kScanThreads) void k1_scan(const FrameDev f, const WorkLists wl, const int group_row0,
                                                         int* __restrict__ error_flag,
                                                         const int* __restrict__ group_list, const int ngroups,
                                                         int* __restrict__ next_counts) {
  constexpr int kPairs = (kNumClasses + 1) / 2;
  __shared__ int s_wave_sum[kScanGroups][kWaves];
  __shared__ uint32_t s_wcls[kScanGroups][kWaves][kPairs];
  __shared__ int s_count[kScanGroups][kNumClasses], s_base[kScanGroups][kNumClasses];
  __shared__ int s_route[ROUTE ? kScanGroups : 1];  // != 0 = the quarter's group is read from its dense slab (FrameDev::group_route)
  __shared__ int s_tmode[kScanGroups][16];  // STRIP: != 0 = a tile of the group (4 x 4 of them) the class kernels keep
  // ENT: exclusive prefix of the slot counts, three channels packed (17 bits each; a run holds at most 65536 entries)
  __shared__ uint64_t s_pref[ENT ? kScanGroups : 1][ENT ? kSlotsPerRun + 1 : 1];
  __shared__ uint64_t s_wpref[ENT ? kScanGroups : 1][kWaves];
  if constexpr (STRIP) {
    if (threadIdx.x < kScanGroups * 16) s_tmode[threadIdx.x / 16][threadIdx.x % 16] = 0;
    // progress flags + ticket counter of the strip kernel that follows
    if (blockIdx.x == 0)
      for (int i = threadIdx.x; i < f.strip_nflags; i += kScanThreads) f.strip_flags[i] = 0;
    __syncthreads();
  }
  // the counters of the NEXT launch (the other set: its last readers finished before this kernel started)
  if (blockIdx.x == 0 && threadIdx.x < kCountLines) next_counts[threadIdx.x * kCountPitch] = 0;
  const int sub = threadIdx.x / kThreads, tid = threadIdx.x % kThreads, lane = tid & 63, wave = tid >> 6;
  const int gi = blockIdx.x * kScanGroups + sub;
  const bool live = gi < ngroups;  // a dead quarter walks through the barriers with an empty group
  const int group = !live ? 0 : group_list ? group_list[gi] : group_row0 * f.xgroups + gi;
  const int bx0 = (group % f.xgroups) * kGroupBlocks, by0 = (group / f.xgroups) * kGroupBlocks;
  const int bw = live ? min(kGroupBlocks, f.xblocks - bx0) : 0, bh = live ? min(kGroupBlocks, f.yblocks - by0) : 0;
  // 4 consecutive blocks of the 32x32 raster per thread
  int sizes[4], types[4], rq4[4], local = 0;
  const int by = tid >> 3, bx4 = (tid & 7) * 4;
  const int gby = min(by0 + by, f.yblocks - 1);
  uint8_t raw4[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const size_t at = (size_t)gby * f.xblocks + min(bx0 + bx4 + i, f.xblocks - 1);
    raw4[i] = f.transform_map[at];
    rq4[i] = f.raw_quant[at];
  }
  // the four blocks of a thread share one colour tile (8 blocks wide, bx4 % 4 == 0)
  const int ci = (gby / kColorTileBlocks) * f.cmap_stride + min(bx0 + bx4, f.xblocks - 1) / kColorTileBlocks;
  const uint32_t cc = (uint32_t)(uint8_t)f.ytox[ci] | (uint32_t)(uint8_t)f.ytob[ci] << 8;
  // ENT: counts of slots 4 tid .. 4 tid + 3 of the three channels, {first entry, entries} of the three runs
  uint32_t cw[3] = {0u, 0u, 0u};
  uint2 run[3] = {make_uint2(0u, 0u), make_uint2(0u, 0u), make_uint2(0u, 0u)};
  bool dense_route = false;
  if constexpr (ENT) {
    if constexpr (ROUTE) {
      dense_route = live && f.group_route[group] != 0;
      if (tid == 0) s_route[sub] = dense_route ? 1 : 0;  // (published by the barriers below)
    }
    if (live && !dense_route) {
#pragma unroll
      for (int c = 0; c < 3; c++) {
        cw[c] = *reinterpret_cast<const uint32_t*>(f.se_counts + ((size_t)group * 3 + c) * kSlotsPerRun + 4 * tid);
        run[c] = f.se_runs[group * 3 + c];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int bx = bx4 + i;
    const uint8_t raw = (bx < bw && by < bh) ? raw4[i] : 0;
    const int type = raw & 127;
    int sz = 0;
    if (raw >= 128) {  // first (top-left) block of a varblock, group.rs:468-473
      if (type < JXLH_NUM_TRANSFORMS) {
        const int cx = 1 << log2_covered_x_reg(type), cy = 1 << log2_covered_y_reg(type);
        sz = cx * cy;
        // Error::InvalidBlockSizeForChromaSubsampling (frame/modular/mod.rs:1058-1060)
        if (f.subsampled && sz > 1) atomicExch(error_flag, JXLH_ERR_INVALID_BLOCK_SIZE);
        // Error::HFBlockOutOfBounds (frame/modular/mod.rs:1061-1064): the varblock must end inside its group
        // and inside the frame.  Such an item is dropped: its pixel stores would leave the plane.
        if (bx + cx > bw || by + cy > bh) {
          atomicExch(error_flag, JXLH_ERR_BLOCK_OUT_OF_BOUNDS);
          sz = 0;
        }
        if constexpr (STRIP) {
          // tiles this varblock keeps away from the strip kernel: all it touches if it is not a small DCT or leaves
          // its tile; every tile of the group if the map is broken
          const bool closed = (bx & 3) + cx <= 4 && (by & 3) + cy <= 4 && class_of_type_reg(type) < kClsSpecial;
          if (sz == 0 || f.subsampled) {
            for (int k = 0; k < 16; k++) atomicOr(&s_tmode[sub][k], 1);
          } else if (!closed) {
            for (int ty = by >> 3; ty <= (by + cy - 1) >> 3; ty++)
              for (int tx = bx >> 3; tx <= (bx + cx - 1) >> 3; tx++) atomicOr(&s_tmode[sub][ty * 4 + tx], 1);
          }
        }
      } else {
        atomicExch(error_flag, JXLH_ERR_INVALID_TRANSFORM);  // Error::InvalidVarDCTTransform
        if constexpr (STRIP)
          for (int k = 0; k < 16; k++) atomicOr(&s_tmode[sub][k], 1);
      }
    }
    sizes[i] = sz;
    types[i] = type;
    local += sz;
  }
  // coefficient offsets (in 64-coefficient slots): prefix sum of the covered areas in raster order (group.rs:612)
  int incl = local;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int n = __shfl_up(incl, d, 64);
    if (lane >= d) incl += n;
  }
  if (lane == 63) s_wave_sum[sub][wave] = incl;
  uint64_t pmine = 0, pincl = 0;
  if constexpr (ENT) {
    auto bytes = [](uint32_t w) { return (uint64_t)((w & 0xffu) + ((w >> 8) & 0xffu) + ((w >> 16) & 0xffu) + (w >> 24)); };
    pmine = bytes(cw[0]) | bytes(cw[1]) << 17 | bytes(cw[2]) << 34;
    pincl = pmine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)pincl, d, 64), hi = (uint32_t)__shfl_up((int)(uint32_t)(pincl >> 32), d, 64);
      if (lane >= d) pincl += (uint64_t)lo | (uint64_t)hi << 32;
    }
    if (lane == 63) s_wpref[sub][wave] = pincl;
  }
  // per-class rank of every varblock in raster order (stable: neighbouring blocks stay neighbours in the lists ->
  // contiguous coefficient reads, full-line pixel writes).  Two classes share a 32-bit word, 16 bits each (a group
  // holds at most 1024 varblocks).
  int slot[4], cls4[4];
  bool strip_tile = false;  // the thread's four blocks share a tile
  if constexpr (STRIP) {
    __syncthreads();
    strip_tile = s_tmode[sub][(by >> 3) * 4 + (bx4 >> 3)] == 0;
  }
  uint32_t mine[kPairs], excl[kPairs];
#pragma unroll
  for (int w = 0; w < kPairs; w++) mine[w] = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int cls = sizes[i] > 0 && !strip_tile ? class_of_type_reg(types[i]) : -1;
    cls4[i] = cls;
    slot[i] = 0;
    const int sh = (cls & 1) * 16;
#pragma unroll
    for (int w = 0; w < kPairs; w++) {
      if ((cls >> 1) == w) {  // cls = -1 matches no pair
        slot[i] = (int)((mine[w] >> sh) & 0xffffu);
        mine[w] += 1u << sh;
      }
    }
  }
#pragma unroll
  for (int w = 0; w < kPairs; w++) {
    uint32_t inc = mine[w];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t n = (uint32_t)__shfl_up((int)inc, d, 64);
      if (lane >= d) inc += n;
    }
    excl[w] = inc - mine[w];
    if (lane == 63) s_wcls[sub][wave][w] = inc;
  }
  __syncthreads();
  if constexpr (ENT) {
    uint64_t base = pincl - pmine;
#pragma unroll
    for (int v = 0; v < kWaves; v++)
      if (v < wave) base += s_wpref[sub][v];
    uint64_t* P = s_pref[sub] + 4 * tid;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      P[k] = base;
      base += (uint64_t)((cw[0] >> (8 * k)) & 0xffu) | (uint64_t)((cw[1] >> (8 * k)) & 0xffu) << 17 |
              (uint64_t)((cw[2] >> (8 * k)) & 0xffu) << 34;
    }
    if (tid == kThreads - 1) P[4] = base;  // the run's total (entry kSlotsPerRun)
  }
  int off64 = incl - local;
  int group_total = 0;
#pragma unroll
  for (int v = 0; v < kWaves; v++) {
    if (v < wave) off64 += s_wave_sum[sub][v];
    group_total += s_wave_sum[sub][v];
  }
  // first-block flags that claim more blocks than the group holds (overlapping varblocks; the reference cannot
  // produce such a map, a caller-built one can): the 10-bit coefficient offset of the work items would overflow,
  // K1 would read past the group's slab and the class lists (sized by area) could overflow.  Nothing of such a
  // group is reconstructed.
  const bool bad_group = group_total > bw * bh;
  if (bad_group && tid == 0) atomicExch(error_flag, JXLH_ERR_BLOCK_OUT_OF_BOUNDS);
#pragma unroll
  for (int w = 0; w < kPairs; w++) {
    uint32_t woff = 0, total = 0;
#pragma unroll
    for (int v = 0; v < kWaves; v++) {
      if (v < wave) woff += s_wcls[sub][v][w];
      total += s_wcls[sub][v][w];
    }
    excl[w] += woff;
    if ((tid >> 1) == w && tid < kNumClasses)
      s_count[sub][tid] = bad_group ? 0 : (int)((total >> ((tid & 1) * 16)) & 0xffffu);
  }
  __syncthreads();
  // one atomic per class for the whole workgroup; the quarters take consecutive ranges in group order.  ENT: the DCT
  // classes of a dense-route quarter count into the dense lists (threads 32 .. 32 + kClsSpecial) instead
  if (threadIdx.x < kNumClasses || (ROUTE && threadIdx.x >= 32 && threadIdx.x < 32 + kClsSpecial)) {
    const bool dlist = ROUTE && threadIdx.x >= 32;
    const int c = dlist ? threadIdx.x - 32 : threadIdx.x;
    auto mine_q = [&](int q) {
      if constexpr (ROUTE) return c >= kClsSpecial || (s_route[q] != 0) == dlist;
      else return true;
    };
    int total = 0;
#pragma unroll
    for (int q = 0; q < kScanGroups; q++) total += mine_q(q) ? s_count[q][c] : 0;
    int base = total > 0 ? atomicAdd(&wl.counts[(dlist ? kCntDense0 + c : c) * kCountPitch], total) : 0;
#pragma unroll
    for (int q = 0; q < kScanGroups; q++) {
      if (mine_q(q)) {
        s_base[q][c] = base;
        base += s_count[q][c];
      }
    }
  }
  // (a dense-route group's slab is already there: nothing to expand for its special / large varblocks)
  if (live && tid == 0 && f.group_dense)
    f.group_dense[group] = !dense_route && (s_count[sub][kClsSpecial] | s_count[sub][kClsLarge]) != 0;
  if constexpr (STRIP) {
    if (live && tid < 16) {
      const int gtx = (group % f.xgroups) * 4 + (tid & 3), gty = (group / f.xgroups) * 4 + (tid >> 2);
      if (gtx < f.strips && gty < f.tile_rows) f.strip_mode[gty * f.strips + gtx] = (bad_group || s_tmode[sub][tid]) ? 1 : 0;
    }
  }
  __syncthreads();
  if (bad_group) return;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    if (sizes[i] > 0) {
      WorkItem it;
      it.packed = (uint32_t)(bx4 + i) | ((uint32_t)by << 5) | ((uint32_t)off64 << 10) | ((uint32_t)types[i] << 20);
      it.group = (uint32_t)group;
      it.raw_quant = rq4[i];
      it.cc = cc;
      const int cls = cls4[i];
      if (cls >= 0) {
        const int sh = (cls & 1) * 16;
        int rank = slot[i];
#pragma unroll
        for (int w = 0; w < kPairs; w++)
          if ((cls >> 1) == w) rank += (int)((excl[w] >> sh) & 0xffffu);
        bool to_dense = false;
        if constexpr (ENT) {
          if (cls < kClsSpecial) {
            if (dense_route) {
              to_dense = true;
              wl.ditems[cls][s_base[sub][cls] + rank] = it;
            } else {
              // (the prefixes were published by the barriers above.)  A count table that claims more than its run holds
              // is cut at the run's end: no entry outside the run is ever read.  A DCT-class varblock has at most 16
              // slots of at most 255 entries: the counts fit their 16 bits
              const uint64_t p0 = s_pref[sub][off64], p1 = s_pref[sub][min(off64 + sizes[i], kSlotsPerRun)];
              EntryItem ei;
              uint32_t n[3];
#pragma unroll
              for (int c = 0; c < 3; c++) {
                const uint32_t a0 = min((uint32_t)(p0 >> (17 * c)) & 0x1ffffu, run[c].y);
                const uint32_t a1 = min((uint32_t)(p1 >> (17 * c)) & 0x1ffffu, run[c].y);
                ei.e0[c] = run[c].x + a0;
                n[c] = min(a1 - a0, 0xffffu);
              }
              ei.nxy = n[0] | n[1] << 16;
              it.group |= n[2] << 16;
              wl.eitems[cls][s_base[sub][cls] + rank] = ei;
            }
          }
        }
        if (!to_dense) wl.items[cls][s_base[sub][cls] + rank] = it;
      }
      if constexpr (STRIP) {
        if (strip_tile) {
          const int cx = 1 << log2_covered_x_reg(types[i]), cy = 1 << log2_covered_y_reg(types[i]);
          const uint32_t d0 = (uint32_t)types[i] | ((uint32_t)off64 << 9) | (1u << 31);
          for (int dy = 0; dy < cy; dy++)
            for (int dx = 0; dx < cx; dx++)
              f.strip_desc[(size_t)(by0 + by + dy) * f.xblocks + bx0 + bx4 + i + dx] =
                  make_uint2(d0 | ((uint32_t)dx << 5) | ((uint32_t)dy << 7), (uint32_t)rq4[i]);
        }
      }
    }
    off64 += sizes[i];
  }
}

template <class S>
__device__ __forceinline__ void stage4(float* __restrict__ buf, int b, int k, float4 v) {
  if constexpr (S::kWide) {
    buf[m_addr<S>(b, k)] = v.x;
    buf[m_addr<S>(b, k + 1)] = v.y;
    buf[m_addr<S>(b, k + 2)] = v.z;
    buf[m_addr<S>(b, k + 3)] = v.w;
  } else {
    *reinterpret_cast<float4*>(buf + m_addr<S>(b, k)) = v;
  }
}

// All batches of one DCT shape assigned to this wave.  PREFETCH: issue the coefficient
// loads of all three channels before touching any (small shapes; 3*E/4 int4 in flight per lane).
// Sparse input: builds the integer coefficients of one channel of the batch in the wave's tile
// (zero, then ds_add of the varblocks' pairs: duplicates from several passes add up before
// dequantisation, like `coeffs[i] += v` in the dense slab).  64 / NB lanes per varblock walk its
// pair range; the layout is the M layout, so the dequantisation pass converts in place.
// The pair ranges of all three channels (sparse_ranges) and the first kSparsePrefetch pairs of each
// (sparse_first) are fetched when the batch starts, so the per-channel step only waits on LDS.
// (Software-pipelining this across batches -- next batch's items / ranges / pairs requested while
// the current one is transformed -- was measured slower: 216 vs 190 us for the 8x8 class at 8K.)
constexpr int kSparsePrefetch = 2;
struct SparseLane {
  uint32_t i0[3], i1[3];
  uint32_t first[3][kSparsePrefetch];
  int first_pos;
};

template <class S>
__device__ __forceinline__ void sparse_ranges(const FrameDev& f, const BlockInfo* __restrict__ binfo, int nb, int lane,
                                              SparseLane& sl) {
  constexpr int LPB = 64 / S::NB;
  const int b = lane / LPB, j = lane % LPB;
  const bool on = b < nb;
  const int base = on ? binfo[b].slot_base : 0;
  sl.first_pos = on ? binfo[b].first_pos : 0;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    sl.i0[c] = on ? f.sp_slot_start[base + c * kSlotTable] + j : 0u;
    sl.i1[c] = on ? f.sp_slot_start[base + c * kSlotTable + S::N / 64] : 0u;
  }
}
template <class S>
__device__ __forceinline__ void sparse_first(const FrameDev& f, SparseLane& sl) {
  constexpr int LPB = 64 / S::NB;
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int k = 0; k < kSparsePrefetch; k++) {
      const uint32_t i = sl.i0[c] + k * LPB;
      sl.first[c][k] = i < sl.i1[c] ? f.sp_sorted[i] : 0u;  // pos 0 + value 0 adds nothing
    }
}

template <class S>
__device__ __forceinline__ void sparse_stage_channel(const FrameDev& f, int ch, float* __restrict__ buf, int lane,
                                                     const SparseLane& sl) {
  int* ibuf = reinterpret_cast<int*>(buf);
  constexpr int kWords = S::NB * S::SM;
  if constexpr (kWords % 4 == 0) {
    for (int i = lane * 4; i < kWords; i += 256) *reinterpret_cast<int4*>(ibuf + i) = make_int4(0, 0, 0, 0);
  } else {
    for (int i = lane; i < kWords; i += 64) ibuf[i] = 0;
  }
  wave_sync();
  constexpr int LPB = 64 / S::NB;
  const int b = lane / LPB;
  auto add = [&](uint32_t p) {
    atomicAdd(&ibuf[m_addr<S>(b, (int)(p & 0xffffu) - sl.first_pos)], (int)(int16_t)(p >> 16));
  };
#pragma unroll
  for (int k = 0; k < kSparsePrefetch; k++)
    if (sl.i0[ch] + k * LPB < sl.i1[ch]) add(sl.first[ch][k]);
  for (uint32_t i = sl.i0[ch] + kSparsePrefetch * LPB; i < sl.i1[ch]; i += LPB) add(f.sp_sorted[i]);
  wave_sync();
}

// ---- the slot-bucketed entries read in place (SPARSE == 2).  The varblock's entry range per channel comes with the
// work item (k1_scan<ENT>), so the first entries of all three channels are requested one memory round trip after the
// item -- the pair form needs two (slot table, then pairs).  The LPB lanes of a varblock share its range evenly
// (entry r of the range goes to lane r % LPB); an entry carries only its position inside its 64-coefficient slot, so
// for varblocks of more than one slot the lane finds the slot of entry r in the exclusive prefix of the varblock's
// slot counts (one byte load per slot and channel, a segmented shuffle scan, 10 bits per channel in one LDS word per
// slot): log2(slots) LDS reads per entry.
// entries a lane holds per channel before it has to go back to memory: a varblock with at most D * LPB entries per
// channel is staged without a second round trip (and may take the direct path below)
// (a d1-like varblock has an entry at ~11 % of its Y positions, half / two thirds of that in X / B:
// profiles/r05_c_entry_counts.txt)
template <class S>
constexpr int ent_depth() {
  constexpr int per_lane = S::E;                    // coefficients per lane and channel: 8, 16 or 32
  constexpr int lpb = 64 / S::NB;
  return per_lane <= 8 ? 3 : per_lane <= 16 ? (lpb <= 8 ? 4 : 3) : (lpb <= 8 ? 6 : lpb <= 16 ? 5 : 4);
}
template <int D>
struct EntLane {
  uint32_t i0[3], i1[3];  // this lane's first entry / the end of the varblock's range, per channel (frame-wide indices)
  uint32_t e[3][D];       // the lane's first D entries of each channel
};
// EX: the word an exclusive slot-count prefix of the three channels is packed in.  uint32_t, 10 bits per channel: enough
// for the varblocks the direct path takes (at most D entries per lane); uint64_t, 21 bits per channel: any count a
// varblock can legally have (16 slots x 255 entries -- repeated positions of several passes, wide values split into
// in-range entries), what the dense dequantisation pass (mode 2, the fallback of the direct kernels) is built with.
template <class EX>
constexpr int excl_bits() { return sizeof(EX) == 8 ? 21 : 10; }
template <class S, int D, class EX>
__device__ __forceinline__ void entries_begin(const FrameDev& f, const BlockInfo* __restrict__ binfo, EX* __restrict__ s_excl,
                                              int nb, int lane, EntLane<D>& sl) {
  constexpr int LPB = 64 / S::NB, NS = S::N / 64;
  static_assert(LPB >= NS, "one lane per slot for the count scan");
  const int b = lane / LPB, j = lane % LPB;
  const bool on = b < nb;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    sl.i0[c] = on ? binfo[b].e0[c] + j : 0u;
    sl.i1[c] = on ? binfo[b].e0[c] + binfo[b].en[c] : 0u;
  }
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int k = 0; k < D; k++) {
      const uint32_t i = sl.i0[c] + k * LPB;
      sl.e[c][k] = i < sl.i1[c] ? (uint32_t)f.se_entries[i] : 0u;
    }
  if constexpr (NS > 1) {
    constexpr int B = excl_bits<EX>();
    EX packed = 0;
    if (on && j < NS) {
      const uint8_t* cp = f.se_counts + binfo[b].cnt_base + j;
      packed = (EX)cp[0] | (EX)cp[kSlotsPerRun] << B | (EX)cp[2 * kSlotsPerRun] << (2 * B);
    }
    auto shfl_up = [](EX v, int d) {
      if constexpr (sizeof(EX) == 8) {
        const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)v, d, LPB), hi = (uint32_t)__shfl_up((int)(uint32_t)(v >> 32), d, LPB);
        return (EX)((uint64_t)lo | (uint64_t)hi << 32);
      } else {
        return (EX)__shfl_up((int)v, d, LPB);
      }
    };
    // exclusive scan over the block's lanes (segments of LPB lanes): shift by one slot, then an inclusive scan
    EX x = shfl_up(packed, 1);
    if (j == 0) x = 0;
#pragma unroll
    for (int d = 1; d < NS; d <<= 1) {
      const EX y = shfl_up(x, d);
      if (j >= d) x += y;
    }
    if (j < NS) s_excl[b * NS + j] = x;  // b < NB always: the tile holds NB * NS words
  }
}
// position (in the varblock's stored order) of entry e, the r-th of the varblock's range of channel ch
template <class S, class EX>
__device__ __forceinline__ int entry_pos(const EX* __restrict__ s_excl, int b, int ch, uint32_t e, uint32_t r) {
  constexpr int NS = S::N / 64, B = excl_bits<EX>();
  int sidx = 0;
  if constexpr (NS > 1) {
#pragma unroll
    for (int step = NS / 2; step >= 1; step >>= 1) {
      const uint32_t v = (uint32_t)(s_excl[b * NS + sidx + step] >> (B * ch)) & ((1u << B) - 1u);
      if (r >= v) sidx += step;
    }
  }
  return sidx * 64 + (int)(e & 63u);
}
template <class S>
__device__ __forceinline__ void zero_tile(int* __restrict__ ibuf, int lane) {
  constexpr int kWords = S::NB * S::SM;
  if constexpr (kWords % 4 == 0) {
    for (int i = lane * 4; i < kWords; i += 256) *reinterpret_cast<int4*>(ibuf + i) = make_int4(0, 0, 0, 0);
  } else {
    for (int i = lane; i < kWords; i += 64) ibuf[i] = 0;
  }
}

// HELD = false (the inline fallback of the direct 8x8 kernel): the lane's first D entries are NOT taken from sl -- every
// entry is requested again (L2 hits), the range comes from the batch's BlockInfo -- so that nothing of sl stays alive
// through the transforms: with it the inline body raised the direct kernel from 76 to 98 VGPRs (6 -> 4 waves per SIMD).
template <class S, int D, class EX, bool HELD = true>
__device__ __forceinline__ void entries_stage_channel(const FrameDev& f, int ch, float* __restrict__ buf,
                                                      const EX* __restrict__ s_excl, int lane, const EntLane<D>& sl,
                                                      const BlockInfo* __restrict__ binfo = nullptr, int nb = 0) {
  int* ibuf = reinterpret_cast<int*>(buf);
  zero_tile<S>(ibuf, lane);
  wave_sync();
  constexpr int LPB = 64 / S::NB;
  const int b = lane / LPB, j = lane % LPB;
  auto add = [&](uint32_t e, uint32_t r) {  // r: index of the entry inside the varblock's range
    atomicAdd(&ibuf[m_addr<S>(b, entry_pos<S, EX>(s_excl, b, ch, e, r))], (int)(e << 16) >> 22);
  };
  constexpr int DH = HELD ? D : 0;
  uint32_t i0 = 0, i1 = 0;
  if constexpr (HELD) {
    i0 = sl.i0[ch];
    i1 = sl.i1[ch];
  } else if (b < nb) {
    i0 = binfo[b].e0[ch] + j;
    i1 = binfo[b].e0[ch] + binfo[b].en[ch];
  }
#pragma unroll
  for (int k = 0; k < DH; k++)
    if (i0 + k * LPB < i1) add(sl.e[ch][k], (uint32_t)(j + k * LPB));
  // what lies beyond the D entries a lane holds (content denser than d1, split wide values): four requests in flight per
  // lane and round -- one dependent load per entry made the dense pass 3x the time of the dense-slab kernels at four
  // times d1's density (round 6, profiles/r06_c_density.txt)
  constexpr int U = 4;
  uint32_t r = (uint32_t)(j + DH * LPB), i = i0 + DH * LPB;
  while (__any(i < i1)) {
    uint32_t ev[U];
#pragma unroll
    for (int u = 0; u < U; u++) ev[u] = i + u * LPB < i1 ? (uint32_t)f.se_entries[i + u * LPB] : 0u;
#pragma unroll
    for (int u = 0; u < U; u++)
      if (i + u * LPB < i1) add(ev[u], r + u * LPB);
    i += U * LPB;
    r += U * LPB;
  }
  wave_sync();
}

// ---- the direct path of the entries form: only the coefficients that HAVE an entry are dequantised; every other
// position of the tile is +0.0f, which is what the reference computes for a zero coefficient under the conditions
// FrameDev::se_direct_ok / tables_ok / AdjTable::nofast establish (group.rs:85-133: (0.0 * bias_c) * mul = +0.0,
// fma(cc, +0.0, +0.0) = +0.0; a non-zero coefficient never dequantises to a zero).  At d1 about one coefficient in ten
// has an entry: the dense pass (tile read, ~10 vector instructions and an LDS table lookup per POSITION, tile write)
// becomes ~15 instructions per ENTRY.  Duplicate positions (several passes' updates in one list) still add up as
// integers first:
//   a  ds_add of the entry's value into the zeroed tile              (integer sums, like `coeffs[i] += v`)
//   b  ds exchange of the word with a sentinel: the first lane to arrive gets the sum and owns the position
//   c  the owner writes the dequantised float over the sentinel
// (LDS operations of a wavefront execute in program order: every lane's step a is done before any step b, every b
// before any c.)  The sentinel INT32_MIN cannot be a sum: a run holds at most 65536 entries of 10 bits.  X and B then
// add the chroma-from-luma term at the positions Y owns, fma(cc, dy, v) in the reference's form; where Y has no entry
// the reference's fma(cc, +0.0, v) returns v itself (v is never -0.0).
template <int D>
struct EntDirect {
  uint32_t ad[3][D];  // position of the entry's coefficient in the varblock | value << 16; kNoEntry = no k-th entry
};
constexpr uint32_t kNoEntry = 0xffffffffu;
constexpr int kClaimed = (int)0x80000000;

template <class S, int CH, int D>
__device__ __forceinline__ void direct_stage_channel(const FrameDev& f, float* __restrict__ buf, int lane,
                                                     const EntDirect<D>& ed, const float* __restrict__ table, int tsize,
                                                     float sdy, float cc, const AdjTable* __restrict__ adj,
                                                     float (&dyw)[D], uint32_t& ywin) {
  int* ibuf = reinterpret_cast<int*>(buf);
  constexpr int LPB = 64 / S::NB;
  const int b = lane / LPB;
  // the dequant weights of this channel's entries: requested first, needed in step c
  float w[D];
#pragma unroll
  for (int k = 0; k < D; k++) w[k] = ed.ad[CH][k] != kNoEntry ? table[CH * tsize + (int)(ed.ad[CH][k] & 0xffffu)] : 0.0f;
  zero_tile<S>(ibuf, lane);
  wave_sync();
  int at[D];
#pragma unroll
  for (int k = 0; k < D; k++) {
    at[k] = m_addr<S>(b, (int)(ed.ad[CH][k] & 0x3ffu));
    if (ed.ad[CH][k] != kNoEntry) atomicAdd(&ibuf[at[k]], (int)ed.ad[CH][k] >> 16);
  }
  wave_sync();
  int sum[D];
#pragma unroll
  for (int k = 0; k < D; k++) {
    sum[k] = kClaimed;
    if (ed.ad[CH][k] != kNoEntry) sum[k] = atomicExch(&ibuf[at[k]], kClaimed);
  }
  wave_sync();
  float sd = sdy;
  if constexpr (CH == 0) sd = sdy * f.x_dm;
  if constexpr (CH == 2) sd = sdy * f.b_dm;
#pragma unroll
  for (int k = 0; k < D; k++) {
    if (sum[k] != kClaimed) {
      // dequant_lane (group.rs:100-133) for one coefficient, the operations of dequant4t
      const int q = sum[k], aq = q < 0 ? -q : q;
      float am = adj->v[CH][min(aq, kAdjN - 1)];
      if (aq >= kAdjN)
        am = __uint_as_float(__float_as_uint(adjust_quant_bias(q, f.quant_biases[CH], f.quant_biases[3])) ^ ((uint32_t)q & 0x80000000u));
      const float a = __uint_as_float(__float_as_uint(am) ^ ((uint32_t)q & 0x80000000u));
      const float mul = w[k] * sd;
      const float v = a * mul;
      if constexpr (CH == 1) {
        dyw[k] = v;
        ywin |= 1u << k;
      }
      buf[at[k]] = v;
    }
  }
  if constexpr (CH != 1) {
    wave_sync();
#pragma unroll
    for (int k = 0; k < D; k++)
      if ((ywin >> k) & 1u) {
        float* t = buf + m_addr<S>(b, (int)(ed.ad[1][k] & 0x3ffu));
        *t = __builtin_fmaf(cc, dyw[k], *t);
      }
  }
  wave_sync();
}

template <class S>
__device__ __forceinline__ int4 tile_q4(const float* __restrict__ buf, int b, int k) {
  const int* ibuf = reinterpret_cast<const int*>(buf);
  if constexpr (S::kWide) {
    return make_int4(ibuf[m_addr<S>(b, k)], ibuf[m_addr<S>(b, k + 1)], ibuf[m_addr<S>(b, k + 2)],
                     ibuf[m_addr<S>(b, k + 3)]);
  } else {
    return *reinterpret_cast<const int4*>(ibuf + m_addr<S>(b, k));
  }
}

// gwave: this wave's index among the nwaves of the grid, rotated by the caller so that the classes one
// kernel handles start on different waves (batch b of a class goes to wave (rotation + b) % nwaves);
// without it every class would pile its batches on the low-numbered waves.  Returns the number of
// batches of the class (the next class's rotation).
// SUB (chroma-subsampled frames, S8x8 only): a channel holds a block only if the block is aligned to the channel's
// sampling (decode_item marks the others with px_off == scrap_off); their coefficients are never decoded (zeros
// in the reference's slab, frame/group.rs:521-524), so the loads are skipped and read as zero, and nothing is stored.
// SPARSE: 0 = dense slabs, 1 = bucketed pair words + slot tables (sp_sorted), 2 = slot-bucketed entries in place (se_*)
// with the dense dequantisation pass, 3 = the same input, direct path only: batches it cannot take are flagged in
// WorkLists::fallback[class] (one word per batch, written with the launch's epoch: no atomics, nothing to clear) and the
// fallback launch (LISTED) runs the flagged ones through mode 2.  CLS: the class id.
// INLINE_FB (mode 3): a batch the direct path cannot take runs through the dense dequantisation pass right here instead
// of going to the fallback list -- for the 8x8 class, whose generic body costs a handful of registers: a frame denser
// than d1 content then degrades batch by batch inside one launch (round 6).
// s_dy (mode 2, shapes with 32 coefficients per lane; nullable): the dequantised Y of the batch waits in LDS for the X / B
// channels' chroma-from-luma instead of in 32 registers through two 32-point IDCTs -- the kernels that take it run two
// workgroups per CU and have the room (S::E * 64 floats per wavefront).
template <class S, bool PREFETCH, int SPARSE, bool SUB = false, int CLS = 0, bool INLINE_FB = false, bool LISTED = false,
          class EX = uint32_t>
__device__ __forceinline__ int run_dct_class(const FrameDev& f, const WorkItem* __restrict__ items,
                                             const EntryItem* __restrict__ eitems, int count, int type,
                                             float* __restrict__ buf, BlockInfo* __restrict__ binfo_base,
                                             EX* __restrict__ s_excl, float* __restrict__ s_lf, int gwave,
                                             int nwaves, int lane, const AdjTable* __restrict__ adj,
                                             uint32_t* __restrict__ wl_fallback = nullptr,
                                             int* __restrict__ fallback_count = nullptr, float* __restrict__ s_dy = nullptr,
                                             uint32_t fb_first_flag = 0, uint32_t* __restrict__ wl_fb_any = nullptr,
                                             int* fb_rank = nullptr) {
  constexpr int NCH = S::E / 4;  // 16-byte chunks per lane per channel
  const int q = quant_table_for_type(type);
  const float* __restrict__ table = f.tables + f.table_offset[q];
  const int tsize = quant_table_size(q);
  const int nbatches = (count + S::NB - 1) / S::NB;
  // The batches this wave runs.  Plain: gwave, gwave + nwaves, ...  LISTED (the fallback launch): the batches whose word
  // in wl_fallback holds this launch's epoch -- a workgroup takes chunks of kFbChunk consecutive batches (one flag per
  // lane, a ballot), its kWaves waves share a chunk's flagged batches round robin; the caller rotates the chunk -> workgroup
  // map from class to class (gwave), so the classes' chunks spread over the whole grid, and hands over the flag word of
  // the workgroup's first FOUR chunks (fb_first_flag: the kernel requests those of all classes at once when it starts, so a
  // workgroup with nothing to do leaves after one memory round trip and one with work does not pay a round trip per
  // class).  (Round 6's first form appended batch ids to a list: one returning atomic per wave and class on one counter,
  // 131 000 of them on a frame that leaves every batch -- 0.27 ms for a launch with nothing else to do.)
  constexpr int kFbChunk = 16;  // (64-batch chunks leave most of the grid idle on dense frames: K1 0.42 -> 0.57 ms at x2)
  struct BatchIter {
    int bi;                    // plain: the batch; listed: the chunk
    unsigned long long mask;   // listed: flagged batches of the chunk not yet handed out
    bool pre;                  // listed: mask = the kernel's prefetch: the workgroup's first FOUR chunks, 16 bits each
  };
  const int it_wave = gwave % kWaves, it_wg = gwave / kWaves, it_nwg = nwaves / kWaves;  // (LISTED: gwave is not rotated)
  auto iter_next = [&](BatchIter& st) -> int {  // the next batch of this wave, -1 when done
    if constexpr (!LISTED) {
      const int b = st.bi;
      st.bi += nwaves;
      return b < nbatches ? b : -1;
    } else {
      for (;;) {
        while (st.mask) {
          const int bit = __builtin_ctzll(st.mask);
          st.mask &= st.mask - 1;
          const int chunk = st.pre ? st.bi + (bit >> 4) * it_nwg : st.bi;
          // (*fb_rank: flagged batches the WORKGROUP has met so far in this launch, over all classes -- its waves see
          // the same flags and count alike -- so the few batches a sparse frame leaves never queue up behind each other
          // on one wave while the other three idle: the launch then lasts one batch's latency, not two)
          if ((*fb_rank)++ % kWaves == it_wave) return chunk * kFbChunk + (bit & 15);
        }
        st.bi += st.pre ? 4 * it_nwg : it_nwg;
        st.pre = false;
        if (st.bi * kFbChunk >= nbatches) return -1;
        const int idx = st.bi * kFbChunk + lane;
        st.mask = __ballot(lane < kFbChunk && idx < nbatches && wl_fallback[idx] == (uint32_t)f.fb_epoch);
      }
    }
  };
  static_assert(kFbChunk == 16, "the prefetched mask holds four 16-batch chunks");
  BatchIter iter = {LISTED ? it_wg : gwave, LISTED ? __ballot(fb_first_flag == (uint32_t)f.fb_epoch && fb_first_flag != 0) : 0ull, LISTED};
  // the weights a lane needs do not depend on the batch
  constexpr bool kPF = PREFETCH && SPARSE != 3;  // (the inline fallback of mode 3 reads its weights per batch)
  float4 tw[kPF ? 3 : 1][NCH];
  if constexpr (kPF) {
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
      for (int j = 0; j < NCH; j++)
        tw[c][j] = *reinterpret_cast<const float4*>(table + c * tsize + ((j * 64 + lane) * 4) % S::N);
  }
  // The entries form is bound by the chain of dependent memory round trips of a batch (item -> entries, then the LF
  // samples of each channel right before its transform: five per batch, 72 % of a wavefront's life parked on them,
  // profiles/r05_b_sparse_pairs_pmc.txt), not by bytes.  kChain: the NEXT batch's items are requested while this one is
  // transformed, and the LF samples of all three channels are requested together with the entries (spread over the
  // lanes, handed over through LDS): one exposed round trip per batch.
  constexpr bool kChain = SPARSE >= 2;
#ifndef JXLH_ITEM_PREFETCH
#define JXLH_ITEM_PREFETCH 1  // 0 none, 1 the 8..16-point classes (register-light bodies), 2 every class
#endif
  constexpr bool kNextItem = kChain && !LISTED && (JXLH_ITEM_PREFETCH == 2 || (JXLH_ITEM_PREFETCH == 1 && PREFETCH));
  constexpr int kLfPerBlock = 3 * (S::R / 8) * (S::C / 8), kLfIters = (S::NB * kLfPerBlock + 63) / 64;
  WorkItem it_next = {};
  EntryItem ei_next = {};
  if constexpr (kNextItem) {
    if (gwave < nbatches && lane < min(S::NB, count - gwave * S::NB)) {
      it_next = items[gwave * S::NB + lane];
      ei_next = eitems[gwave * S::NB + lane];
    }
  }
  int n_dense_pass = 0;  // (wave-uniform) batches this wave ran through the dense pass: inline (mode 3) or listed
  bool flagged_any = false;
  for (int batch = iter_next(iter); batch >= 0; batch = iter_next(iter)) {
    const int nb = min(S::NB, count - batch * S::NB);
    if constexpr (LISTED) n_dense_pass++;
    BlockInfo* __restrict__ binfo = binfo_base;
    bool direct_ok = true;  // (wave-uniform)
    if constexpr (kChain) {
      WorkItem it = it_next;
      EntryItem ei = ei_next;
      if constexpr (kNextItem) {
        const int nxt = batch + nwaves;
        if (nxt < nbatches && lane < min(S::NB, count - nxt * S::NB)) {
          it_next = items[nxt * S::NB + lane];
          ei_next = eitems[nxt * S::NB + lane];
        }
      } else if (lane < nb) {
        it = items[batch * S::NB + lane];
        ei = eitems[batch * S::NB + lane];
      }
      bool item_ok = true;
      if (lane < nb) {
        decode_item(f, it, &binfo[lane]);
#pragma unroll
        for (int c = 0; c < 3; c++) binfo[lane].e0[c] = ei.e0[c];
        binfo[lane].en[0] = ei.nxy & 0xffffu;
        binfo[lane].en[1] = ei.nxy >> 16;
        binfo[lane].en[2] = it.group >> 16;
        if constexpr (SPARSE == 3) {
          // the direct path takes a varblock whose raw_quant the division survives and that has no more entries per
          // channel than its lanes hold
          const float sdy = binfo[lane].sdy;
          item_ok = sdy > 0.0f && sdy < __builtin_inff() &&
                    max(max(ei.nxy & 0xffffu, ei.nxy >> 16), it.group >> 16) <= (uint32_t)(ent_depth<S>() * (64 / S::NB));
        }
      }
      if constexpr (SPARSE == 3 && !INLINE_FB) {
        // left to the fallback launch BEFORE anything of the batch is requested: its word gets the launch's epoch
        if (!__all(item_ok)) {
          if (lane == 0) {
            wl_fallback[batch] = (uint32_t)f.fb_epoch;
            // ... and one of the launch's kFbAny summary words (own cache lines, picked by the wave): plain stores of
            // the same value, a wave's first reject only
            if (!flagged_any) wl_fb_any[(gwave % kFbAny) * kFbAnyPitch] = (uint32_t)f.fb_epoch;
          }
          flagged_any = true;
          continue;
        }
      } else {
        direct_ok = __all(item_ok);
      }
    } else if (lane < nb) {
      const WorkItem it = items[batch * S::NB + lane];
      decode_item(f, it, &binfo[lane]);
    }
    wave_sync();
    SparseLane sl;
    constexpr int D = ent_depth<S>();
    EntLane<D> el;
    if constexpr (SPARSE == 1) {
      sparse_ranges<S>(f, binfo, nb, lane, sl);
      sparse_first<S>(f, sl);
    } else if constexpr (SPARSE >= 2) {
      entries_begin<S, D, EX>(f, binfo, s_excl, nb, lane, el);
    }
    if constexpr (kChain) {
      // LF sample idx = (block, channel, y, x) of the batch: requested now, in LDS before the first transform
      float lfv[kLfIters];
#pragma unroll
      for (int i = 0; i < kLfIters; i++) {
        const int idx = i * 64 + lane, b = idx / kLfPerBlock, rem = idx % kLfPerBlock;
        const int c = rem / ((S::R / 8) * (S::C / 8)), yx = rem % ((S::R / 8) * (S::C / 8));
        const int y = yx / (S::C / 8), x = yx % (S::C / 8);
        lfv[i] = 0.0f;
        if (b < nb) {
          const float* __restrict__ lp = c == 0 ? f.lf[0] : c == 1 ? f.lf[1] : f.lf[2];
          lfv[i] = lp[binfo[b].lf_off[c] + y * f.xblocks + x];
        }
      }
#pragma unroll
      for (int i = 0; i < kLfIters; i++)
        if (i * 64 + lane < S::NB * kLfPerBlock) s_lf[i * 64 + lane] = lfv[i];
      // (published by the wave_sync that follows the tile's zero fill in entries_stage_channel)
    }
    auto transform_channel = [&](auto ch_tag) {
      constexpr int CH = decltype(ch_tag)::value;
      const float* __restrict__ lfp = f.lf[CH];
      float* __restrict__ plane = f.planes[CH];
      const PixLayout lay = pix_layout(f);
      const int xblocks = f.xblocks;
      idct_batch<S>(
          buf, nb, lane,
          [&](int b, int y, int x) {
            if constexpr (kChain) return s_lf[b * kLfPerBlock + (CH * (S::R / 8) + y) * (S::C / 8) + x];
            else return lfp[binfo[b].lf_off[CH] + y * xblocks + x];
          },
          [&](int b, int x, int yb, const float(&v)[8]) {
            if (SUB && binfo[b].px_off[CH] == f.scrap_off) return;
            float* dst = plane + binfo[b].px_off[CH] + lay.xoff(x) + yb * lay.ystep_blk;
            if (lay.tiled) {  // rows 0-3 and rows 4-7 of the lane's column: two 16-byte stores, 128 bytes apart
              gstore_f4<JXLH_NT_K1_STORE>(dst, make_float4(v[0], v[1], v[2], v[3]));
              gstore_f4<JXLH_NT_K1_STORE>(dst + 32, make_float4(v[4], v[5], v[6], v[7]));
            } else {
#pragma unroll
              for (int i = 0; i < 8; i++) dst[i * lay.ystep8] = v[i];
            }
          });
    };
    using TagX = std::integral_constant<int, 0>;
    using TagY = std::integral_constant<int, 1>;
    using TagB = std::integral_constant<int, 2>;
    // the dense dequantisation pass: every coefficient position of the batch (GM: 0 dense slabs, 1 pair words, 2 entries)
    auto generic_batch = [&](auto mode_tag) {
      constexpr int GM = decltype(mode_tag)::value;
      int4 qv[kPF ? 3 : 1][NCH];
      if constexpr (kPF && !GM) {
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
          for (int j = 0; j < NCH; j++) {
            const int fl = (j * 64 + lane) * 4;
            const int b = fl / S::N, k = fl % S::N;
            qv[c][j] = make_int4(0, 0, 0, 0);
            if (b < nb && (!SUB || binfo[b].px_off[c] != f.scrap_off))
              qv[c][j] = gload_i4<JXLH_NT_COEF>(f.coeffs + binfo[b].coef_off + c * kGroupArea + k);
          }
      }
      // The dequantised Y the X / B channels' chroma-from-luma needs: kept in registers across the channels -- except for
      // the shape with 32 of them per lane (32x32, dense input), where they would sit through two 32-point IDCTs
      // (167 VGPRs + 16 spilled + 68 bytes of scratch for that class alone): there X and B dequantise their Y values again
      // (the coefficient read hits L2, the table-driven dequantisation is ~10 instructions per value).
      constexpr bool kRecomputeY = !GM && !PREFETCH && S::E > 16;
      constexpr bool kDyLds = GM == 2 && SPARSE == 2 && S::E > 16;  // (callers of mode 2 pass s_dy)
      float dy[kRecomputeY || kDyLds ? 4 : S::E];
      auto stage_channel = [&](auto ch_tag) {
        constexpr int CH = decltype(ch_tag)::value;
        if constexpr (GM == 1) sparse_stage_channel<S>(f, CH, buf, lane, sl);
        if constexpr (GM == 2) entries_stage_channel<S, D, EX, SPARSE == 2>(f, CH, buf, s_excl, lane, el, binfo, nb);
#pragma unroll
        for (int j = 0; j < NCH; j++) {
          const int fl = (j * 64 + lane) * 4;
          const int b = fl / S::N, k = fl % S::N;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          float d4[4];
          if constexpr (kRecomputeY) {
            d4[0] = d4[1] = d4[2] = d4[3] = 0.0f;
          } else if constexpr (kDyLds) {
            d4[0] = d4[1] = d4[2] = d4[3] = 0.0f;
            if constexpr (CH != 1) {
              const float4 t = *reinterpret_cast<const float4*>(s_dy + (j * 64 + lane) * 4);
              d4[0] = t.x;
              d4[1] = t.y;
              d4[2] = t.z;
              d4[3] = t.w;
            }
          } else {
            d4[0] = dy[j * 4];
            d4[1] = dy[j * 4 + 1];
            d4[2] = dy[j * 4 + 2];
            d4[3] = dy[j * 4 + 3];
          }
          if (b < nb) {
            const BlockInfo bi = binfo[b];
            int4 qq;
            float4 tt;
            if constexpr (GM != 0) {
              qq = tile_q4<S>(buf, b, k);  // converted in place: this lane alone touches (b, k..k+3)
              if constexpr (kPF) tt = tw[CH][j];
              else tt = *reinterpret_cast<const float4*>(table + CH * tsize + k);
            } else if constexpr (kPF) {
              qq = qv[CH][j];
              tt = tw[CH][j];
            } else {
              qq = gload_i4<JXLH_NT_COEF>(f.coeffs + bi.coef_off + CH * kGroupArea + k);
              tt = *reinterpret_cast<const float4*>(table + CH * tsize + k);
              if constexpr (kRecomputeY && CH != 1) {
                const int4 qy = gload_i4<false>(f.coeffs + bi.coef_off + kGroupArea + k);
                const float4 ty = *reinterpret_cast<const float4*>(table + tsize + k);
                (void)dequant4t<1>(f, qy, ty, bi, adj, d4);
              }
            }
            v = dequant4t<CH>(f, qq, tt, bi, adj, d4);
          }
          if constexpr (CH == 1 && kDyLds) {
            *reinterpret_cast<float4*>(s_dy + (j * 64 + lane) * 4) = make_float4(d4[0], d4[1], d4[2], d4[3]);
          } else if constexpr (CH == 1 && !kRecomputeY) {
            dy[j * 4] = d4[0];
            dy[j * 4 + 1] = d4[1];
            dy[j * 4 + 2] = d4[2];
            dy[j * 4 + 3] = d4[3];
          }
          stage4<S>(buf, b, k, v);
        }
        wave_sync();
      };
      // channel order of the reference: Y, X, B (group.rs:223)
      stage_channel(TagY{});
      transform_channel(TagY{});
      stage_channel(TagX{});
      transform_channel(TagX{});
      stage_channel(TagB{});
      transform_channel(TagB{});
    };
    if constexpr (SPARSE == 3) {
      // direct path: every varblock of the batch has a raw_quant the division survives and no more entries per channel
      // than its lanes hold -- otherwise the batch is left to the dense pass (k1_entries_fallback)
      constexpr int LPB = 64 / S::NB;
      const int b = lane / LPB, j = lane % LPB;
      float sdy = 0.0f, xcc = 0.0f, bcc = 0.0f;
      if (b < nb) {
        sdy = binfo[b].sdy;
        xcc = binfo[b].x_cc;
        bcc = binfo[b].b_cc;
      }
      if constexpr (INLINE_FB) {
        if (!direct_ok) {
          static_assert(S::N == 64, "the inline fallback has no slot prefixes: one-slot varblocks only");
          generic_batch(std::integral_constant<int, 2>{});
          n_dense_pass++;
          wave_sync();  // binfo / s_lf are rewritten by the next batch
          continue;
        }
      }
      wave_sync();  // the slot prefixes (entries_begin) are in LDS
      EntDirect<D> ed;
#pragma unroll
      for (int c = 0; c < 3; c++)
#pragma unroll
        for (int k = 0; k < D; k++) {
          ed.ad[c][k] = kNoEntry;
          if (el.i0[c] + k * LPB < el.i1[c]) {
            const uint32_t e = el.e[c][k];
            ed.ad[c][k] = (uint32_t)entry_pos<S, EX>(s_excl, b, c, e, (uint32_t)(j + k * LPB)) | (uint32_t)((int)(e << 16) >> 22) << 16;
          }
        }
      float dyw[D];
      uint32_t ywin = 0;
#pragma unroll
      for (int k = 0; k < D; k++) dyw[k] = 0.0f;
      direct_stage_channel<S, 1, D>(f, buf, lane, ed, table, tsize, sdy, 0.0f, adj, dyw, ywin);
      transform_channel(TagY{});
      direct_stage_channel<S, 0, D>(f, buf, lane, ed, table, tsize, sdy, xcc, adj, dyw, ywin);
      transform_channel(TagX{});
      direct_stage_channel<S, 2, D>(f, buf, lane, ed, table, tsize, sdy, bcc, adj, dyw, ywin);
      transform_channel(TagB{});
    } else {
      generic_batch(std::integral_constant<int, SPARSE>{});
    }
  }
  // statistics (jxlh_frame_k1_counters; only while kernel timing is on: 16 000 waves' atomics on one counter cost 0.2 ms)
  if (f.k1_stats && (LISTED || INLINE_FB) && n_dense_pass && lane == 0) atomicAdd(fallback_count, n_dense_pass);
  return nbatches;
}

// wave index rotated by the batches the previous classes of the kernel occupy
__device__ __forceinline__ int rotate_wave(int gw, int used, int nw) {
  const int r = (gw - used % nw) % nw;
  return r < 0 ? r + nw : r;
}

template <class S>
struct ShapeTag {
  using type = S;
};
using S8x8 = Shape<8, 8>;
using S16x16 = Shape<16, 16>;
using S32x32 = Shape<32, 32>;
using S16x8 = Shape<16, 8>;
using S8x16 = Shape<8, 16>;
using S32x8 = Shape<32, 8, 4>;  // tall 32x8 at NB = 8 would need a 3136-word tile
using S8x32 = Shape<8, 32>;
using S32x16 = Shape<32, 16>;
using S16x32 = Shape<16, 32>;

constexpr int kTileA = S8x8::kTile;                                               // 832 words
constexpr int kTileC = cmax(cmax(cmax(S32x8::kTile, S8x32::kTile), cmax(S32x16::kTile, S16x32::kTile)),
                            S32x32::kTile);                                       // 2624

// family A: DCT 8x8 -- the dominant transform
// INLINE (mode 3): batches beyond the direct path's depth take the dense dequantisation pass inside this launch (one-slot
// varblocks need no slot prefixes) instead of going to the fallback launch.  The host picks the form per frame from the
// density of its entries (FrameDev::se_dense_hint): the inline body costs registers (79 -> 98 VGPRs, 6 -> 4 waves per
// SIMD: +12 us on a d1 frame), the fallback launch a second pass over the work items (0.84 against 0.68 ms for K1 at
// four times d1's density; profiles/r06_c_density.txt).
// (the direct form without the inline body sits one register above six waves per SIMD: the bound makes the compiler fit)
template <int SPARSE, bool SUB = false, bool INLINE = false>
#ifndef JXLH_K1_INLINE_WPE
#define JXLH_K1_INLINE_WPE 1  // waves per SIMD the inline form is compiled for (1 = whatever its 96 VGPRs allow: 5)
#endif
__global__ __launch_bounds__(kThreads, SPARSE == 3 ? (INLINE ? JXLH_K1_INLINE_WPE : 6) : 1) void k1_dct8(const FrameDev f, const WorkLists wl) {
  __shared__ __attribute__((aligned(16))) float s_buf[kWaves * kTileA];
  __shared__ BlockInfo s_binfo[kWaves][S8x8::NB];
  __shared__ AdjTable s_adj;
  build_adj_table(f, &s_adj, threadIdx.x, kThreads);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ float s_lf[SPARSE >= 2 ? kWaves : 1][S8x8::NB * 3];
  // (mode 3: batches beyond the direct path's depth take the dense dequantisation pass inline -- one-slot varblocks need
  // no slot prefixes, and the generic body fits the kernel's registers)
  static_assert(!INLINE || SPARSE == 3, "the inline fallback belongs to the direct form");
  run_dct_class<S8x8, true, SPARSE, SUB, kClsDct8, INLINE>(f, wl.items[kClsDct8], wl.eitems[kClsDct8], wl.counts[(kClsDct8) * kCountPitch], 0,
                                                   s_buf + wave * kTileA, s_binfo[wave], (uint32_t*)nullptr, s_lf[SPARSE >= 2 ? wave : 0],
                                                   blockIdx.x * kWaves + wave, gridDim.x * kWaves, lane, &s_adj,
                                                   wl.fallback[kClsDct8], wl.counts + (kCntFallback0 + kClsDct8) * kCountPitch, nullptr, 0u,
                                                   wl.fb_any);
}

// families B (16x8, 8x16, 16x16) + C (everything with a 32-point side) in ONE launch (round 3): as two kernels both
// ran at three waves per SIMD (145 / 168 VGPRs), so merging them costs no occupancy and removes a kernel boundary -- one
// fill / drain less in K1's serial sequence -- while the eight class lists spread over one grid (rotated start waves).
// Occupancy is not what limits these classes' batches anyway: at two waves per SIMD the 32-point family runs 24 %
// slower, the 16-point one 8 %, and DCT8 is flat between 3 and 5 (profiles/r03_n_k1.txt); requesting the LF samples
// of a batch ahead of the coefficients (through an LDS scratch) measured flat as well
// exclusive slot-count prefixes of a batch's varblocks (entries form): NB * (N / 64) words, at most 40 (8 x 32)
constexpr int kExclWords = 40;
#ifndef JXLH_K1_MERGED
// Dense slabs, frames of >= 512 groups: every DCT class in ONE launch (k1_dct16_32<0, false, true>).  Built in round 6,
// -1.7 % on K1's own time with one frame in flight on one box -- and 4-5 % SLOWER on the pipelined headline (two frames
// in flight: 0.718 against 0.691 ms per frame; three boxes' worth of alternating runs): one long launch at three waves
// per SIMD leaves the other frame's kernels nothing to overlap with.  Off; the one-launch form still runs the
// dense-route lists of a routed frame (a handful of groups).  profiles/r06_f_k1_merged.txt
#define JXLH_K1_MERGED 0
#endif
#ifndef JXLH_K1_DIRECT_WPE
#define JXLH_K1_DIRECT_WPE 3  // waves per SIMD the direct form of k1_dct16_32 is compiled for
#endif
// mode 2 (the dense dequantisation pass of the entries form: frames whose parameters rule the direct path out) is
// compiled for two workgroups per CU: its 32-point bodies then hold everything in registers + the LDS stash of the
// dequantised Y (at three they spilled 250-330 bytes per lane)
constexpr int kDyWords = 32 * 64;  // S::E * 64 for the shapes with a 32-point side
// FB (mode 2 only): the fallback launch of the direct form -- per class, the batches flagged in WorkLists::fallback[class]
// (what the direct kernels could not take: more entries than their lanes hold, raw_quant == 0) instead of the class's
// whole list.  Usually a handful of batches (a workgroup reads a few words of flags per class and leaves); on content
// denser than d1 it is the main route of the 16..32-point classes.  (Round 5's fallback kernel dispatched one mixed list through a switch over the nine
// bodies: 203 spilled VGPRs, 792 bytes of scratch per lane.)
template <int SPARSE, bool FB = false, bool ALL = false>
__global__ __launch_bounds__(kThreads, SPARSE == 3 ? JXLH_K1_DIRECT_WPE : SPARSE == 2 ? 2 : 3) void k1_dct16_32(const FrameDev f, const WorkLists wl) {
  static_assert(!FB || SPARSE == 2, "the fallback launch runs the dense dequantisation pass of the entries form");
  static_assert(!ALL || SPARSE == 0, "one launch for every DCT class: dense slabs only");
  // FB: the flag word of this lane in the workgroup's first four chunks of every class, requested together before anything
  // else of the workgroup is set up (order = the order the classes run in below; a class's chunk c belongs to workgroup
  // (c + chunks of the classes before it) % grid).  Nothing flagged there and no later chunk: the workgroup leaves.
  uint32_t fb_pre[kClsSpecial] = {};
  uint32_t fb_live = 0;  // FB: classes (bit = position in the FB order) with a flagged batch or chunks beyond the prefetch
  if constexpr (FB) {
    // first level: the launch's summary words (one hot load; the direct kernels set one on a wave's first reject).  A
    // frame that left nothing -- the usual one -- costs this launch ~4 us instead of the ~25 us of the flag prefetch.
    // (the class counters are requested in the same round trip)
    const uint32_t summary = wl.fb_any[(threadIdx.x & (kFbAny - 1)) * kFbAnyPitch];
    constexpr int kOrder[kClsSpecial] = {kClsDct32x32, kClsDct32x16, kClsDct16x32, kClsDct32x8, kClsDct8x32,
                                         kClsDct16x16, kClsDct16x8,  kClsDct8x16,  kClsDct8};
    constexpr int kNb[kClsSpecial] = {S32x32::NB, S32x16::NB, S16x32::NB, S32x8::NB, S8x32::NB, S16x16::NB, S16x8::NB, S8x16::NB, S8x8::NB};
    const int grid = (int)gridDim.x, l = threadIdx.x & 63;
    int nbat[kClsSpecial];
#pragma unroll
    for (int k = 0; k < kClsSpecial; k++) nbat[k] = (wl.counts[kOrder[k] * kCountPitch] + kNb[k] - 1) / kNb[k];
    if (!__any(summary == (uint32_t)f.fb_epoch)) return;
    int used = 0;
#pragma unroll
    for (int k = 0; k < kClsSpecial; k++) {
      // (chunks of 16 batches: lanes 16 r .. 16 r + 15 take the workgroup's chunk of round r, fc + r * grid)
      const int nch = (nbat[k] + 15) / 16, fc = rotate_wave((int)blockIdx.x, used, grid), idx = (fc + (l >> 4) * grid) * 16 + (l & 15);
      fb_pre[k] = idx < nbat[k] ? wl.fallback[kOrder[k]][idx] : 0u;
      if (fc + 4 * grid < nch) fb_live |= 1u << k;
      used += nch;
    }
#pragma unroll
    for (int k = 0; k < kClsSpecial; k++)
      if (__any(fb_pre[k] == (uint32_t)f.fb_epoch)) fb_live |= 1u << k;
    if (!fb_live) return;  // (every wave of the workgroup reads the same words: uniform)
  }
  __shared__ __attribute__((aligned(16))) float s_buf[kWaves * kTileC];
  __shared__ BlockInfo s_binfo[kWaves][8];
  using EX = std::conditional_t<SPARSE == 2, uint64_t, uint32_t>;  // mode 2 takes any entry count (excl_bits)
  __shared__ EX s_excl[SPARSE >= 2 ? kWaves : 1][kExclWords];
  __shared__ AdjTable s_adj;
  build_adj_table(f, &s_adj, threadIdx.x, kThreads);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* buf = s_buf + wave * kTileC;
  EX* ex = s_excl[SPARSE >= 2 ? wave : 0];
  __shared__ float s_lfs[SPARSE >= 2 ? kWaves : 1][96];  // LF samples of a batch: NB * 3 * (R / 8) * (C / 8) <= 96
  float* lfs = s_lfs[SPARSE >= 2 ? wave : 0];
  __shared__ __attribute__((aligned(16))) float s_dyb[SPARSE == 2 ? kWaves * kDyWords : 4];
  float* sdy = SPARSE == 2 ? s_dyb + wave * kDyWords : nullptr;
  const int gw = blockIdx.x * kWaves + wave, nw = gridDim.x * kWaves;
  auto cnt = [&](int cls) { return wl.counts[cls * kCountPitch]; };
  // the long batches (32-point sides) first: the tail of the launch is then made of the short ones
  int used = 0, fb_rank = 0;
  auto run = [&](auto shape_tag, auto pf_tag, auto cls_tag, int type, int fb_k = 0) {  // fb_k: index in the FB order
    using S = typename decltype(shape_tag)::type;
    constexpr bool PF = decltype(pf_tag)::value;
    constexpr int CLS = decltype(cls_tag)::value;
    if constexpr (FB) {
      // nothing of the class for this workgroup: not even the class's preamble (a sparse frame's few batches otherwise
      // start ~1.4 us later per class in front of them: 12-14 us for the 8x8 class, the last one)
      if (!((fb_live >> fb_k) & 1u)) {
        used += ((cnt(CLS) + S::NB - 1) / S::NB + 15) / 16;
        return;
      }
    }
    // (FB: the rotation counts workgroups -- a class's chunk c goes to workgroup (c + chunks of the classes before) % grid)
    const int nbat = run_dct_class<S, PF, SPARSE, false, CLS, false, FB>(
        f, wl.items[CLS], wl.eitems[CLS], cnt(CLS), type, buf, s_binfo[wave], ex, lfs,
        FB ? rotate_wave((int)blockIdx.x, used, (int)gridDim.x) * kWaves + wave : rotate_wave(gw, used, nw), nw, lane, &s_adj,
        wl.fallback[CLS], wl.counts + (kCntFallback0 + CLS) * kCountPitch, sdy, FB ? fb_pre[fb_k] : 0u, wl.fb_any, &fb_rank);
    used += FB ? (nbat + 15) / 16 : nbat;
  };
  run(ShapeTag<S32x32>{}, std::false_type{}, std::integral_constant<int, kClsDct32x32>{}, 5, 0);
  run(ShapeTag<S32x16>{}, std::false_type{}, std::integral_constant<int, kClsDct32x16>{}, 10, 1);
  run(ShapeTag<S16x32>{}, std::false_type{}, std::integral_constant<int, kClsDct16x32>{}, 11, 2);
  run(ShapeTag<S32x8>{}, std::false_type{}, std::integral_constant<int, kClsDct32x8>{}, 8, 3);
  run(ShapeTag<S8x32>{}, std::false_type{}, std::integral_constant<int, kClsDct8x32>{}, 9, 4);
  run(ShapeTag<S16x16>{}, std::true_type{}, std::integral_constant<int, kClsDct16x16>{}, 4, 5);
  run(ShapeTag<S16x8>{}, std::true_type{}, std::integral_constant<int, kClsDct16x8>{}, 6, 6);
  run(ShapeTag<S8x16>{}, std::true_type{}, std::integral_constant<int, kClsDct8x16>{}, 7, 7);
  // (the 8x8 class: only when its kernel ran without the inline fallback; the list stays empty otherwise)
  if constexpr (FB) run(ShapeTag<S8x8>{}, std::true_type{}, std::integral_constant<int, kClsDct8>{}, 0, 8);
  // ALL: the 8x8 class too -- every DCT class in one launch (the dense-route lists of a routed frame; as the form of
  // whole dense frames it lost on the pipelined headline: JXLH_K1_MERGED)
  if constexpr (ALL) run(ShapeTag<S8x8>{}, std::true_type{}, std::integral_constant<int, kClsDct8>{}, 0);
}

// family D: the nine 8x8 special transform types (IDENTITY, DCT2X2, DCT4X4, DCT4X8, DCT8X4, AFV0-3).
// Each lane transforms one block of one channel with the block's 64 coefficients in registers, so a wavefront must
// hold blocks of ONE code path or the nine paths serialise.  The special work list is in raster order (mixed types): a
// wave takes a (chunk of 512 items, type bin) pair, compacts the chunk's items of its bin through a ballot, and runs
// them in batches of kSpecBlk = 21 blocks.  Round 3: the three channels of a batch are transformed TOGETHER -- lane =
// (channel, block), 63 of 64 lanes busy in one pass of the (long, fully unrolled) per-type code -- where round 2 ran the
// transform once per channel on 32 lanes; the tile is in place (a lane reads its row into registers, writes the
// pixels over it) and the dequantised Y a batch's X / B rows need stays in 24 registers.  16K all types: 394 -> see
// profiles/r03_d_large_path.txt.
constexpr int kSpecWaves = 2;
constexpr int kSpecThreads = kSpecWaves * 64;
constexpr int kSpecChunk = 512;
constexpr int kSpecBins = 9;
constexpr int kSpecBlk = 21;                       // blocks per batch: 3 x 21 = 63 transform lanes
constexpr int kSpecIters = (kSpecBlk * 16 + 63) / 64;  // 16-byte groups of one channel of a batch, per lane
__device__ __forceinline__ int special_bin(int type) {  // 1, 2, 3, 12, 13, 14, 15, 16, 17 -> 0..8
  return type <= 3 ? type - 1 : type - 9;
}

// one lane: its block's 64 coefficients into registers, pixels written over them
template <int TYPE>
__device__ __forceinline__ void special_8x8_inplace(float* __restrict__ row, const float* __restrict__ afv_basis = kAfvBasisDev) {
  float c[64], o[64];
#pragma unroll
  for (int i = 0; i < 64; i++) c[i] = row[i];
  special_8x8_t<TYPE>(c, o, afv_basis);
#pragma unroll
  for (int i = 0; i < 64; i++) row[i] = o[i];
}

// BIN0 .. BIN1: the type bins one instantiation handles.  Round 5: two launches -- IDENTITY / DCT2X2 / DCT4X4 / DCT4X8 /
// DCT8X4 (bins 0-4) and AFV0-3 (bins 5-8) -- instead of one: the kernel's registers are the maximum over its bodies, and
// the AFV bodies (a 16 x 16 basis product on top of the 4x4 and 4x8 transforms) set it for everybody (230 VGPRs, 355
// spilled SGPRs, two waves per SIMD).
template <int BIN0, int BIN1>
__global__ __launch_bounds__(kSpecThreads) void k1_special(const FrameDev f, const WorkLists wl) {
  __shared__ float s_tile[kSpecWaves][64 * kSpecPitch];
  __shared__ BlockInfo s_binfo[kSpecWaves][kSpecBlk];
  __shared__ int s_idx[kSpecWaves][kSpecChunk];
  __shared__ AdjTable s_adj;
  __shared__ float s_afv[BIN1 > 5 ? 256 : 1];  // the AFV basis (bins 5-8 only)
  if constexpr (BIN1 > 5)
    for (int i = threadIdx.x; i < 256; i += kSpecThreads) s_afv[i] = kAfvBasisDev[i];
  build_adj_table(f, &s_adj, threadIdx.x, kSpecThreads);  // (ends with a workgroup barrier)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const WorkItem* __restrict__ items = wl.items[kClsSpecial];
  const int count = wl.counts[(kClsSpecial) * kCountPitch];
  float* tile = s_tile[wave];
  BlockInfo* binfo = s_binfo[wave];
  int* mine = s_idx[wave];
  constexpr int kBins = BIN1 - BIN0;
  const int npairs = ((count + kSpecChunk - 1) / kSpecChunk) * kBins;
  for (int pair = blockIdx.x * kSpecWaves + wave; pair < npairs; pair += gridDim.x * kSpecWaves) {
    const int chunk = pair / kBins, bin = BIN0 + pair % kBins;
    // ---- this wave's items of the chunk: the ones whose type falls in `bin`
    int nmine = 0;
    uint32_t packed[kSpecChunk / 64];
#pragma unroll
    for (int i = 0; i < kSpecChunk / 64; i++) {  // all loads in flight before the first ballot
      const int idx = chunk * kSpecChunk + i * 64 + lane;
      packed[i] = items[min(idx, max(count - 1, 0))].packed;
    }
#pragma unroll
    for (int i = 0; i < kSpecChunk / 64; i++) {
      const int idx = chunk * kSpecChunk + i * 64 + lane;
      const bool m = idx < count && special_bin((int)(packed[i] >> 20) & 31) == bin;
      const unsigned long long mask = __ballot(m);
      if (m) mine[nmine + __popcll(mask & ((1ull << lane) - 1ull))] = idx;
      nmine += __popcll(mask);
    }
    wave_sync();
    // every item of the pair has a type of this bin, and the types of a bin share one dequant table
    // (quant_weights.rs:321-343): a wave-uniform pointer instead of a per-lane lookup
    constexpr int kBinType[kSpecBins] = {1, 2, 3, 12, 13, 14, 15, 16, 17};
    int bin_type = kBinType[0];
#pragma unroll
    for (int i = 1; i < kSpecBins; i++) bin_type = bin == i ? kBinType[i] : bin_type;
    const float* __restrict__ bin_table = f.tables + f.table_offset[quant_table_for_type(bin_type)];
    const int k0 = (lane & 15) * 4;  // the four coefficient positions this lane stages, in every block it touches
    for (int b0 = 0; b0 < nmine; b0 += kSpecBlk) {
      const int nb = min(kSpecBlk, nmine - b0);
      if (lane < nb) {
        const WorkItem it = items[mine[b0 + lane]];
        decode_item(f, it, &binfo[lane]);
      }
      wave_sync();
      // ---- stage the three channels: row = ch * kSpecBlk + block, reference order Y, X, B (group.rs:223)
      float dy[4 * kSpecIters];
      auto stage_channel = [&](auto ch_tag) {
        constexpr int CH = decltype(ch_tag)::value;
        const float4 tv = *reinterpret_cast<const float4*>(bin_table + CH * 64 + k0);
        int4 qv[kSpecIters];
        float b_sdy[kSpecIters], b_xcc[kSpecIters], b_bcc[kSpecIters];
#pragma unroll
        for (int j = 0; j < kSpecIters; j++) {  // every global load of the channel in flight first
          const int b = min(j * 4 + (lane >> 4), nb - 1);  // lanes past the batch re-read its last block (discarded)
          const BlockInfo* bp = &binfo[b];
          b_sdy[j] = bp->sdy;
          b_xcc[j] = bp->x_cc;
          b_bcc[j] = bp->b_cc;
          qv[j] = gload_i4<JXLH_NT_COEF>(f.coeffs + bp->coef_off + CH * kGroupArea + k0);
        }
#pragma unroll
        for (int j = 0; j < kSpecIters; j++) {
          const int b = j * 4 + (lane >> 4);
          float d4[4] = {dy[j * 4], dy[j * 4 + 1], dy[j * 4 + 2], dy[j * 4 + 3]};
          BlockInfo bi;
          bi.sdy = b_sdy[j];
          bi.x_cc = b_xcc[j];
          bi.b_cc = b_bcc[j];
          const float4 v = dequant4t<CH>(f, qv[j], tv, bi, &s_adj, d4);
          if constexpr (CH == 1) {
            dy[j * 4] = d4[0];
            dy[j * 4 + 1] = d4[1];
            dy[j * 4 + 2] = d4[2];
            dy[j * 4 + 3] = d4[3];
          }
          if (b < nb) {
            float* dst = tile + (CH * kSpecBlk + b) * kSpecPitch + k0;
            dst[0] = v.x;
            dst[1] = v.y;
            dst[2] = v.z;
            dst[3] = v.w;
          }
        }
      };
      stage_channel(std::integral_constant<int, 1>{});
      stage_channel(std::integral_constant<int, 0>{});
      stage_channel(std::integral_constant<int, 2>{});
      // ---- transform: lane = (channel, block); transform_buffer[0] = lf[0] first
      const int tch = lane / kSpecBlk, tb = lane % kSpecBlk;
      const bool on = tch < 3 && tb < nb;
      const float lf0 = f.lf[min(tch, 2)][binfo[min(tb, nb - 1)].lf_off[min(tch, 2)]];
      wave_sync();
      if (on) {
        float* row = tile + lane * kSpecPitch;  // lane == tch * kSpecBlk + tb
        row[0] = lf0;
        auto in_range = [](int b) { return b >= BIN0 && b < BIN1; };
        // (wave-uniform; only the bodies of this instantiation's bins are compiled in)
        if (in_range(0) && bin == 0) special_8x8_inplace<1>(row);
        if (in_range(1) && bin == 1) special_8x8_inplace<2>(row);
        if (in_range(2) && bin == 2) special_8x8_inplace<3>(row);
        if (in_range(3) && bin == 3) special_8x8_inplace<12>(row);
        if (in_range(4) && bin == 4) special_8x8_inplace<13>(row);
        if (in_range(5) && bin == 5) special_8x8_inplace<14>(row, s_afv);
        if (in_range(6) && bin == 6) special_8x8_inplace<15>(row, s_afv);
        if (in_range(7) && bin == 7) special_8x8_inplace<16>(row, s_afv);
        if (in_range(8) && bin == 8) special_8x8_inplace<17>(row, s_afv);
      }
      wave_sync();
      // ---- store: gather first, then store: all stores of the lane issue back to back
#pragma unroll
      for (int ch = 0; ch < 3; ch++) {
        float* __restrict__ plane = f.planes[ch];
        float4 ov[kSpecIters];
        int oo[kSpecIters];
#pragma unroll
        for (int j = 0; j < kSpecIters; j++) {
          const int b = min(j * 4 + (lane >> 4), nb - 1), p = k0;
          const int px = binfo[b].px_off[ch];
          const float* r = tile + (ch * kSpecBlk + b) * kSpecPitch;
          if (f.tiled) {  // memory order inside the block is (y & 4) * 8 + x * 4 + (y & 3)
            const int x = (p & 31) >> 2, y0 = (p >> 5) * 4;
            const float* src = r + y0 * 8 + x;
            ov[j] = make_float4(src[0], src[8], src[16], src[24]);
            oo[j] = px + p;
          } else {
            ov[j] = make_float4(r[p], r[p + 1], r[p + 2], r[p + 3]);
            oo[j] = px + (p / 8) * (int)f.plane_stride + (p % 8);
          }
        }
#pragma unroll
        for (int j = 0; j < kSpecIters; j++)
          if (j * 4 + (lane >> 4) < nb) *reinterpret_cast<float4*>(plane + oo[j]) = ov[j];
      }
      wave_sync();
    }
    wave_sync();  // `mine` is rewritten by the next pair
  }
}

}  // namespace

// ---- host side -----------------------------------------------------------------------------
// the two-pass units (nblocks / 32 + 16) and the three fused lists (/ 32, / 128, / 256, + 16 each), k_vardct_large.hip
static size_t large_unit_capacity(size_t nblocks) { return nblocks / 8 + 64; }
// k_vardct_large.hip
void launch_vardct_large(hipStream_t s, const FrameDev& f, const WorkLists& wl, int nblk, uint32_t* large_units,
                         size_t unit_capacity, size_t nblocks);

size_t vardct_worklist_bytes(const FrameDev& f) {
  const size_t nblocks = (size_t)f.xblocks * f.yblocks;
  size_t items = 0;
  for (int c = 0; c < kNumClasses; c++) items += nblocks / class_min_area(c) + 1;
  for (int c = 0; c < kClsSpecial; c++) items += 2 * (nblocks / class_min_area(c) + 1);  // entry side items of the DCT
                                                                                         // classes + their dense-route lists
  for (int c = 0; c < kClsSpecial; c++) items += nblocks / (8 * class_min_area(c)) + 4;  // the fallback flags (u32 per batch)
  // + the unit lists of the large transforms: one u32 per 4096 samples of a 256-pixel varblock (two-pass units) and
  //   one per varblock of the smaller types (three lists by slabs per channel; worst case one entry per 32 blocks)
  // + the LLF planes of the large transforms (3 x nblocks floats, k1_large_llf)
  // + the fallback launch's summary words
  return items * sizeof(WorkItem) + 2 * kCountBytes + large_unit_capacity(nblocks) * sizeof(uint32_t) + 64 +
         3 * nblocks * sizeof(float) + (size_t)kFbAny * kFbAnyPitch * sizeof(uint32_t);
}

void vardct_worklist_reset(hipStream_t s, void* worklist_mem, uint32_t* launch_parity) {
  (void)hipMemsetAsync(worklist_mem, 0, 2 * kCountBytes, s);
  *launch_parity = 0;
}

const void* vardct_worklist_counters(const void* worklist_mem, uint32_t launch, size_t* bytes, int* lines) {
  static_assert(kCntFallback0 == kNumClasses + 4 && kCntDense0 == kCntFallback0 + kClsSpecial && kNumClasses == 11,
                "jxlh_frame_k1_counters maps the lines by position");
  *bytes = kCountBytes;
  *lines = kCountLines;
  return reinterpret_cast<const char*>(worklist_mem) + (launch & 1u) * kCountBytes;
}

void launch_vardct_groups(hipStream_t s, const FrameDev& f_in, int group_row0, int group_row1,
                          void* worklist_mem, uint32_t* launch_parity, int* error_flag, int32_t* dense_coeffs,
                          const int* group_list, int n_list, bool has_special, bool has_large, int n_dense_route) {
  const int ngroups = group_list ? n_list : (group_row1 - group_row0) * f_in.xgroups;
  if (ngroups <= 0) return;
  // The value that flags a batch for the fallback launch: unique per launch across the process, never 0.  The flag words
  // are never cleared and start out as whatever the allocation held: a word that happens to equal the epoch sends an
  // already reconstructed batch through the dense pass once more, which writes the same pixels (the two passes are
  // bit-identical, tests/test_gpu_parity.py) behind the direct kernel in stream order.
  static std::atomic<uint32_t> epoch_counter{0};
  FrameDev f = f_in;
  uint32_t epoch = ++epoch_counter;
  if (epoch == 0) epoch = ++epoch_counter;
  f.fb_epoch = (int)epoch;
  // carve the work-list memory: [two sets of counters, one 128-byte line each] [class 0 items] [class 1 items] ...
  WorkLists wl;
  const uint32_t set = (*launch_parity)++ & 1u;
  wl.counts = reinterpret_cast<int*>(reinterpret_cast<char*>(worklist_mem) + set * kCountBytes);
  int* next_counts = reinterpret_cast<int*>(reinterpret_cast<char*>(worklist_mem) + (set ^ 1u) * kCountBytes);
  char* p = reinterpret_cast<char*>(worklist_mem) + 2 * kCountBytes;
  const size_t nblocks = (size_t)f.xblocks * f.yblocks;
  for (int c = 0; c < kNumClasses; c++) {
    wl.items[c] = reinterpret_cast<WorkItem*>(p);
    p += (nblocks / class_min_area(c) + 1) * sizeof(WorkItem);
  }
  for (int c = 0; c < kClsSpecial; c++) {
    wl.eitems[c] = reinterpret_cast<EntryItem*>(p);
    p += (nblocks / class_min_area(c) + 1) * sizeof(EntryItem);
  }
  for (int c = 0; c < kClsSpecial; c++) {
    wl.ditems[c] = reinterpret_cast<WorkItem*>(p);
    p += (nblocks / class_min_area(c) + 1) * sizeof(WorkItem);
  }
  for (int c = 0; c < kClsSpecial; c++) {  // one word per batch of the class (at least 2 varblocks per batch)
    wl.fallback[c] = reinterpret_cast<uint32_t*>(p);
    p += (nblocks / (2 * class_min_area(c)) + 16) * sizeof(uint32_t);
  }
  wl.fb_any = reinterpret_cast<uint32_t*>(p);
  p += (size_t)kFbAny * kFbAnyPitch * sizeof(uint32_t);
  uint32_t* large_units = reinterpret_cast<uint32_t*>(p);  // behind the last list
  const dim3 gscan((ngroups + kScanGroups - 1) / kScanGroups);
  if (f.strip_desc)
    hipLaunchKernelGGL(k1_scan<true>, gscan, dim3(kScanThreads), 0, s, f, wl, group_row0, error_flag, group_list, ngroups,
                       next_counts);
  else if (f.se_entries && f.group_route)
    hipLaunchKernelGGL((k1_scan<false, true, true>), gscan, dim3(kScanThreads), 0, s, f, wl, group_row0, error_flag, group_list,
                       ngroups, next_counts);
  else if (f.se_entries)
    hipLaunchKernelGGL((k1_scan<false, true>), gscan, dim3(kScanThreads), 0, s, f, wl, group_row0, error_flag, group_list,
                       ngroups, next_counts);
  else
    hipLaunchKernelGGL(k1_scan<false>, gscan, dim3(kScanThreads), 0, s, f, wl, group_row0, error_flag, group_list, ngroups,
                       next_counts);
  if (f.strip_desc && f.strip_all_closed) return;  // the host saw the whole map: no tile is left to the class kernels
  // grids: enough waves to fill the chip; kernels stride over their lists (counts are device-side)
  const int nblk = ngroups * kGroupBlocks * kGroupBlocks;
  auto grid_for = [](long work_items, int items_per_wg, int cap) {
    long g = (work_items + items_per_wg - 1) / items_per_wg;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
  };
  // one stream: forking the class kernels onto side streams measured no gain on the d1 mix (event
  // overhead ~ tail savings) and extra streams compete for the runtime's few hardware queues
  // entries form: the direct kernels (+ the fallback launch) when a zero coefficient provably reconstructs to +0.0f
  const int sparse = f.se_entries ? (f.se_direct_ok ? 3 : 2) : f.sp_sorted ? 1 : 0;
  if (sparse && dense_coeffs && (has_special || has_large)) {
    // groups that hold special / large varblocks (flagged by k1_scan) still get a dense slab
    if (sparse >= 2)
      launch_expand_entries(s, dense_coeffs, f.se_entries, f.se_counts, f.se_runs, f.group_dense, f.xgroups * f.ygroups);
    else
      launch_expand_sorted(s, dense_coeffs, f.sp_sorted, f.sp_slot_start, f.group_dense, f.xgroups * f.ygroups);
  }
  // caps measured flat between 768 and 8192 workgroups at 8K (tools/bench_variants.sh)
  const dim3 g8(grid_for(nblk, kWaves * S8x8::NB * 2, 4096)), g16(grid_for(nblk / 2, kWaves * 8 * 2, 2048)),
      g32(grid_for(nblk / 4, kWaves * 4 * 2, 2048));
  bool inline8 = false, dense1632 = false;  // entries form: the dense-pass choices of a frame denser than d1 (se_dense_hint)
  if (f.subsampled) {
    // (the fallback launch has no sub-sampled form: such a frame always takes the inline fallback)
    if (sparse == 3) hipLaunchKernelGGL((k1_dct8<3, true, true>), g8, dim3(kThreads), 0, s, f, wl);
    else if (sparse == 2) hipLaunchKernelGGL((k1_dct8<2, true>), g8, dim3(kThreads), 0, s, f, wl);
    else if (sparse == 1) hipLaunchKernelGGL((k1_dct8<1, true>), g8, dim3(kThreads), 0, s, f, wl);
    else hipLaunchKernelGGL((k1_dct8<0, true>), g8, dim3(kThreads), 0, s, f, wl);
    // the other DCT classes are empty in a sub-sampled frame (k1_scan reports larger varblocks as an error)
  } else {
    const dim3 g1632(std::min(4096u, g16.x + g32.x));
    if (sparse == 3) {
      static const int force_inline = [] {  // JXLH_K1_INLINE=0 / 1: A/B runs of the 8x8 class's two forms
        const char* e = getenv("JXLH_K1_INLINE");
        return e && *e ? atoi(e) : -1;
      }();
      static const int force_dense1632 = [] {  // JXLH_K1_DENSE1632=0 / 1: the same for the 16..32-point classes
        const char* e = getenv("JXLH_K1_DENSE1632");
        return e && *e ? atoi(e) : -1;
      }();
      inline8 = force_inline >= 0 ? force_inline != 0 : f.se_dense_hint >= 2;
      dense1632 = force_dense1632 >= 0 ? force_dense1632 != 0 : f.se_dense_hint >= 1;
      if (inline8) hipLaunchKernelGGL((k1_dct8<3, false, true>), g8, dim3(kThreads), 0, s, f, wl);
      else hipLaunchKernelGGL(k1_dct8<3>, g8, dim3(kThreads), 0, s, f, wl);
      if (dense1632) hipLaunchKernelGGL(k1_dct16_32<2>, g1632, dim3(kThreads), 0, s, f, wl);
      else hipLaunchKernelGGL(k1_dct16_32<3>, g1632, dim3(kThreads), 0, s, f, wl);
    } else if (sparse == 2) {
      hipLaunchKernelGGL(k1_dct8<2>, g8, dim3(kThreads), 0, s, f, wl);
      hipLaunchKernelGGL(k1_dct16_32<2>, g1632, dim3(kThreads), 0, s, f, wl);
    } else if (sparse == 1) {
      hipLaunchKernelGGL(k1_dct8<1>, g8, dim3(kThreads), 0, s, f, wl);
      hipLaunchKernelGGL(k1_dct16_32<1>, g1632, dim3(kThreads), 0, s, f, wl);
    } else {
      if (JXLH_K1_MERGED && ngroups >= 512) {
        hipLaunchKernelGGL((k1_dct16_32<0, false, true>), dim3(std::min(8192u, g8.x + g1632.x)), dim3(kThreads), 0, s, f, wl);
      } else {
        hipLaunchKernelGGL(k1_dct8<0>, g8, dim3(kThreads), 0, s, f, wl);
        hipLaunchKernelGGL(k1_dct16_32<0>, g1632, dim3(kThreads), 0, s, f, wl);
      }
    }
  }
  // what the direct form of k1_dct16_32 left (usually next to nothing: the workgroups read one counter and leave)
  // (a sparse frame leaves a handful of batches: the workgroups read one counter and go; a frame denser than d1 gets
  // the whole chip)
  // (the fallback as TWO launches -- the classes without a 32-point side apart: 32 KB of LDS, three waves per SIMD --
  // measured no better on the outlier frame and 4 % worse on dense ones: profiles/r06_c_density.txt)
  // (nothing can be left when both choices were made)
  if (sparse == 3 && !f.subsampled && !(inline8 && dense1632))
    hipLaunchKernelGGL((k1_dct16_32<2, true>), dim3(std::min(512, std::max(1, nblk / 2048))), dim3(kThreads), 0, s, f, wl);
  // entries form, groups routed to their dense slabs (FrameDev::group_route): the same class kernels in their dense
  // form on those groups' lists; the grids follow the routed share of the frame
  if (sparse >= 2 && n_dense_route > 0) {
    WorkLists wd = wl;
    for (int c = 0; c < kClsSpecial; c++) wd.items[c] = wl.ditems[c];
    wd.counts = wl.counts + kCntDense0 * kCountPitch;
    FrameDev fd = f;
    fd.se_entries = nullptr;
    const long dblk = (long)std::min(n_dense_route, ngroups) * kGroupBlocks * kGroupBlocks;
    const dim3 d8(grid_for(dblk, kWaves * S8x8::NB * 2, 4096));
    if (f.subsampled) {
      hipLaunchKernelGGL((k1_dct8<0, true>), d8, dim3(kThreads), 0, s, fd, wd);
    } else {
      // (one launch for every DCT class of the routed groups: the form big dense frames take)
      hipLaunchKernelGGL((k1_dct16_32<0, false, true>), dim3(std::min(8192u, d8.x + (unsigned)grid_for(dblk / 2, kWaves * 8 * 2, 4096))),
                         dim3(kThreads), 0, s, fd, wd);
    }
  }
  // an empty special list (the d1 mix) pays for every launched workgroup: the grid follows the list's worst case
  if (has_special)
  {
    hipLaunchKernelGGL((k1_special<0, 5>), dim3(grid_for((long)(nblk / kSpecChunk + 1) * 5, kSpecWaves, 2048)),
                       dim3(kSpecThreads), 0, s, f, wl);
    hipLaunchKernelGGL((k1_special<5, 9>), dim3(grid_for((long)(nblk / kSpecChunk + 1) * 4, kSpecWaves, 2048)),
                       dim3(kSpecThreads), 0, s, f, wl);
  }
  // (the large transforms on a side stream next to the memory-bound DCT classes, one workgroup per CU: K1 of the 16K
  // all-types frame 1.872 -> 1.847 ms, 1.937 with their usual two per CU -- inside the noise, not kept)
  if (has_large) launch_vardct_large(s, f, wl, nblk, large_units, large_unit_capacity(nblocks), nblocks);
}

}  // namespace jxlh
