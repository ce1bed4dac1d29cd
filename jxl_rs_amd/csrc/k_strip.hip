// K1 + K2 + K3 in ONE kernel: dequantisation, chroma-from-luma, LLF-from-LF, the variable-size IDCT, Gaborish, EPF1 and
// EPF2 without the 12 B/px write + 12 B/px read of the reconstructed planes between the transform and the filter
// kernels.
//
// What the reference does: decode_vardct_group hands a group's pixels straight to the render pipeline
// (jxl/src/frame/group.rs:579-611 -> render/mod.rs:128-137) and the low-memory pipeline streams rows through the
// filter stages in ring buffers (render/low_memory_pipeline/render_group.rs:21-503, stage list frame/render.rs:569-622):
// the unfiltered image never exists in DRAM.  Here neither.
//
// Decomposition.  The frame is cut into STRIPS, 64 pixels wide, and each strip into a few BANDS of tile rows; one
// persistent workgroup walks a (band, strip) downwards in STEPS of one 64x64 tile:
//
//   window (LDS, 3 channels x 72 rows x 72 floats)        frame rows of step t
//     rows  0..7   carry: the previous tile's last 8 rows   64t-8 .. 64t-1
//     rows  8..71  the tile the step reconstructs           64t   .. 64t+63
//     cols  0..3 / 68..71  halo: the neighbour strips' edge columns, through HBM (exchange buffer + progress flag)
//     cols  4..67  the strip's own 64 columns
//
//   step t:  (1) IDCT of tile t IN PLACE in the window: a varblock's coefficient (u, v) is dequantised into the window
//                position of pixel (x0 + u, y0 + v); pass 1 (along u) transforms window rows, pass 2 (along v) window
//                columns -- no staging tile, the transform costs no LDS beyond the pixels themselves.
//            (2) the tile's 4 left / right columns are published (agent-scope stores, then one progress flag)
//            (3) the neighbours' columns of the same step are fetched once their flags say so
//            (4) rows 64..71 are saved (they are the next step's carry: the stages below work in place)
//            (5) Gaborish -> EPF1 -> EPF2 in place on shrinking regions (filters_core.inc: the same code and operation
//                order as k23_fused_filters), output rows 64t-4 .. 64t+59 straight to the result planes
//            (6) the saved rows become rows 0..7
//
// The vertical filter halo therefore never leaves LDS; the horizontal one costs 2 x 4 of 64 columns (12.5 % of the
// pixels written once and read once, 1 KB contiguous per channel and side).  A band's first step reconstructs the
// tile above the band without filtering it (seed), its last step the tile below (for the 4 rows the last output rows
// read): 2 extra transforms per band and strip instead of a dependency between bands.
//
// Which tiles: a tile is transformed here iff every varblock touching it lies inside it and is a DCT with sides <= 32
// (k1_scan decides per tile and writes a descriptor per block: varblocks need not be aligned in the format, frame/
// modular/mod.rs:1061-1064 only confines them to their group, but libjxl's encoder aligns them to their own size,
// which closes every 64x64 tile).  Any other tile is reconstructed by K1's class kernels into `planes` as before and
// merely LOADED into the window here, so a frame may mix both kinds freely.
//
// Synchronisation.  Workgroups take a ticket; ticket k = strip k % S of band k / S, so a workgroup's neighbours hold
// adjacent tickets.  A workgroup publishes step q before it waits for its neighbours' step q, hence with R resident
// workgroups the chain of waits is a staircase (ticket R-1-j is held at step j by its unstarted right neighbour at
// worst) and ticket 0 always runs to completion when R exceeds the step count; waits are bounded by a deadline that
// raises JXLH_ERR_DEVICE instead of hanging the device.
//
// Bit-exactness: every arithmetic step is the code K1 and the fused filter kernel run (dequant4, llf_from_lf,
// idct1d, filters_core.inc) in the same order; only where the operands live differs.
#include "k_vardct_common.h"

#ifndef JXLH_STRIP_WAVES
#define JXLH_STRIP_WAVES 12
#endif
#ifndef JXLH_STRIP_WPE
#define JXLH_STRIP_WPE 6
#endif
#ifndef JXLH_FAST_RECIP
#define JXLH_FAST_RECIP 1
#endif
#ifndef JXLH_E1_ROLLED
#define JXLH_E1_ROLLED 1
#endif
#ifndef JXLH_SPARSE_MAX
#define JXLH_SPARSE_MAX 62
#endif
#ifndef JXLH_DENSE_ITEMS
#define JXLH_DENSE_ITEMS 40
#endif

#ifdef JXLH_STRIP_PROF
// development variant (tools/build_variant.sh ... -DJXLH_STRIP_PROF): thread 0 of every workgroup accumulates the
// s_memrealtime ticks (10 ns) it spends between the phase marks; read with jxlh_strip_prof_read
__device__ unsigned long long g_strip_prof[16];
#define PROF_MARK(i)                                          \
  do {                                                        \
    if (tid_kernel == 0) {                                    \
      const unsigned long long now_ = now_ticks();            \
      s_prof[i] += now_ - s_prof_t;                           \
      s_prof_t = now_;                                        \
    }                                                         \
  } while (0)
#else
#define PROF_MARK(i) \
  do {               \
  } while (0)
#endif

namespace jxlh {
namespace {

constexpr int kTW = 64, kTH = 64, kB = 4;
constexpr int kCarry = 8;               // rows of the previous tile kept in the window
constexpr int kBW = kTW + 2 * kB;       // 72 floats = 18 strips of 4
constexpr int kBH = kTH + 2 * kB;       // 72 rows: 8 carry + 64 tile; the filters' output tile is rows 4..67
constexpr int kStrips = kBW / 4;        // 18
constexpr int kPlane = kBW * kBH;
constexpr int kNW = JXLH_STRIP_WAVES, kNT = kNW * 64;
constexpr int kUse = 62;                // useful lanes of a wavefront in the dense filter forms (lanes 1..62)
constexpr int kSigW = kBW / 8 + 2, kSigH = kBH / 8 + 2;
constexpr int kDenseItems = JXLH_DENSE_ITEMS, kSparseMax = JXLH_SPARSE_MAX;
static_assert(kCarry == 2 * kB, "the carry is the filter halo above the output tile plus the rows the output lags");
static_assert(kNW * kUse >= ((kTH + 6) / 2) * kStrips, "one pass per in-place stage");
static_assert(kNW >= 12, "one wavefront per (quadrant, channel) in the transform phase");

struct FusedArgs {  // what filters_core.inc and the stage driver read (same names as k23_fused_filters' argument)
  float* out[3];
  const float* inv_sigma;
  uint32_t stride, sigma_stride;
  int w, h;
  float gab_k[3][3];
  float scale[3];
  float sm0, bsm0, sm1, bsm1, sm2, bsm2;
};

struct StripArgs {
  FusedArgs fa;
  const uint2* desc;         // per block: {type | dx << 5 | dy << 7 | off64 << 9 | 1 << 31, raw_quant of the varblock}
  const uint8_t* tile_mode;  // per tile: 0 = transformed here, 1 = loaded from f.planes (K1's class kernels)
  float* xchg;               // [channel][side][strip][row][4]
  int* flags;                // bands * strips progress counters, then the ticket counter
  int* error_flag;
  int strips, tile_rows, bands;
  int xchg_rows;
  unsigned long long deadline_ticks;  // s_memrealtime ticks (100 MHz) a wait may last
};

template <class T>
__device__ __forceinline__ T& at_bytes(float* base, uint32_t byte_off) {
  return *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off);
}
template <class T>
__device__ __forceinline__ const T& at_bytes(const float* base, uint32_t byte_off) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}

// Lane <-> strip mapping of the dense filter forms: a wavefront holds 64 CONSECUTIVE strips of the (row pair, strip)
// raster of the 18-strip rows, so the strip to the left / right is the previous / next lane of the whole wavefront
// (DPP wave shifts; lane 0 / 63 get 0 and are helper lanes that recompute the neighbouring wavefront's edge item).
__device__ __forceinline__ float dpp_from_left(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_from_right(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

#include "filters_core.inc"

// Overwrites out-of-frame positions of a region [kB - m, kB + T + m) of the window with the values at their mirrored
// in-frame coordinates (jxl/src/render/simple_pipeline/run_stage.rs:129-146: every stage sees ITS input mirrored).
__device__ __forceinline__ void mirror_fill(float* __restrict__ buf, int m, int tx0, int ty0, int w, int h, int tid) {
  const int rw = kTW + 2 * m, rh = kTH + 2 * m;
  for (int idx = tid; idx < rw * rh; idx += kNT) {
    const int bx = kB - m + idx % rw, by = kB - m + idx / rw;
    const int fx = tx0 - kB + bx, fy = ty0 - kB + by;
    if (fx >= 0 && fx < w && fy >= 0 && fy < h) continue;
    const int sx = mirror(fx, w) - (tx0 - kB), sy = mirror(fy, h) - (ty0 - kB);
    if (sx < 0 || sx >= kBW || sy < 0 || sy >= kBH) continue;  // outside this window: never consumed
#pragma unroll
    for (int c = 0; c < 3; c++) buf[c * kPlane + by * kBW + bx] = buf[c * kPlane + sy * kBW + sx];
  }
}

__device__ __forceinline__ unsigned long long now_ticks() { return __builtin_amdgcn_s_memrealtime(); }

// a block's descriptor, validated: whatever stale or hostile bits it holds, the varblock it describes lies inside the
// 32x32 quadrant of the tile its block is in, is one of the nine DCT shapes with sides <= 32 and its coefficients
// inside the group's slab
struct Blk {
  bool on;
  int type, dx, dy, lcx, lcy, off64;
};
__device__ __forceinline__ Blk decode_desc(uint32_t d, int bx, int by) {
  Blk b;
  b.type = (int)(d & 31u);
  b.dx = (int)((d >> 5) & 3u);
  b.dy = (int)((d >> 7) & 3u);
  b.off64 = (int)((d >> 9) & 1023u);
  const bool known = (d >> 31) != 0 && b.type < JXLH_NUM_TRANSFORMS && class_of_type_reg(min(b.type, JXLH_NUM_TRANSFORMS - 1)) < kClsSpecial;
  b.lcx = known ? log2_covered_x_reg(b.type) : 0;
  b.lcy = known ? log2_covered_y_reg(b.type) : 0;
  const int cx = 1 << b.lcx, cy = 1 << b.lcy;
  b.on = known && b.dx < cx && b.dy < cy && b.dx <= (bx & 3) && b.dy <= (by & 3) && (bx & 3) - b.dx + cx <= 4 &&
         (by & 3) - b.dy + cy <= 4 && b.off64 + cx * cy <= 1024;
  return b;
}

// one task of an IDCT pass: N floats at stride STEP (1 = a window row, 16-byte accesses; kBW = a window column)
template <int N, int STEP>
__device__ __forceinline__ void idct_line(float* __restrict__ p) {
  float x[N];
  if constexpr (STEP == 1) {
#pragma unroll
    for (int i = 0; i < N; i += 4) {
      const float4 v = lds_load4(p + i);
      x[i] = v.x; x[i + 1] = v.y; x[i + 2] = v.z; x[i + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < N; i++) x[i] = p[i * STEP];
  }
  idct1d<N, true>(x);
  if constexpr (STEP == 1) {
#pragma unroll
    for (int i = 0; i < N; i += 4) lds_store4(p + i, make_float4(x[i], x[i + 1], x[i + 2], x[i + 3]));
  } else {
#pragma unroll
    for (int i = 0; i < N; i++) p[i * STEP] = x[i];
  }
}

// LLF-from-LF of one varblock-channel into its window corner (transform.rs:412-509 via llf_from_lf)
template <int CY, int CX>
__device__ __forceinline__ void llf_to_window(const float* __restrict__ lf, int xblocks, float* __restrict__ org) {
  float a[CY * CX];
#pragma unroll
  for (int y = 0; y < CY; y++)
#pragma unroll
    for (int x = 0; x < CX; x++) a[y * CX + x] = lf[y * xblocks + x];
  llf_from_lf<CY, CX>(a);
  constexpr int MN = cmin(CY, CX), MX = cmax(CY, CX);
  constexpr bool kWide = CY < CX;
#pragma unroll
  for (int r = 0; r < MN; r++)
#pragma unroll
    for (int q = 0; q < MX; q++) {
      // stored position r * max(R, C) + q: (u, v) = (r, q) for R >= C, (q, r) for the wide shapes (m_addr)
      const int u = kWide ? q : r, v = kWide ? r : q;
      org[v * kBW + u] = a[r * MX + q];
    }
}

template <bool GAB, bool E1, bool E2>
__global__ __launch_bounds__(kNT, JXLH_STRIP_WPE) void k123_strip(const FrameDev f, const StripArgs sa) {
  __shared__ __attribute__((aligned(16))) float s_buf[3 * kPlane];
  __shared__ __attribute__((aligned(16))) float s_save[3 * kCarry * kBW];
  __shared__ float s_sigma[kSigH * kSigW];
  __shared__ uint16_t s_list[kBH * kStrips];
  __shared__ uint8_t s_wtask[12][448];  // per wavefront: pass 1 / pass 2 tasks of its quadrant by length class
  __shared__ uint32_t s_desc[64];
  __shared__ float s_sdy[64];
  __shared__ float s_lf[192];  // the tile's LF samples: [channel][block row][block column]
  __shared__ int s_cnt, s_nsw, s_ticket, s_abort, s_pub;
#ifdef JXLH_STRIP_PROF
  __shared__ unsigned long long s_prof[16], s_prof_t;
  if (threadIdx.x == 0) {
    for (int i = 0; i < 16; i++) s_prof[i] = 0;
    s_prof_t = now_ticks();
  }
#endif
  const FusedArgs& a = sa.fa;
  const int tid_kernel = threadIdx.x;
  constexpr int kBorder = (GAB ? 1 : 0) + (E1 ? 2 : 0) + (E2 ? 1 : 0);
  static_assert(kBorder >= 1 && kBorder <= kB, "at least one stage");

  if (tid_kernel == 0) {
    s_ticket = atomicAdd(&sa.flags[sa.bands * sa.strips], 1);
    s_cnt = 0;
    s_nsw = 0;
    s_abort = 0;
  }
  __syncthreads();
  const int ticket = __builtin_amdgcn_readfirstlane(s_ticket);  // uniform: everything derived from it lives in SGPRs
  const int band = ticket / sa.strips, strip = ticket % sa.strips;
  if (band >= sa.bands) return;
  const int tr0 = (int)((long)band * sa.tile_rows / sa.bands), tr1 = (int)((long)(band + 1) * sa.tile_rows / sa.bands);
  const int t_begin = tr0 > 0 ? tr0 - 1 : 0;
  const int y_lo = tr0 * kTH, y_hi = min(tr1 * kTH, a.h);
  // the step below the band exists for output rows [64 tr1 - 4, 64 tr1)
  const int t_end = (tr1 * kTH - kB < y_hi) ? tr1 : tr1 - 1;
  const int tx0 = strip * kTW;
  int* my_flag = sa.flags + band * sa.strips + strip;
  const int gx = strip / 4;  // group column (256 = 4 tiles)

  // What a step's transform phase needs from global memory before it can address anything -- the tile's mode, its 64
  // block descriptors, its 3 x 64 LF samples -- is fetched one step ahead into registers (threads 0..191).
  uint2 nx_desc = make_uint2(0u, 1u);
  float nx_lf = 0.0f;
  int nx_mode = 1;
  auto prefetch_meta = [&](int tt, int tid) {
    if (tt >= sa.tile_rows || tt > t_end) return;
    nx_mode = sa.tile_mode[tt * sa.strips + strip];
    if (tid < 192) {
      const int bi = tid & 63, gbx = strip * 8 + (bi & 7), gby = tt * 8 + (bi >> 3);
      const bool in = gbx < f.xblocks && gby < f.yblocks;
      const size_t at = (size_t)gby * f.xblocks + gbx;
      if (tid < 64) nx_desc = in ? sa.desc[at] : make_uint2(0u, 1u);
      nx_lf = in ? f.lf[tid >> 6][at] : 0.0f;
    }
  };
  prefetch_meta(t_begin, tid_kernel);

  for (int t = t_begin; t <= t_end; t++) {
    // Opaque copy of the thread id, per step: what a phase derives from it (addresses, lane masks, task geometry) is
    // then computed where it is used instead of being hoisted out of the step loop, where it would sit in (spilled)
    // registers for the whole kernel.
    int tid = tid_kernel;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const bool has_tile = t < sa.tile_rows;
    const int ty0 = t * kTH - kB;  // frame row of window row kB (the filters' output tile starts there)
    const int seq = t - t_begin + 1;
    if (has_tile) {
      const int mode = nx_mode;
      if (tid < 64) {
        s_desc[tid] = nx_desc.x;
        s_sdy[tid] = f.inv_global_scale / (float)(uint32_t)nx_desc.y;  // group.rs:153
      }
      if (tid < 192) s_lf[tid] = nx_lf;
      if (tid == 0) s_pub = 0;
      __syncthreads();
      prefetch_meta(t + 1, tid);
      PROF_MARK(0);
      // Everything up to the filters is WAVE-LOCAL: wavefront (q, c) = (quadrant of the tile, channel) takes its 16
      // blocks' worth of one channel from coefficients to pixels, publishes the 4 edge columns of its 32 rows and
      // fetches the neighbour strip's 4 columns beside them -- only wave-scope synchronisation, so the twelve
      // wavefronts (and the other workgroup of the CU) overlap each other's memory latencies the way K1's batches do.
      // (k1_scan hands a tile to this kernel only if every varblock lies inside one 32x32 quadrant.)
      const int q = wave / 3, ch = wave % 3;
      const int qbx = (q & 1) * 4, qby = (q >> 1) * 4;  // the quadrant's first block inside the tile
      if (mode == 0 && wave < 12) {
        uint8_t* wl = s_wtask[wave];
        // ---- (1a) this wavefront's tasks of the two IDCT passes, by length: [0, 128) 8, [128, 192) 16, [192, 224) 32
        // pass 1 candidates: (row y of the quadrant, block column): a task iff the block is the leftmost of its
        // varblock; pass 2: (column x, block row): iff it is the topmost
        int n1[3] = {0, 0, 0}, n2[3] = {0, 0, 0};
#pragma unroll
        for (int r = 0; r < 2; r++) {
          {
            const int cand = r * 64 + lane, y = cand >> 2, bxc = cand & 3;
            const Blk b = decode_desc(s_desc[(qby + (y >> 3)) * 8 + qbx + bxc], qbx + bxc, qby + (y >> 3));
            const bool act = b.on && b.dx == 0;
#pragma unroll
            for (int k = 0; k < 3; k++) {
              const unsigned long long m = __ballot(act && b.lcx == k);
              if (act && b.lcx == k) wl[(k == 0 ? 0 : k == 1 ? 128 : 192) + n1[k] + __popcll(m & ((1ull << lane) - 1ull))] = (uint8_t)cand;
              n1[k] += __popcll(m);
            }
          }
          {
            const int cand = r * 64 + lane, x = cand & 31, byc = cand >> 5;
            const Blk b = decode_desc(s_desc[(qby + byc) * 8 + qbx + (x >> 3)], qbx + (x >> 3), qby + byc);
            const bool act = b.on && b.dy == 0;
#pragma unroll
            for (int k = 0; k < 3; k++) {
              const unsigned long long m = __ballot(act && b.lcy == k);
              if (act && b.lcy == k) wl[224 + (k == 0 ? 0 : k == 1 ? 128 : 192) + n2[k] + __popcll(m & ((1ull << lane) - 1ull))] = (uint8_t)cand;
              n2[k] += __popcll(m);
            }
          }
        }
        PROF_MARK(1);
        // ---- (1c) LLF-from-LF (one lane per varblock) over the lowest frequencies; the dequantisation below skips
        // that corner, which the LLF overwrites in the reference (transform.rs:450)
        if (lane < 16) {
          const int bx = qbx + (lane & 3), by = qby + (lane >> 2);
          const Blk b = decode_desc(s_desc[by * 8 + bx], bx, by);
          if (b.on && b.dx == 0 && b.dy == 0) {
            const float* lf = s_lf + ch * 64 + by * 8 + bx;
            float* org = s_buf + ch * kPlane + (kCarry + 8 * by) * kBW + kB + 8 * bx;
            switch (b.type) {
              case 0: org[0] = lf[0]; break;
              case 4: llf_to_window<2, 2>(lf, 8, org); break;
              case 5: llf_to_window<4, 4>(lf, 8, org); break;
              case 6: llf_to_window<2, 1>(lf, 8, org); break;
              case 7: llf_to_window<1, 2>(lf, 8, org); break;
              case 8: llf_to_window<4, 1>(lf, 8, org); break;
              case 9: llf_to_window<1, 4>(lf, 8, org); break;
              case 10: llf_to_window<4, 2>(lf, 8, org); break;
              default: llf_to_window<2, 4>(lf, 8, org); break;  // 11
            }
          }
        }
        // ---- (1b) dequantisation + chroma-from-luma straight into the window (dequant_block, group.rs:137-177): a lane
        // takes 4 consecutive stored coefficients; the X / B wavefronts dequantise the Y values they need themselves
        {
          const int g = (t / 4) * f.xgroups + gx;
          const int cti = t * f.cmap_stride + strip;  // the tile IS a colour tile (64 x 64)
          BlockInfo bi;
          bi.x_cc = f.base_x + (float)f.ytox[cti] / f.color_factor;  // color_correlation_map.rs:76-78
          bi.b_cc = f.base_b + (float)f.ytob[cti] / f.color_factor;
#pragma unroll
          for (int half = 0; half < 2; half++) {
            int4 qo[2], qy[2];
            float4 to[2], ty[2];
            int woff[2], skip[2];
            float sdy[2];
#pragma unroll
            for (int it = 0; it < 2; it++) {
              const int idx = (half * 2 + it) * 64 + lane;  // 16 blocks x 16 chunks
              const int bq = idx >> 4, qd = idx & 15;
              const int bx = qbx + (bq & 3), by = qby + (bq >> 2);
              const Blk b = decode_desc(s_desc[by * 8 + bx], bx, by);
              const int k = ((b.dy << b.lcx) + b.dx) * 64 + 4 * qd;  // index inside the varblock's stored coefficients
              const int lr = 3 + b.lcy, lc = 3 + b.lcx;              // log2 of R, C
              const bool wide = lr < lc;
              // stored in[u * R + v] for R >= C, in[v * C + u] for the wide shapes (tests.rs:119-132)
              const int u = wide ? (k & ((1 << lc) - 1)) : (k >> lr), v = wide ? (k >> lc) : (k & ((1 << lr) - 1));
              // woff < 0: nothing to do; bit 30: the four values go down a column (stride kBW) instead of along a row
              woff[it] = !b.on ? -1 : (((kCarry + 8 * (by - b.dy) + v) * kBW + kB + 8 * (bx - b.dx) + u) | (wide ? 0 : 1 << 30));
              // leading elements inside the LLF corner (u < cx, v < cy): written by the LLF lanes instead
              skip[it] = wide ? ((u == 0 && v < (1 << b.lcy)) ? (1 << b.lcx) : 0) : ((v == 0 && u < (1 << b.lcx)) ? (1 << b.lcy) : 0);
              sdy[it] = s_sdy[by * 8 + bx];
              const int qt = b.on ? quant_table_for_type(b.type) : 0;
              const int tsize = quant_table_size(qt);
              const float* tb = f.tables + f.table_offset[qt] + k;
              const int* cf = f.coeffs + ((size_t)g * 3 * kGroupArea + b.off64 * 64 + k);
              qo[it] = qy[it] = make_int4(0, 0, 0, 0);
              to[it] = ty[it] = make_float4(0.f, 0.f, 0.f, 0.f);
              if (b.on) {
                qy[it] = gload_i4<true>(cf + kGroupArea);
                ty[it] = *reinterpret_cast<const float4*>(tb + tsize);
                if (ch != 1) {
                  qo[it] = gload_i4<true>(cf + ch * kGroupArea);
                  to[it] = *reinterpret_cast<const float4*>(tb + ch * tsize);
                }
              }
            }
#pragma unroll
            for (int it = 0; it < 2; it++) {
              if (woff[it] < 0) continue;
              bi.sdy = sdy[it];
              float dy[4];
              float4 vv = dequant4<1>(f, qy[it], ty[it], bi, dy);
              if (ch == 0) vv = dequant4<0>(f, qo[it], to[it], bi, dy);
              else if (ch == 2) vv = dequant4<2>(f, qo[it], to[it], bi, dy);
              float* dc = s_buf + ch * kPlane + (woff[it] & 0xffffff);
              const bool col = (woff[it] >> 30) != 0;
              const int sk = skip[it];
              if (!col && sk == 0) {
                lds_store4(dc, vv);
              } else {
                const int st = col ? kBW : 1;
                if (sk < 1) dc[0] = vv.x;
                if (sk < 2) dc[st] = vv.y;
                if (sk < 3) dc[2 * st] = vv.z;
                if (sk < 4) dc[3 * st] = vv.w;
              }
            }
          }
        }
        wave_sync();
        PROF_MARK(2);
        // ---- (1d) pass 1 (along u: window rows), (1e) pass 2 (along v: window columns); idct2d.rs:111-131 order
        float* qorg = s_buf + ch * kPlane + (kCarry + 8 * qby) * kBW + kB + 8 * qbx;
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
#pragma unroll
          for (int k = 2; k >= 0; k--) {  // the long transforms first
            const int n = pass ? n2[k] : n1[k];
            for (int i0 = 0; i0 < n; i0 += 64) {
              if (i0 + lane < n) {
                const int e = wl[pass * 224 + (k == 0 ? 0 : k == 1 ? 128 : 192) + i0 + lane];
                if (pass == 0) {
                  float* p = qorg + (e >> 2) * kBW + (e & 3) * 8;
                  if (k == 2) idct_line<32, 1>(p);
                  else if (k == 1) idct_line<16, 1>(p);
                  else idct_line<8, 1>(p);
                } else {
                  float* p = qorg + (e >> 5) * 8 * kBW + (e & 31);
                  if (k == 2) idct_line<32, kBW>(p);
                  else if (k == 1) idct_line<16, kBW>(p);
                  else idct_line<8, kBW>(p);
                }
              }
            }
          }
          wave_sync();
          PROF_MARK(3 + pass);
        }
      } else if (mode != 0) {
        // ---- (1') a tile K1's class kernels reconstructed: 8x8-tiled planes -> window (a lane fetches 4 rows of a column)
        for (int idx = tid; idx < 1024; idx += kNT) {
          const int col = idx & 63, yg = idx >> 6;
          const int gbx = strip * 8 + (col >> 3), gby = t * 8 + (yg >> 1);
          if (gbx >= f.xblocks || gby >= f.yblocks) continue;
          const uint32_t off = 4u * ((uint32_t)(gby * f.xblocks + gbx) * 64u + (uint32_t)((yg & 1) * 32 + (col & 7) * 4));
#pragma unroll
          for (int c = 0; c < 3; c++) {
            const float4 v = gload_f4<false>(&at_bytes<float>(f.planes[c], off));
            float* d = s_buf + c * kPlane + (kCarry + yg * 4) * kBW + kB + col;
            d[0] = v.x;
            d[kBW] = v.y;
            d[2 * kBW] = v.z;
            d[3 * kBW] = v.w;
          }
        }
        __syncthreads();  // a wavefront publishes rows other wavefronts loaded
      }
      // ---- (2) publish the 4 edge columns of this wavefront's 32 rows: written through at agent scope (the neighbour
      // runs on another XCD, whose L2 is not coherent with this one's), acknowledged (vmcnt) before the count goes up;
      // the last of the twelve raises the strip's progress flag.  (3) then the neighbour's columns beside them.
      if (sa.strips > 1 && wave < 12) {
        const int side = q & 1, row = (q >> 1) * 32 + (lane & 31);
        if (lane < 32) {
          const float4 v = lds_load4(s_buf + ch * kPlane + (kCarry + row) * kBW + (side ? kTW : kB));
          unsigned long long* dst = reinterpret_cast<unsigned long long*>(
              sa.xchg + ((((size_t)ch * 2 + side) * sa.strips + strip) * sa.xchg_rows + (size_t)t * kTH + row) * 4);
          const unsigned long long lo = (unsigned long long)__float_as_uint(v.x) | (unsigned long long)__float_as_uint(v.y) << 32;
          const unsigned long long hi = (unsigned long long)__float_as_uint(v.z) | (unsigned long long)__float_as_uint(v.w) << 32;
          __hip_atomic_store(dst, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(dst + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wavefront's stores are acknowledged
        if (lane == 0 && atomicAdd(&s_pub, 1) == 11) __hip_atomic_store(my_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        PROF_MARK(5);
        const int nb = strip + (side ? 1 : -1);
        if (nb >= 0 && nb < sa.strips) {
          if (lane == 0) {
            const int* fl = sa.flags + band * sa.strips + nb;
            const unsigned long long t0 = now_ticks();
            int spins = 0;
            while (__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < seq) {
              __builtin_amdgcn_s_sleep(2);
              if ((++spins & 255) == 0 && now_ticks() - t0 > sa.deadline_ticks) {
                atomicExch(sa.error_flag, JXLH_ERR_DEVICE);
                s_abort = 1;
                break;
              }
            }
          }
          PROF_MARK(6);
          // (lane 0's loop ends before the wavefront goes on: the other lanes wait at the reconvergence point)
          if (lane < 32) {
            const unsigned long long* src = reinterpret_cast<const unsigned long long*>(
                sa.xchg + ((((size_t)ch * 2 + (1 - side)) * sa.strips + nb) * sa.xchg_rows + (size_t)t * kTH + row) * 4);
            const unsigned long long lo = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long hi = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lds_store4(s_buf + ch * kPlane + (kCarry + row) * kBW + (side ? kB + kTW : 0),
                       make_float4(__uint_as_float((uint32_t)lo), __uint_as_float((uint32_t)(lo >> 32)),
                                   __uint_as_float((uint32_t)hi), __uint_as_float((uint32_t)(hi >> 32))));
          }
        }
        PROF_MARK(7);
      }
    }
    __syncthreads();
    PROF_MARK(8);
    if (s_abort) {
      // a neighbour never showed up (deadline): let the rest of the band fall through as well
      if (tid == 0) __hip_atomic_store(my_flag, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    // ---- (4) the next step's carry: the last 8 rows of the window as they are now (the stages work in place)
    if (t < t_end) {
      for (int i = tid; i < 3 * kCarry * kStrips; i += kNT) {
        const int c = i / (kCarry * kStrips), r = (i / kStrips) % kCarry, s4 = (i % kStrips) * 4;
        lds_store4(s_save + (c * kCarry + r) * kBW + s4, lds_load4(s_buf + c * kPlane + (kBH - kCarry + r) * kBW + s4));
      }
    }
    if (t >= tr0) {
      // ---- (5) the stage list on the window; output rows ty0 .. ty0 + 63 clipped to the band
      const bool edge = tx0 - kB < 0 || ty0 - kB < 0 || tx0 + kTW + kB > a.w || ty0 + kTH + kB > a.h;
      const int sbx0 = max(tx0 - kB, 0) >> 3, sby0 = max(ty0 - kB, 0) >> 3;
      if constexpr (E1 || E2) {
        if (tid < kSigH * kSigW) {
          const int sx = min(sbx0 + tid % kSigW, (a.w - 1) >> 3), sy = min(sby0 + tid / kSigW, (a.h - 1) >> 3);
          s_sigma[tid] = at_bytes<float>(a.inv_sigma, 4u * ((uint32_t)sy * a.sigma_stride + (uint32_t)sx));
        }
      }
      if (edge) {
        __syncthreads();
        mirror_fill(s_buf, kBorder, tx0, ty0, a.w, a.h, tid);
      }
      __syncthreads();
      PROF_MARK(9);

      auto run_stage = [&](auto stage_tag, auto margin_tag) {
        constexpr int STAGE = decltype(stage_tag)::value;  // 0 gaborish, 1 epf1, 2 epf2
        constexpr int margin = decltype(margin_tag)::value;
        constexpr bool last = margin == 0;
        constexpr int rows = kTH + 2 * margin;
        constexpr int n = (rows / 2) * kStrips;
        static_assert(rows % 2 == 0 && n <= kNW * kUse, "4x2 items, one pass");
        int tid = tid_kernel;
        asm volatile("" : "+v"(tid));
        const int ln = tid & 63, wv = tid >> 6;
        auto store = [&](int bx0, int fy, int fx0, int c, float4 o) {
          if (bx0 >= kB && bx0 < kB + kTW && fy >= y_lo && fy < y_hi && fx0 < a.w)
            at_bytes<float4>(a.out[c], 4u * ((uint32_t)fy * a.stride + (uint32_t)fx0)) = o;
        };
        if constexpr (STAGE == 2 && last) {
          // EPF2 as the last stage: one strip (4 pixels of one row) per lane
          constexpr int ns = rows * kStrips;
          auto strip_geom = [&](int tt, int& by, int& bx0, int& fy, int& fx0, float& sigma) {
            by = kB + tt / kStrips;
            bx0 = (tt % kStrips) * 4;
            fy = ty0 - kB + by;
            fx0 = tx0 - kB + bx0;
            const int sx = (min(max(fx0, 0), a.w - 1) >> 3) - sbx0, sy = (min(max(fy, 0), a.h - 1) >> 3) - sby0;
            sigma = s_sigma[sy * kSigW + sx];
          };
#pragma unroll 1
          for (int t0 = 0; t0 < ns; t0 += kNW * kUse) {
            if (t0 + wv * kUse - 1 >= ns) break;  // wave-uniform
            const int tt = t0 + wv * kUse + ln - 1;
            const bool live = tt >= 0 && tt < ns && ln >= 1 && ln <= kUse && tt < t0 + kNW * kUse;
            int by, bx0, fy, fx0;
            float sigma;
            strip_geom(min(max(tt, 0), ns - 1), by, bx0, fy, fx0, sigma);
            const float* p = s_buf + by * kBW + bx0;
            auto put = [&](int c, float4 o) {
              if (live) store(bx0, fy, fx0, c, o);
            };
            const bool act = live && !(sigma < kMinSigma);
            const int cnt = __popcll(__ballot(act));
            if (cnt == 0) {
#pragma unroll
              for (int c = 0; c < 3; c++) put(c, lds_load4(p + c * kPlane));
            } else if (cnt >= kDenseItems) {
              epf2_strip<false>(p, fx0, fy, sigma, a, put);
            } else {
              if (!act) {
#pragma unroll
                for (int c = 0; c < 3; c++) put(c, lds_load4(p + c * kPlane));
              } else {
                s_list[atomicAdd(&s_cnt, 1)] = (uint16_t)tt;
              }
            }
          }
          __syncthreads();
          const int cnt = s_cnt;
#pragma unroll 1
          for (int i0 = 0; i0 < cnt; i0 += kNT) {
            if (i0 + (tid & ~63) >= cnt) break;
            const int i = i0 + tid;
            const bool on = i < cnt;
            const int tt = s_list[min(i, cnt - 1)];
            int by, bx0, fy, fx0;
            float sigma;
            strip_geom(tt, by, bx0, fy, fx0, sigma);
            const float* p = s_buf + by * kBW + bx0;
            auto put = [&](int c, float4 o) {
              if (on) store(bx0, fy, fx0, c, o);
            };
            epf2_strip<true>(p, fx0, fy, sigma, a, put, bx0 == 0, bx0 == kBW - 4);
          }
          return;
        } else {
          // item of this lane: 64 consecutive items per wavefront, overlapping its neighbours by one on each side
          auto item = [&](int tl) { return (tl >> 6) * kUse + (tl & 63) - 1; };
          auto geom_of = [&](int tt) -> Geom {
            Geom g;
            g.live = tt >= 0 && tt < n;
            const int tc = min(max(tt, 0), n - 1);
            const int by = kB - margin + (tc / kStrips) * 2;
            g.bx0 = (tc % kStrips) * 4;
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
            if (g.live) store(g.bx0, g.fy + r, g.fx0, c, o);
          };
          const bool owner = ln >= 1 && ln <= kUse;
          const bool wave_has = wv * kUse - 1 < n;
          float4 held[2][3];
          int held_e = -1;  // -1 nothing; a 4x2 item t (both rows); bit 15 set: compacted strip entry t * 2 + r in held[0]
          if constexpr (STAGE == 0) {
            if (wave_has) {
              const Geom g = geom_of(item(tid));
              auto put_g = [&](int r, int c, float4 o) {
                if constexpr (last) {
                  if (owner) put_global(g, r, c, o);
                } else {
                  held[r][c] = o;
                }
              };
#pragma unroll
              for (int c = 0; c < 3; c++) gab_pair(g.p + c * kPlane, c, a.gab_k[c][0], a.gab_k[c][1], a.gab_k[c][2], put_g);
              if (owner && g.live) held_e = item(tid);
            }
          } else {
            auto geom = [&]() -> Geom {
              int tl = tid_kernel;
              asm volatile("" : "+v"(tl));
              return geom_of(item(tl));
            };
            bool dense = false;
            int my_slot = -1;
            if (wave_has) {
              const Geom g = geom();
              const bool mine = owner && g.live;
              const bool act0 = mine && !(g.sigma0 < kMinSigma), act1 = mine && !(g.sigma1 < kMinSigma);
              const unsigned long long m0 = __ballot(act0), m1 = __ballot(act1);
              const int cnt = __popcll(m0) + __popcll(m1);
              dense = STAGE == 1 ? cnt > kSparseMax : cnt > 0;
              if (!dense) {
                int slot = 0, base = 0;
                if (ln == 0) {
                  slot = atomicAdd(&s_nsw, 1);
                  if (cnt) base = atomicAdd(&s_cnt, cnt);
                }
                my_slot = __builtin_amdgcn_readfirstlane(slot);
                base = __builtin_amdgcn_readfirstlane(base);
                const unsigned long long below = (1ull << ln) - 1ull;
                const int it = item(tid);
                if (act0) s_list[base + __popcll(m0 & below)] = (uint16_t)(it * 2);
                if (act1) s_list[base + __popcll(m0) + __popcll(m1 & below)] = (uint16_t)(it * 2 + 1);
                if constexpr (last) {
#pragma unroll
                  for (int r = 0; r < 2; r++) {
                    if ((r ? act1 : act0) || !mine) continue;
#pragma unroll
                    for (int c = 0; c < 3; c++) put_global(g, r, c, lds_load4(g.p + c * kPlane + r * kBW));
                  }
                }
              }
            }
            __syncthreads();  // the list is complete
            if (dense) {
              const Geom g = geom();
              auto put = [&](const Geom& gg, int r, int c, float4 o) {
                if constexpr (last) {
                  if (owner) put_global(gg, r, c, o);
                } else {
                  held[r][c] = o;
                }
              };
              if constexpr (STAGE == 1) epf1_pair(g.p, geom, a, put);
              else epf2_pair(g.p, g.fx0, g.fy, g.sigma0, g.sigma1, a, [&](int r, int c, float4 o) { put(g, r, c, o); });
              if (owner && g.live) held_e = item(tid);
            } else if constexpr (STAGE == 1) {
              const int cnt = s_cnt;
              const int i = my_slot >= 0 ? my_slot * 64 + ln : cnt;
              if (__any(i < cnt)) {
                const bool on = i < cnt;
                const int e = (int)s_list[min(i, max(cnt - 1, 0))];
                const Geom g = geom_of(e >> 1);
                const int r = e & 1;
                epf1_strip_g(g.p + r * kBW, g.fx0, g.fy + r, r ? g.sigma1 : g.sigma0, g.bx0 == 0, g.bx0 == kBW - 4, a,
                             [&](int c, float4 o) {
                               if constexpr (last) {
                                 if (on) put_global(g, r, c, o);
                               } else {
                                 held[0][c] = o;
                               }
                             });
                if (on) held_e = e | 0x8000;
              }
            }
          }
          if constexpr (!last) {
            __syncthreads();  // every read of the stage's input is done
            if (held_e >= 0) {
              const bool strip_e = (held_e & 0x8000) != 0;
              const int tt = strip_e ? (held_e & 0x7fff) >> 1 : held_e, r0 = strip_e ? (held_e & 1) : 0;
              float* d = s_buf + (kB - margin + (tt / kStrips) * 2 + r0) * kBW + (tt % kStrips) * 4;
#pragma unroll
              for (int c = 0; c < 3; c++) lds_store4(d + c * kPlane, held[0][c]);
              if (!strip_e) {
#pragma unroll
                for (int c = 0; c < 3; c++) lds_store4(d + c * kPlane + kBW, held[1][c]);
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
              mirror_fill(s_buf, margin, tx0, ty0, a.w, a.h, tid);
              __syncthreads();
            }
          }
        }
      };
      constexpr int kMg = kBorder - (GAB ? 1 : 0);
      constexpr int kMe1 = kMg - (E1 ? 2 : 0);
      if constexpr (GAB) run_stage(std::integral_constant<int, 0>{}, std::integral_constant<int, kMg>{});
      PROF_MARK(10);
      if constexpr (E1) run_stage(std::integral_constant<int, 1>{}, std::integral_constant<int, kMe1>{});
      PROF_MARK(11);
      if constexpr (E2) run_stage(std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{});
    }
    // ---- (6) the carry moves up
    __syncthreads();
    PROF_MARK(12);
    if (tid == 0) {
      s_cnt = 0;
      s_nsw = 0;
    }
    if (t < t_end) {
      for (int i = tid; i < 3 * kCarry * kStrips; i += kNT) {
        const int c = i / (kCarry * kStrips), r = (i / kStrips) % kCarry, s4 = (i % kStrips) * 4;
        lds_store4(s_buf + c * kPlane + r * kBW + s4, lds_load4(s_save + (c * kCarry + r) * kBW + s4));
      }
    }
    PROF_MARK(13);
  }
#ifdef JXLH_STRIP_PROF
  if (tid_kernel == 0)
    for (int i = 0; i < 16; i++) atomicAdd(&g_strip_prof[i], s_prof[i]);
#endif
}

template <bool GAB, bool E1, bool E2>
void launch_variant(hipStream_t s, const FrameDev& f, const StripArgs& sa) {
  hipLaunchKernelGGL((k123_strip<GAB, E1, E2>), dim3(sa.bands * sa.strips), dim3(kNT), 0, s, f, sa);
}

}  // namespace

int strip_tile_rows(const FrameDev& f) { return (f.yblocks + 7) / 8; }
int strip_strips(const FrameDev& f) { return (f.xblocks + 7) / 8; }
size_t strip_xchg_floats(const FrameDev& f) {
  return (size_t)3 * 2 * strip_strips(f) * ((size_t)strip_tile_rows(f) * kTH) * 4;
}
size_t strip_flag_ints(const FrameDev& f, int bands) { return (size_t)bands * strip_strips(f) + 1; }

// The frame's whole reconstruction chain behind k1_scan (which wrote the block descriptors and tile modes) and, for
// the tiles it left to them, K1's class kernels: planes of the result go to f.tmp (raster).  false = stage list not
// covered (EPF0, or no filter stage at all).
bool launch_strip(hipStream_t s, const FrameDev& f, const uint2* desc, const uint8_t* tile_mode, float* xchg, int* flags,
                  int bands, int* error_flag, float deadline_s) {
  const bool gab = f.gab != 0, e1 = f.epf_iters >= 1, e2 = f.epf_iters >= 2;
  if (f.epf_iters >= 3 || (!gab && !e1)) return false;
  StripArgs sa;
  FusedArgs& a = sa.fa;
  for (int c = 0; c < 3; c++) {
    a.out[c] = f.tmp[c];
    a.scale[c] = f.epf_channel_scale[c];
    for (int k = 0; k < 3; k++) a.gab_k[c][k] = f.gab_k[c][k];
  }
  a.inv_sigma = f.inv_sigma;
  a.stride = (uint32_t)f.plane_stride;
  a.sigma_stride = (uint32_t)f.xblocks;
  a.w = f.xsize;
  a.h = f.ysize;
  a.sm0 = f.epf_sm[0];
  a.bsm0 = f.epf_bsm[0];
  a.sm1 = f.epf_sm[1];
  a.bsm1 = f.epf_bsm[1];
  a.sm2 = f.epf_sm[2];
  a.bsm2 = f.epf_bsm[2];
  sa.desc = desc;
  sa.tile_mode = tile_mode;
  sa.xchg = xchg;
  sa.flags = flags;
  sa.error_flag = error_flag;
  sa.strips = strip_strips(f);
  sa.tile_rows = strip_tile_rows(f);
  sa.bands = bands;
  sa.xchg_rows = sa.tile_rows * kTH;
  sa.deadline_ticks = (unsigned long long)(deadline_s * 1.0e8);
  if (gab && e1 && e2) launch_variant<true, true, true>(s, f, sa);
  else if (gab && e1) launch_variant<true, true, false>(s, f, sa);
  else if (gab) launch_variant<true, false, false>(s, f, sa);
  else if (e1 && e2) launch_variant<false, true, true>(s, f, sa);
  else launch_variant<false, true, false>(s, f, sa);
  return true;
}

}  // namespace jxlh

#ifdef JXLH_STRIP_PROF
extern "C" int jxlh_strip_prof_read(unsigned long long* out, int reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_strip_prof), sizeof(g_strip_prof)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_strip_prof), z, sizeof z) != hipSuccess) return -1;
  }
  return 0;
}
#endif
