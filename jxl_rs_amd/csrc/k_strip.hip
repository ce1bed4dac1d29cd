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
// Which tiles: a tile is transformed here iff every varblock touching it is a DCT with sides <= 32 that lies inside
// one 32x32 quadrant of it (k1_scan decides per tile and writes a descriptor per block: varblocks need not be aligned
// in the format, frame/modular/mod.rs:1061-1064 only confines them to their group, but libjxl's encoder aligns them to
// their own size, which closes every quadrant).  Any other tile is reconstructed by K1's class kernels into `planes` as
// before and merely LOADED into the window here, so a frame may mix both kinds freely.
//
// Status (round 4): bit-identical to the two-kernel path on every test, 2.35 GB of HBM traffic per
// 8K frame instead of 3.58 -- and 1.0-1.14 ms against 0.74-0.80: opt-in (JXLH_FRAME_STRIP).  Both forms are bound by
// instruction issue on MI355X; this one issues 1.3x the instructions on one critical path with two workgroups per CU
// to hide its barriers and exchange latency.
//
// Synchronisation.  Workgroups take a ticket; ticket k = strip k % S of band k / S, so a workgroup's neighbours hold
// adjacent tickets.  A workgroup publishes step q before it waits for its neighbours' step q, hence with R resident
// workgroups the chain of waits is a staircase (ticket R-1-j is held at step j by its unstarted right neighbour at
// worst) and ticket 0 always runs to completion when R exceeds the step count; waits are bounded by a deadline that
// raises JXLH_ERR_DEVICE instead of hanging the device.
//
// Bit-exactness: every arithmetic step is the code K1 and the fused filter kernel run (dequant4t, llf_from_lf,
// idct1d, filters_core.inc) in the same order; only where the operands live differs.
#include "k_vardct_common.h"

#ifndef JXLH_STRIP_WAVES
#define JXLH_STRIP_WAVES 12
#endif
#ifndef JXLH_STRIP_WPE
#define JXLH_STRIP_WPE 6
#endif
// development only: -DJXLH_STRIP_ABLATE=<bits> removes phases (wrong pixels) to attribute time: 1 Gaborish, 2 EPF1, 4 EPF2,
// 8 dequantisation, 16 IDCT passes, 32 exchange, 64 task lists + LLF, 128 save / restore of the carry
#ifndef JXLH_STRIP_ABLATE
#define JXLH_STRIP_ABLATE 0
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
__device__ unsigned long long g_strip_end[2048];  // per ticket: end tick, then XCC id / CU id
__device__ unsigned long long g_strip_prof[20];  // [16] min start, [17] max start, [18] min end, [19] max end
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
  int skew_ticks;                     // the second half of the bands starts this much later (see launch_strip)
};

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

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
// Visits only the out-of-frame columns (all region rows) and the out-of-frame rows (all region columns): a strip on
// the left / right frame edge fills 4 columns per stage, not the window.
__device__ __forceinline__ void mirror_fill(float* __restrict__ buf, int m, int tx0, int ty0, int w, int h, int tid) {
  const int r0 = kB - m, rw = kTW + 2 * m, rh = kTH + 2 * m;  // region origin (both axes) and size
  const int wx0 = tx0 - kB, wy0 = ty0 - kB;                   // frame coordinates of window (0, 0)
  auto fill = [&](int bx, int by) {
    const int fx = wx0 + bx, fy = wy0 + by;
    if (fx >= 0 && fx < w && fy >= 0 && fy < h) return;
    const int sx = mirror(fx, w) - wx0, sy = mirror(fy, h) - wy0;
    if (sx < 0 || sx >= kBW || sy < 0 || sy >= kBH) return;  // outside this window: never consumed
#pragma unroll
    for (int c = 0; c < 3; c++) buf[c * kPlane + by * kBW + bx] = buf[c * kPlane + sy * kBW + sx];
  };
  // out-of-frame columns: [r0, r0 + nl) on the left, [r0 + rw - nr, r0 + rw) on the right
  const int nl = min(rw, max(0, -(wx0 + r0))), nr = min(rw - nl, max(0, wx0 + r0 + rw - w));
  for (int idx = tid; idx < (nl + nr) * rh; idx += kNT) {
    const int j = idx % (nl + nr), by = r0 + idx / (nl + nr);
    fill(j < nl ? r0 + j : r0 + rw - nr + (j - nl), by);
  }
  const int nt = min(rh, max(0, -(wy0 + r0))), nbm = min(rh - nt, max(0, wy0 + r0 + rh - h));
  for (int idx = tid; idx < (nt + nbm) * rw; idx += kNT) {
    const int j = idx / rw, bx = r0 + idx % rw;
    fill(bx, j < nt ? r0 + j : r0 + rh - nbm + (j - nt));
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
  __shared__ uint16_t s_task[2][896];  // pass 1 / pass 2 tasks of the tile by length class: [0, 512) 8, [512, 768) 16, [768, 896) 32
  __shared__ int s_ntask[2][3];        // ... and their counts
  __shared__ uint32_t s_bk[64];        // per block of the tile, see (0)
  __shared__ int s_bt[64], s_bc[64];
  __shared__ float s_sdy[64];
  __shared__ AdjTable s_adj;           // adjust_quant_bias of +i per channel, see dequant4t
  __shared__ int s_toff[JXLH_NUM_QUANT_TABLES];  // FrameDev::table_offset (a dynamically indexed kernel argument would live in scratch)
  __shared__ float s_lf[2][192];  // the tile's LF samples [channel][block row][block column]; this step's / the next's
  __shared__ uint32_t s_ndesc[64], s_nrq[64];  // the next tile's block descriptors as fetched
  __shared__ int s_cnt, s_nsw, s_ticket, s_abort, s_pub;
#ifdef JXLH_STRIP_PROF
  __shared__ unsigned long long s_prof[16], s_prof_t;
  if (threadIdx.x == 0) {
    for (int i = 0; i < 16; i++) s_prof[i] = 0;
    s_prof_t = now_ticks();
    atomicMin(&g_strip_prof[16], s_prof_t);
    atomicMax(&g_strip_prof[17], s_prof_t);
  }
#endif
  const FusedArgs& a = sa.fa;
  const int tid_kernel = threadIdx.x;
  constexpr int kBorder = (GAB ? 1 : 0) + (E1 ? 2 : 0) + (E2 ? 1 : 0);
  static_assert(kBorder >= 1 && kBorder <= kB, "at least one stage");

  build_adj_table(f, &s_adj, tid_kernel, kNT);
  if (tid_kernel == 0) {
#pragma unroll
    for (int i = 0; i < JXLH_NUM_QUANT_TABLES; i++) s_toff[i] = f.table_offset[i];
    s_ticket = atomicAdd(&sa.flags[sa.bands * sa.strips], 1);
    s_cnt = 0;
    s_nsw = 0;
    s_abort = 0;
  }
  __syncthreads();
  const int ticket = __builtin_amdgcn_readfirstlane(s_ticket);  // uniform: everything derived from it lives in SGPRs
  const int band = ticket / sa.strips, strip = ticket % sa.strips;
  if (band >= sa.bands) return;
  if (sa.skew_ticks > 0 && band >= (sa.bands + 1) / 2) {
    // The two workgroups of a CU were dispatched half a grid apart, i.e. belong to bands b and b + bands / 2: started
    // together they would sit in the same phase (both fetching coefficients, then both filtering) for the whole
    // kernel; shifted by half a step one transforms while the other filters.
    const unsigned long long t0 = now_ticks();
    while (now_ticks() - t0 < (unsigned long long)sa.skew_ticks) __builtin_amdgcn_s_sleep(8);
  }
  const int tr0 = (int)((long)band * sa.tile_rows / sa.bands), tr1 = (int)((long)(band + 1) * sa.tile_rows / sa.bands);
  const int t_begin = tr0 > 0 ? tr0 - 1 : 0;
  const int y_lo = tr0 * kTH, y_hi = min(tr1 * kTH, a.h);
  // the step below the band exists for output rows [64 tr1 - 4, 64 tr1)
  const int t_end = (tr1 * kTH - kB < y_hi) ? tr1 : tr1 - 1;
  const int tx0 = strip * kTW;
  int* my_flag = sa.flags + band * sa.strips + strip;
  const int gx = strip / 4;  // group column (256 = 4 tiles)

  // What a step's transform phase needs from global memory before it can address anything -- the tile's mode, its 64
  // block descriptors, its 3 x 64 LF samples -- is fetched one step ahead, straight into LDS (`global_load_lds`: no
  // register holds the value in flight, so nothing waits for it until the next step reads it).  Blocks outside the frame
  // fetch a clamped address and are masked when decoded.
  int nx_mode = 1;
  auto prefetch_meta = [&](int tt, int tid) {
    if (tt >= sa.tile_rows || tt > t_end) return;
    nx_mode = sa.tile_mode[tt * sa.strips + strip];
    if (tid < 192) {
      const int c = __builtin_amdgcn_readfirstlane(tid >> 6);
      const int bi = tid & 63, gbx = min(strip * 8 + (bi & 7), f.xblocks - 1), gby = min(tt * 8 + (bi >> 3), f.yblocks - 1);
      const size_t at = (size_t)gby * f.xblocks + gbx;
      if (c == 0) {
        __builtin_amdgcn_global_load_lds((gptr_t)&sa.desc[at].x, (lptr_t)s_ndesc, 4, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)&sa.desc[at].y, (lptr_t)s_nrq, 4, 0, 0);
      }
      const float* lfp = c == 0 ? f.lf[0] : c == 1 ? f.lf[1] : f.lf[2];
      __builtin_amdgcn_global_load_lds((gptr_t)(lfp + at), (lptr_t)(s_lf[(tt - t_begin) & 1] + c * 64), 4, 0, 0);
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
      // ---- (0) what the tile's 64 blocks need decoded once: where the varblock of a block starts in the window, which
      // 64-coefficient piece of it the block stands for, its shape, its coefficients and its dequantisation weights
      if (tid < 192) __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wavefront's prefetches have landed
      if (tid < 64) {
        const int bx = tid & 7, by = tid >> 3;
        const bool in = strip * 8 + bx < f.xblocks && t * 8 + by < f.yblocks;
        const Blk b = decode_desc(in ? s_ndesc[tid] : 0u, bx, by);
        const int org = (kCarry + 8 * (by - b.dy)) * kBW + kB + 8 * (bx - b.dx);
        s_bk[tid] = !b.on ? 0u
                          : (uint32_t)org | (uint32_t)((b.dy << b.lcx) + b.dx) << 13 | (uint32_t)b.lcy << 17 |
                                (uint32_t)b.lcx << 19 | (uint32_t)b.type << 21 | (b.dx == 0 ? 1u << 26 : 0u) |
                                (b.dy == 0 ? 1u << 27 : 0u) | 1u << 31;
        // quant_table_for_type for the nine small DCTs: 0, 4, 5 -> themselves; 6, 7 -> 6; 8, 9 -> 7; 10, 11 -> 8
        s_bt[tid] = b.on ? s_toff[b.type < 6 ? b.type : 6 + ((b.type - 6) >> 1)] : 0;
        s_bc[tid] = b.off64 * 64;
        s_sdy[tid] = f.inv_global_scale / (float)s_nrq[tid];  // group.rs:153
      }
      const float* lf_tile = s_lf[(t - t_begin) & 1];
      if (tid == 0) s_pub = 0;
      if (tid >= 64 && tid < 70) s_ntask[(tid - 64) / 3][(tid - 64) % 3] = 0;
      __syncthreads();
      prefetch_meta(t + 1, tid);
      PROF_MARK(0);
      // wavefront (q, ch) = (side / row quarter of the exchange, channel) in the edge-column exchange below
      const int q = wave / 3, ch = wave % 3;
      if (mode == 0) {
        if (JXLH_STRIP_ABLATE & 64) {
        } else if (wave < 8) {
          // ---- (1a) the tasks of the two IDCT passes, tile-wide and the same for every channel, by length: [0, 512) 8,
          // [512, 768) 16, [768, 896) 32.  Wavefront w: pass 1 candidates (row y of block row w, block column): a task
          // iff the block is the leftmost of its varblock; pass 2 (column x of block column w, block row): iff it is
          // the topmost.  Pooled over the whole tile and (below) the three channels, so that the rounds of the long
          // transforms run with full lanes: per-quadrant lists left them a quarter full (99 M of the kernel's 307 M
          // vector instructions per 8K frame were these passes).
          {
            const int bxc = lane & 7, y = wave * 8 + (lane >> 3);
            const uint32_t bk = s_bk[wave * 8 + bxc];
            const bool act = (bk >> 26) & 1u;
            const int lcx = (bk >> 19) & 3;
#pragma unroll
            for (int k = 0; k < 3; k++) {
              const unsigned long long m = __ballot(act && lcx == k);
              if (m) {
                int base = 0;
                if (lane == 0) base = atomicAdd(&s_ntask[0][k], __popcll(m));
                base = __builtin_amdgcn_readfirstlane(base);
                if (act && lcx == k)
                  s_task[0][(k == 0 ? 0 : k == 1 ? 512 : 768) + base + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)(y << 3 | bxc);
              }
            }
          }
          {
            const int byc = lane >> 3, x = wave * 8 + (lane & 7);
            const uint32_t bk = s_bk[byc * 8 + wave];
            const bool act = (bk >> 27) & 1u;
            const int lcy = (bk >> 17) & 3;
#pragma unroll
            for (int k = 0; k < 3; k++) {
              const unsigned long long m = __ballot(act && lcy == k);
              if (m) {
                int base = 0;
                if (lane == 0) base = atomicAdd(&s_ntask[1][k], __popcll(m));
                base = __builtin_amdgcn_readfirstlane(base);
                if (act && lcy == k)
                  s_task[1][(k == 0 ? 0 : k == 1 ? 512 : 768) + base + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)(x << 3 | byc);
              }
            }
          }
        } else if (wave < 11) {
          // ---- (1c) LLF-from-LF of channel wave - 8 (one lane per block; the first block of a varblock acts) over the
          // lowest frequencies; the dequantisation skips that corner, which the LLF overwrites in the reference
          // (transform.rs:450)
          const int c = wave - 8;
          const uint32_t bk = s_bk[lane];
          if ((bk >> 26 & 3u) == 3u) {
            const float* lf = lf_tile + c * 64 + lane;
            float* org = s_buf + c * kPlane + (bk & 0x1fffu);
            switch ((bk >> 21) & 31u) {
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
        PROF_MARK(1);
        // ---- (1b) dequantisation + chroma-from-luma straight into the window (dequant_block, group.rs:137-177): a lane
        // takes 4 consecutive stored coefficients of the three channels; 1024 chunks, the wavefronts that built no task
        // list and ran no LLF take a second one
        {
          const int g = (t / 4) * f.xgroups + gx;
          const int cti = t * f.cmap_stride + strip;  // the tile IS a colour tile (64 x 64)
          const float x_cc = f.base_x + (float)f.ytox[cti] / f.color_factor;  // color_correlation_map.rs:76-78
          const float b_cc = f.base_b + (float)f.ytob[cti] / f.color_factor;
#pragma unroll 1
          for (int it = 0; it < ((JXLH_STRIP_ABLATE & 8) ? 0 : 2); it++) {
            if (it == 1 && wave < 8) break;  // wave-uniform
            const int idx = it == 0 ? tid : 768 + (tid - 512);
            const int bi = (idx >> 4) & 63, qd = idx & 15;
            const uint32_t bk = s_bk[bi];
            if (idx >= 1024 || !(bk >> 31)) continue;
            const int k = (int)((bk >> 13) & 15u) * 64 + 4 * qd;  // index inside the varblock's stored coefficients
            const int lr = 3 + (int)((bk >> 17) & 3u), lc = 3 + (int)((bk >> 19) & 3u);  // log2 of R, C
            const bool wide = lr < lc;
            // stored in[u * R + v] for R >= C, in[v * C + u] for the wide shapes (tests.rs:119-132)
            const int u = wide ? (k & ((1 << lc) - 1)) : (k >> lr), v = wide ? (k >> lc) : (k & ((1 << lr) - 1));
            // leading elements inside the LLF corner (u < cx, v < cy): written by the LLF lanes instead
            const int sk = wide ? ((u == 0 && v < (1 << (lr - 3))) ? (1 << (lc - 3)) : 0) : ((v == 0 && u < (1 << (lc - 3))) ? (1 << (lr - 3)) : 0);
            const int tsize = 1 << (lr + lc);
            const float* tb = f.tables + s_bt[bi] + k;
            const int* cf = f.coeffs + ((size_t)g * 3 * kGroupArea + s_bc[bi] + k);
            const int4 q1 = gload_i4<true>(cf + kGroupArea), q0 = gload_i4<true>(cf), q2 = gload_i4<true>(cf + 2 * kGroupArea);
            const float4 t1 = *reinterpret_cast<const float4*>(tb + tsize), t0 = *reinterpret_cast<const float4*>(tb),
                         t2 = *reinterpret_cast<const float4*>(tb + 2 * tsize);
            const float sd = s_sdy[bi];
            float dy[4];
            BlockInfo binf;
            binf.sdy = sd;
            binf.x_cc = x_cc;
            binf.b_cc = b_cc;
            // channel order of the reference: Y, X, B
            const float4 vy = dequant4t<1>(f, q1, t1, binf, &s_adj, dy);
            const float4 vx = dequant4t<0>(f, q0, t0, binf, &s_adj, dy);
            const float4 vb = dequant4t<2>(f, q2, t2, binf, &s_adj, dy);
            float* d = s_buf + (bk & 0x1fffu) + v * kBW + u;
            const float4 vv[3] = {vx, vy, vb};
#pragma unroll
            for (int c = 0; c < 3; c++) {
              float* dc = d + c * kPlane;
              if (wide && sk == 0) {
                lds_store4(dc, vv[c]);
              } else {
                const int st = wide ? 1 : kBW;
                if (sk < 1) dc[0] = vv[c].x;
                if (sk < 2) dc[st] = vv[c].y;
                if (sk < 3) dc[2 * st] = vv[c].z;
                if (sk < 4) dc[3 * st] = vv[c].w;
              }
            }
          }
        }
        __syncthreads();
        PROF_MARK(2);
        // ---- (1d) pass 1 (along u: window rows), (1e) pass 2 (along v: window columns); idct2d.rs:111-131 order.
        // Batches of 64 tasks of one length over (channel, list entry), the long ones first, batch b on wavefront b % 12.
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
          const int m8 = s_ntask[pass][0], m16 = s_ntask[pass][1], m32 = s_ntask[pass][2];
          const int b32 = (3 * m32 + 63) >> 6, b16 = (3 * m16 + 63) >> 6, b8 = (3 * m8 + 63) >> 6;
          const int nb = (JXLH_STRIP_ABLATE & 16) ? 0 : b32 + b16 + b8;
          for (int bt = wave; bt < nb; bt += kNW) {
            const int cls = bt < b32 ? 2 : bt < b32 + b16 ? 1 : 0;  // wave-uniform
            const int per = cls == 2 ? m32 : cls == 1 ? m16 : m8;
            const int i = (bt - (cls == 2 ? 0 : cls == 1 ? b32 : b32 + b16)) * 64 + lane;
            if (i < 3 * per) {
              const int c = (i >= per ? 1 : 0) + (i >= 2 * per ? 1 : 0);
              const int e = s_task[pass][(cls == 0 ? 0 : cls == 1 ? 512 : 768) + i - c * per];
              float* p = s_buf + c * kPlane + kCarry * kBW + kB +
                         (pass == 0 ? (e >> 3) * kBW + (e & 7) * 8 : (e & 7) * 8 * kBW + (e >> 3));
              if (pass == 0) {
                if (cls == 2) idct_line<32, 1>(p);
                else if (cls == 1) idct_line<16, 1>(p);
                else idct_line<8, 1>(p);
              } else {
                if (cls == 2) idct_line<32, kBW>(p);
                else if (cls == 1) idct_line<16, kBW>(p);
                else idct_line<8, kBW>(p);
              }
            }
          }
          __syncthreads();
          PROF_MARK(3 + pass);
        }
      } else {
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
      if (sa.strips > 1 && wave < 12 && !(JXLH_STRIP_ABLATE & 32)) {
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
        // (Relaxed on purpose.  The edge columns are themselves agent-scope atomics -- performed at the device's
        // coherence point, acknowledged by the vmcnt wait above -- and the twelve wavefronts meet on the LDS counter
        // before the flag goes up.  A release store / acquire load at agent scope costs a write-back / invalidate of
        // the XCD's L2 per step and spin iteration: measured round 5, 1.0 -> 5.6 ms per 8K frame.)
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
    if (t < t_end && !(JXLH_STRIP_ABLATE & 128)) {
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
      if constexpr (GAB && !(JXLH_STRIP_ABLATE & 1)) run_stage(std::integral_constant<int, 0>{}, std::integral_constant<int, kMg>{});
      PROF_MARK(10);
      if constexpr (E1 && !(JXLH_STRIP_ABLATE & 2)) run_stage(std::integral_constant<int, 1>{}, std::integral_constant<int, kMe1>{});
      PROF_MARK(11);
      if constexpr (E2 && !(JXLH_STRIP_ABLATE & 4)) run_stage(std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{});
    }
    // ---- (6) the carry moves up
    __syncthreads();
    PROF_MARK(12);
    if (tid == 0) {
      s_cnt = 0;
      s_nsw = 0;
    }
    if (t < t_end && !(JXLH_STRIP_ABLATE & 128)) {
      for (int i = tid; i < 3 * kCarry * kStrips; i += kNT) {
        const int c = i / (kCarry * kStrips), r = (i / kStrips) % kCarry, s4 = (i % kStrips) * 4;
        lds_store4(s_buf + c * kPlane + r * kBW + s4, lds_load4(s_save + (c * kCarry + r) * kBW + s4));
      }
    }
    PROF_MARK(13);
  }
#ifdef JXLH_STRIP_PROF
  if (tid_kernel == 0) {
    for (int i = 0; i < 16; i++) atomicAdd(&g_strip_prof[i], s_prof[i]);
    const unsigned long long e = now_ticks();
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (s_ticket < 1024) {
      g_strip_end[s_ticket] = e;
      g_strip_end[1024 + s_ticket] = (unsigned long long)hwid | (unsigned long long)xcc << 32;
    }
    atomicMin(&g_strip_prof[18], e);
    atomicMax(&g_strip_prof[19], e);
  }
#endif
}

template <bool GAB, bool E1, bool E2>
void launch_variant(hipStream_t s, const FrameDev& f, const StripArgs& sa) {
  hipLaunchKernelGGL((k123_strip<GAB, E1, E2>), dim3(sa.bands * sa.strips), dim3(kNT), 0, s, f, sa);
}

}  // namespace

// Workgroups of the strip kernel the device keeps resident at once (every variant has the same launch bounds and LDS
// window).  The strips of a band wait for their neighbours' progress flags: a launch is only safe while ALL of its
// workgroups are resident, so the caller sizes bands * strips by this (0 = query failed).
int strip_resident_workgroups(int cu_count) {
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k123_strip<true, true, true>, kNT, 0) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return per_cu * cu_count;
}

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
  static const int skew = [] {
    const char* e = getenv("JXLH_STRIP_SKEW_US");
    return e ? atoi(e) * 100 : 0;
  }();
  sa.skew_ticks = bands > 1 ? skew : 0;
  if (gab && e1 && e2) launch_variant<true, true, true>(s, f, sa);
  else if (gab && e1) launch_variant<true, true, false>(s, f, sa);
  else if (gab) launch_variant<true, false, false>(s, f, sa);
  else if (e1 && e2) launch_variant<false, true, true>(s, f, sa);
  else launch_variant<false, true, false>(s, f, sa);
  return true;
}

}  // namespace jxlh

#ifdef JXLH_STRIP_PROF
extern "C" int jxlh_strip_end_read(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_strip_end), sizeof(g_strip_end)) == hipSuccess ? 0 : -1;
}
extern "C" int jxlh_strip_prof_read(unsigned long long* out, int reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_strip_prof), sizeof(g_strip_prof)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[20] = {0};
    z[16] = z[18] = ~0ull;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_strip_prof), z, sizeof z) != hipSuccess) return -1;
  }
  return 0;
}
#endif
