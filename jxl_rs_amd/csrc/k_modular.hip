// K4 RCT, K5 Palette (non-delta), K6 inverse Squeeze -- Modular transforms on whole i32
// planes (the GPU does not tile: neighbour-border plumbing of transforms/step.rs vanishes).
// All arithmetic is wrapping 32-bit, as in the reference's SIMD paths.
//
// Reference: rct.rs:14-157; palette.rs:24-199; squeeze.rs:107-141 (smooth_tendency_impl),
// :171-185 (unsqueeze_impl), :389-437 (hsqueeze), :576-644 (vsqueeze).
//
// Squeeze is a non-associative recurrence along the squeezed axis (the previous output b
// feeds the next tendency), so the only parallelism is across lines: one lane per line,
// loads software-pipelined ahead of the dependent chain.  Vertical steps are naturally
// coalesced (lanes = adjacent columns); horizontal steps walk rows (lanes = adjacent rows)
// and lean on L1/L2 for the 128-byte lines they share across iterations.
#include "jxlh_internal.h"

namespace jxlh {
namespace {

__device__ __forceinline__ int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
__device__ __forceinline__ int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }

template <int OP>
__device__ __forceinline__ void rct_op(int32_t v0, int32_t v1, int32_t v2, int32_t& w0, int32_t& w1, int32_t& w2) {
  w0 = v0;
  w1 = v1;
  w2 = v2;
  if constexpr (OP == 1) {
    w2 = wadd(v2, v0);
  } else if constexpr (OP == 2) {
    w1 = wadd(v1, v0);
  } else if constexpr (OP == 3) {
    w1 = wadd(v1, v0);
    w2 = wadd(v2, v0);
  } else if constexpr (OP == 4) {
    w1 = wadd(v1, wadd(v0, v2) >> 1);
  } else if constexpr (OP == 5) {
    const int32_t t2 = wadd(v0, v2);
    w1 = wadd(v1, wadd(v0, t2) >> 1);
    w2 = t2;
  } else if constexpr (OP == 6) {
    int32_t y = wsub(v0, v2 >> 1);
    const int32_t g = wadd(v2, y);
    y = wsub(y, v1 >> 1);
    w0 = wadd(y, v1);
    w1 = g;
    w2 = y;
  }
}

// perm: which output plane receives w0/w1/w2 (rct.rs:132-156)
template <int OP>
__global__ void k4_rct(int32_t* __restrict__ p0, int32_t* __restrict__ p1, int32_t* __restrict__ p2, size_t n,
                       int perm, size_t nvec) {
  int32_t* o[3];
  switch (perm) {
    default:
    case 0: o[0] = p0; o[1] = p1; o[2] = p2; break;
    case 1: o[0] = p1; o[1] = p2; o[2] = p0; break;  // Gbr: out[1,2,0] = in[0,1,2]
    case 2: o[0] = p2; o[1] = p0; o[2] = p1; break;  // Brg
    case 3: o[0] = p0; o[1] = p2; o[2] = p1; break;  // Rbg
    case 4: o[0] = p1; o[1] = p0; o[2] = p2; break;  // Grb
    case 5: o[0] = p2; o[1] = p1; o[2] = p0; break;  // Bgr
  }
  // nvec = n / 4 when the three planes are 16-byte aligned (launch_rct), else 0: everything takes the scalar loop
  const size_t stride = (size_t)gridDim.x * blockDim.x;
#ifndef JXLH_RCT_NT
#define JXLH_RCT_NT true  // streamed once, in place: `nt` on both directions 0.29-0.34 -> 0.254-0.259 ms at 8192^2 x 3 (6.3 TB/s)
#endif
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    const int4 a = gload_i4<JXLH_RCT_NT>(p0 + 4 * i);
    const int4 b = gload_i4<JXLH_RCT_NT>(p1 + 4 * i);
    const int4 c = gload_i4<JXLH_RCT_NT>(p2 + 4 * i);
    int4 x, y, z;
    rct_op<OP>(a.x, b.x, c.x, x.x, y.x, z.x);
    rct_op<OP>(a.y, b.y, c.y, x.y, y.y, z.y);
    rct_op<OP>(a.z, b.z, c.z, x.z, y.z, z.z);
    rct_op<OP>(a.w, b.w, c.w, x.w, y.w, z.w);
    gstore_i4<JXLH_RCT_NT>(o[0] + 4 * i, x);
    gstore_i4<JXLH_RCT_NT>(o[1] + 4 * i, y);
    gstore_i4<JXLH_RCT_NT>(o[2] + 4 * i, z);
  }
  // tail
  for (size_t i = nvec * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    int32_t x, y, z;
    rct_op<OP>(p0[i], p1[i], p2[i], x, y, z);
    o[0][i] = x;
    o[1][i] = y;
    o[2][i] = z;
  }
}

// the same on rows of w samples at a pitch of `stride` samples (padded planes: the padding is not touched); row y of
// the grid's y dimension and beyond, one sample per thread along x
template <int OP>
__global__ void k4_rct_rows(int32_t* __restrict__ p0, int32_t* __restrict__ p1, int32_t* __restrict__ p2, uint32_t w,
                            uint32_t h, size_t stride, int perm) {
  int32_t* o[3];
  switch (perm) {
    default:
    case 0: o[0] = p0; o[1] = p1; o[2] = p2; break;
    case 1: o[0] = p1; o[1] = p2; o[2] = p0; break;
    case 2: o[0] = p2; o[1] = p0; o[2] = p1; break;
    case 3: o[0] = p0; o[1] = p2; o[2] = p1; break;
    case 4: o[0] = p1; o[1] = p0; o[2] = p2; break;
    case 5: o[0] = p2; o[1] = p1; o[2] = p0; break;
  }
  for (uint32_t y = blockIdx.y; y < h; y += gridDim.y) {
    const size_t row = (size_t)y * stride;
    for (uint32_t x = blockIdx.x * blockDim.x + threadIdx.x; x < w; x += gridDim.x * blockDim.x) {
      int32_t a, b, c;
      rct_op<OP>(p0[row + x], p1[row + x], p2[row + x], a, b, c);
      o[0][row + x] = a;
      o[1][row + x] = b;
      o[2][row + x] = c;
    }
  }
}

__constant__ int16_t kDeltaPalette[72][3] = {
#include "delta_palette.inc"
};

// get_palette_value (palette.rs:39-163)
__device__ __forceinline__ int32_t palette_value(const int32_t* __restrict__ palette, size_t pstride, int32_t index,
                                                 int c, int palette_size, int bit_depth) {
  if (index < 0) {
    if (c >= 3) return 0;
    uint32_t i = (uint32_t)(-(index + 1));
    i %= 1 + 2 * (72 - 1);
    int32_t r = kDeltaPalette[(i + 1) >> 1][c];
    if ((i & 1) == 0) r = -r;
    if (bit_depth > 8) r *= 1 << (bit_depth - 8);
    return r;
  }
  uint32_t i = (uint32_t)index;
  const uint32_t ps = (uint32_t)palette_size;
  if (i >= ps && i < ps + 64) {
    if (c >= 3) return 0;
    i -= ps;
    i >>= c * 2;
    const int sh = bit_depth > 3 ? bit_depth - 3 : 0;
    return (int32_t)(((uint64_t)(i % 4) * (uint64_t)((1u << bit_depth) - 1)) >> 2) + (1 << sh);
  } else if (i >= ps + 64) {
    if (c >= 3) return 0;
    i -= ps + 64;
    if (c == 1) i /= 5;
    if (c == 2) i /= 25;
    return (int32_t)(((uint64_t)(i % 5) * (uint64_t)((1u << bit_depth) - 1)) >> 2);
  }
  return palette[(size_t)c * pstride + i];
}

// ---- palette step with delta entries and / or a neighbour predictor (do_palette_step_general, palette.rs:228-251)
// Entries below num_deltas are ADDED to a prediction from already reconstructed neighbours (left, top row up to
// x + 2, the row above that): a raster-order dependency.  Pixel (x, y) can go once (x - 1, y) and (x + 2, y - 1)
// are done, so all pixels with the same x + 3y are independent -- a wavefront of w + 3h steps:
//  * a workgroup owns a band of kDeltaRows rows of one channel, lane l = row; at step s lane l handles x = s - 3l.
//    A row's recent outputs live in an LDS ring of 8 columns (the row below reads columns x - 1 .. x + 2, the one
//    after that column x, while the owner is writing x + 3 / x + 6), its own left / leftleft in registers; one
//    LDS-only barrier per step.
//  * bands are pipelined ACROSS workgroups: band k trails band k - 1 by 3 * kDeltaRows steps.  The producer
//    publishes its completed step count every kDeltaPublish steps (device-scope fence, then one store); the consumer
//    copies the next 64 columns of the two rows above it into LDS every 64 steps, after the counter says they are
//    final, with device-coherent loads (they bypass the CU's L1, which may hold the lines from before they were
//    written).  A waiting band only depends on bands with a lower block index, which were dispatched earlier.
constexpr int kDeltaRows = 256;
constexpr int kDeltaPublish = 32;

template <int predictor>
__device__ __forceinline__ int64_t predict_one(int64_t left, int64_t top, int64_t toptop, int64_t topleft,
                                               int64_t topright, int64_t leftleft, int64_t toprightright) {
  switch (predictor) {  // Predictor::predict_one, modular/predict.rs:152-198 (i64, `/` truncates)
    case 1: return left;
    case 2: return top;
    case 3: return (top + left) / 2;
    case 4: {
      const int64_t p = left + top - topleft;
      const int64_t dl = p - left < 0 ? left - p : p - left, dt = p - top < 0 ? top - p : p - top;
      return dl < dt ? left : top;
    }
    case 5: {
      const int64_t mn = left < top ? left : top, mx = left < top ? top : left;
      const int64_t grad = left + top - topleft;
      const int64_t gmax = topleft < mn ? mx : grad;
      return topleft > mx ? mn : gmax;
    }
    case 7: return topright;
    case 8: return topleft;
    case 9: return leftleft;
    case 10: return (left + topleft) / 2;
    case 11: return (top + topleft) / 2;
    case 12: return (top + topright) / 2;
    case 13: return (6 * top - 2 * toptop + 7 * left + leftleft + toprightright + 3 * topright + 8) / 16;
    default: return 0;
  }
}

// `out` already holds the palette entry of every pixel (the parallel gather kernel ran first): the wavefront only
// adds the prediction where index < num_deltas.
//
// Global memory is touched in CHUNKS of kDeltaChunk steps, never inside a step.  A wave with loads and stores both
// outstanding has to drain everything (one in-order vmcnt, vmcnt(0)) whenever it needs a loaded value, so a per-step
// "load the column 8 steps ahead, store this column" costs a full memory round trip per step, prefetch or not --
// that, not the prediction, was the 0.7-1.6 us step of the first version.  Per chunk a lane now: stores the outputs of
// the previous chunk (LDS -> global), moves the next chunk's index / entry values from registers to LDS (they were
// requested a whole chunk ago) and requests the chunk after that.  Steps read and write LDS only (lane-private rows of
// 33 dwords: conflict-free).  Progress is published one chunk late, right before a chunk's stores are issued, when
// the previous chunk's stores have long been acknowledged: the fence is free.
constexpr int kDeltaChunk = 32;
static_assert(kDeltaPublish == kDeltaChunk, "progress is published per chunk");

template <int PREDICTOR>
__global__ __launch_bounds__(kDeltaRows) void k5_palette_delta(const int32_t* __restrict__ index, int w, int h,
                                                               int num_deltas, int32_t* out_base, int* progress_base) {
  constexpr int C = kDeltaChunk;
  // rows 0 / 1 of the ring are the two rows ABOVE the band (y0 - 2, y0 - 1), fed from s_above one step ahead by lane
  // 0, so that every lane reads its neighbours the same way, unconditionally and in one batch; lane l owns row l + 2
  __shared__ int32_t s_ring[kDeltaRows + 2][9];
  __shared__ int32_t s_idx[kDeltaRows][C + 1], s_ent[kDeltaRows][C + 1], s_outc[kDeltaRows][C + 1];
  __shared__ int32_t s_above[2][128];  // rows y0 - 1 and y0 - 2 (of the previous band), a window of 128 columns
  __shared__ int s_avail;
  const int c = blockIdx.x, band = blockIdx.y, nbands = gridDim.y, l = threadIdx.x;
  int32_t* out = out_base + (size_t)c * (size_t)w * h;  // no __restrict__: rows are read back
  int* progress = progress_base + (size_t)c * nbands;
  const int y0 = band * kDeltaRows;
  const int rows = min(kDeltaRows, h - y0);
  const int y = y0 + l;
  const bool live = l < rows;
  int32_t left_v = 0, leftleft_v = 0;  // out[y][x - 1], out[y][x - 2]
  const int nsteps = w + 3 * (rows - 1);
  const int nchunks = (nsteps + C - 1) / C;
  // steps the previous band (always kDeltaRows rows) takes; its last row finishes column x at step x + 3*(R-1)
  const int prod_steps = w + 3 * (kDeltaRows - 1);
  int avail = 0;  // completed (and stored) steps of the previous band, as last observed
  // Chunk k of row r = columns [k C - 3 r, k C - 3 r + C): 32 contiguous samples.  The chunk is moved TRANSPOSED:
  // thread t handles column t % 32 of rows t / 32 + 8 i, so that a wave instruction touches two 128-byte row segments
  // (a lane fetching its own row's samples costs the texture path one cache-line request per lane: ~0.25 us per
  // instruction, which was the whole step time); the LDS tiles turn rows back into lanes.  Columns outside the row
  // read the nearest valid one, rows below the image read its last row; neither is ever used.
  static_assert(C == 32 && kDeltaRows == 256, "the mover mapping below assumes 32-column chunks and 256-row bands");
  constexpr int NI = 32;  // rows per thread and chunk
  const int mcol = l & 31, mrow0 = l >> 5;
  int32_t nq_i[NI], nq_e[NI];  // the chunk after the current one, in flight
  auto fetch = [&](int k) {
#pragma unroll
    for (int i = 0; i < NI; i++) {
      const int r = mrow0 + 8 * i;
      const int xi = min(max(k * C - 3 * r + mcol, 0), w - 1);
      const size_t off = (size_t)min(y0 + r, h - 1) * w + xi;
      nq_i[i] = index[off];
      nq_e[i] = out[off];
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < NI; i++) {
      s_idx[mrow0 + 8 * i][mcol] = nq_i[i];
      s_ent[mrow0 + 8 * i][mcol] = nq_e[i];
    }
  };
  auto flush = [&](int k) {
#pragma unroll
    for (int i = 0; i < NI; i++) {
      const int r = mrow0 + 8 * i;
      const int x = k * C - 3 * r + mcol;
      // agent-scope (write-through) stores: the band below reads these rows from another XCD, whose L2 is not
      // coherent with this one's.  With plain stores every publish needs a release fence = a write-back of this
      // XCD's whole L2 (measured ~10 us per chunk, 60 % of a band's time); written through, "visible to the
      // agent" is simply "acknowledged", which the publish below waits for with vmcnt(0).
      if (r < rows && x >= 0 && x < w)
        __hip_atomic_store(&out[(size_t)(y0 + r) * w + x], s_outc[r][mcol], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  fetch(0);
  stage();
  if (nchunks > 1) fetch(1);
  __syncthreads();
  for (int k = 0; k < nchunks; k++) {
    const int s0 = k * C;
    if (band > 0 && (s0 & 63) == 0) {
      // columns this band's first rows touch during steps s0 .. s0 + 63: up to s0 + 66 -> copy [lo, hi)
      const int lo = s0 == 0 ? 0 : s0 + 3, hi = min(w, s0 + 67);  // lane 0 copies column s + 3 during step s
      if (lo < hi) {
        const int need = min(prod_steps, (hi - 1) + 3 * (kDeltaRows - 1) + 1);
        if (avail < need) {
          if (l == 0) {
            int v;
            while ((v = __hip_atomic_load(&progress[band - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) < need)
              __builtin_amdgcn_s_sleep(2);
            s_avail = v;
          }
          __syncthreads();
          avail = s_avail;
        }
        for (int t = l; t < 2 * (hi - lo); t += kDeltaRows) {
          const int r = t & 1, col = lo + (t >> 1);
          s_above[r][col & 127] =
              __hip_atomic_load(&out[(size_t)(y0 - 1 - r) * w + col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
      }
    }
    if (band > 0 && s0 == 0) {  // the ring's view of the rows above, for step 0: columns 0..2 / column 0
      if (l < 3) s_ring[1][l] = s_above[0][l];
      if (l == 0) s_ring[0][0] = s_above[1][0];
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    const int send = min(C, nsteps - s0);
    for (int j = 0; j < send; j++) {
      const int s = s0 + j;
      const int x = s - 3 * l;
      // one batch of LDS reads, no branches: values that do not exist (first rows / columns) are read from valid
      // addresses and discarded by the selects below
      const int32_t idx = s_idx[l][j];
      const int32_t ent = s_ent[l][j];
      const int32_t t_m1 = s_ring[l + 1][(x - 1) & 7], t_0 = s_ring[l + 1][x & 7], t_p1 = s_ring[l + 1][(x + 1) & 7];
      const int32_t t_p2 = s_ring[l + 1][(x + 2) & 7], tt_0 = s_ring[l][x & 7];
      const bool active = live && x >= 0 && x < w;
      // PredictionData::get_rows, modular/predict.rs:96-128
      const int64_t left = x > 0 ? left_v : (y > 0 ? t_0 : 0);
      const int64_t top = y > 0 ? t_0 : left;
      const int64_t topleft = (x > 0 && y > 0) ? t_m1 : left;
      const int64_t topright = (x + 1 < w && y > 0) ? t_p1 : top;
      const int64_t leftleft = x > 1 ? leftleft_v : left;
      const int64_t toptop = y > 1 ? tt_0 : top;
      const int64_t toprightright = (x + 2 < w && y > 0) ? t_p2 : topright;
      const int64_t pred = predict_one<PREDICTOR>(left, top, toptop, topleft, topright, leftleft, toprightright);
      const int32_t val = idx < num_deltas ? (int32_t)(uint32_t)(uint64_t)(pred + (int64_t)ent) : ent;
      if (active) {
        s_outc[l][j] = val;
        s_ring[l + 2][x & 7] = val;
        leftleft_v = left_v;
        left_v = val;
      }
      if (band > 0 && l == 0) {  // the rows above, one step ahead (columns past the row's end are never selected)
        s_ring[1][(s + 3) & 7] = s_above[0][(s + 3) & 127];
        s_ring[0][(s + 1) & 7] = s_above[1][(s + 1) & 127];
      }
      // LDS writes of this step visible to the workgroup
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    // chunk boundary: what chunk k - 1 stored a whole chunk ago is what the band below may now read
    if (band + 1 < nbands && k > 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's write-through stores of chunk k - 1 have landed
      __syncthreads();                                  // ... and every other thread's
      if (l == 0) __hip_atomic_store(&progress[band], s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    flush(k);
    if (k + 1 < nchunks) stage();  // chunk k + 1 (every lane is past its last read of chunk k: the step barriers)
    if (k + 2 < nchunks) fetch(k + 2);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the staged chunk is visible; loads stay in flight
  }
  if (band + 1 < nbands) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (l == 0) __hip_atomic_store(&progress[band], nsteps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---- the same step with Predictor::Weighted (do_palette_step_general, palette.rs:200-227): EVERY pixel runs the
// self-correcting predictor (WeightedPredictorState::predict_and_property + update_errors, modular/predict.rs:312-517),
// delta entries are added to its prediction.  The predictor's state is, per pixel, the signed error TE of the final
// prediction and the four sub-predictors' absolute errors E[4]; pixel (x, y) reads TE / E of (x - 1 .. x + 1, y - 1)
// and of (x - 1, y), (x - 2, y).  The reference keeps two rows of them and ADDS E(x, y) into the previous row's slot
// x + 1 (:508-515), so that the "north" sum of the next pixel already contains its west neighbour's errors; here every
// row keeps its own E in the LDS ring next to its outputs and the lane adds its own E(x - 1, y) / E(x - 2, y) from
// registers: err_n = E(x, y-1) + E(x-1, y), err_nw = E(x-1, y-1) + E(x-2, y), err_ne = E(x+1, y-1), with the
// reference's edge rules (pos_nw = max(x-1, 0), pos_ne = min(x+1, w-1) index the SAME summed slots).  Same wavefront
// as k5_palette_delta (x + 3y), same band pipeline; a band's last row also publishes its TE / E row for the band below.
struct WpParams {
  uint32_t w[4];
  int32_t p1c, p2c, p3c[5];
};
__constant__ uint32_t kWpDivLookup[64] = {  // (1 << 24) / (i + 1), predict.rs:206-213
    16777216, 8388608, 5592405, 4194304, 3355443, 2796202, 2396745, 2097152, 1864135, 1677721, 1525201, 1398101,
    1290555,  1198372, 1118481, 1048576, 986895,  932067,  883011,  838860,  798915,  762600,  729444,  699050,
    671088,   645277,  621378,  599186,  578524,  559240,  541200,  524288,  508400,  493447,  479349,  466033,
    453438,   441505,  430185,  419430,  409200,  399457,  390167,  381300,  372827,  364722,  356962,  349525,
    342392,   335544,  328965,  322638,  316551,  310689,  305040,  299593,  294337,  289262,  284359,  279620,
    275036,   270600,  266305,  262144};

// wp_rows: per channel and band, five rows of w ints (TE, E0..E3 of the band's last row).
// Memory movement as in k5_palette_delta: chunks of 32 steps moved as coalesced row segments through LDS, neighbour and
// state reads in one unconditional batch (the rows above the band are fed into ring rows 0 / 1 one step ahead by lane
// 0), outputs and the published state row written through at agent scope.
__global__ __launch_bounds__(kDeltaRows) void k5_palette_wp(const int32_t* __restrict__ index, int w, int h,
                                                            int num_deltas, int32_t* out_base, int* progress_base,
                                                            int32_t* wp_rows_base, const WpParams P) {
  constexpr int C = kDeltaChunk;
  // ring row 0 / 1 = rows y0 - 2 / y0 - 1, lane l owns row l + 2 (s_te / s_e: row 1 = y0 - 1, lane l owns row l + 2)
  __shared__ int32_t s_ring[kDeltaRows + 2][9];
  __shared__ int32_t s_te[kDeltaRows + 2][9];
  __shared__ uint32_t s_e[4][kDeltaRows + 2][9];
  __shared__ int32_t s_idx[kDeltaRows][C + 1], s_ent[kDeltaRows][C + 1], s_outc[kDeltaRows][C + 1];
  __shared__ int32_t s_state[5][C];    // TE, E0..E3 of the band's last row for the current chunk
  __shared__ int32_t s_above[2][128];  // out rows y0 - 1 and y0 - 2 (of the previous band), a window of 128 columns
  __shared__ int32_t s_above_te[128];  // TE and E of row y0 - 1
  __shared__ uint32_t s_above_e[4][128];
  __shared__ int s_avail;
  // the division table in LDS: from constant memory each lookup is a global load on the step's dependent path (two
  // rounds per step, ~0.4 us each -- most of what this kernel's step cost over k5_palette_delta's)
  __shared__ uint32_t s_div[64];
  const int c = blockIdx.x, band = blockIdx.y, nbands = gridDim.y, l = threadIdx.x;
  if (l < 64) s_div[l] = kWpDivLookup[l];
  int32_t* out = out_base + (size_t)c * (size_t)w * h;
  int* progress = progress_base + (size_t)c * nbands;
  int32_t* wp_mine = wp_rows_base + ((size_t)c * nbands + band) * 5 * (size_t)w;        // written for the band below
  const int32_t* wp_prev = wp_rows_base + ((size_t)c * nbands + band - 1) * 5 * (size_t)w;  // read when band > 0
  const int y0 = band * kDeltaRows;
  const int rows = min(kDeltaRows, h - y0);
  const int y = y0 + l;
  const bool live = l < rows;
  const bool publishes = band + 1 < nbands;  // then rows == kDeltaRows and the last row is lane kDeltaRows - 1
  int32_t left_v = 0, te1 = 0;              // out[y][x - 1], TE(x - 1, y)
  uint32_t e1[4] = {0, 0, 0, 0}, e2[4] = {0, 0, 0, 0};  // E(x - 1, y), E(x - 2, y)
  const int nsteps = w + 3 * (rows - 1);
  const int nchunks = (nsteps + C - 1) / C;
  const int prod_steps = w + 3 * (kDeltaRows - 1);
  int avail = 0;
  static_assert(C == 32 && kDeltaRows == 256, "the mover mapping below assumes 32-column chunks and 256-row bands");
  constexpr int NI = 32;
  const int mcol = l & 31, mrow0 = l >> 5;
  int32_t nq_i[NI], nq_e[NI];
  auto fetch = [&](int k) {
#pragma unroll
    for (int i = 0; i < NI; i++) {
      const int r = mrow0 + 8 * i;
      const int xi = min(max(k * C - 3 * r + mcol, 0), w - 1);
      const size_t off = (size_t)min(y0 + r, h - 1) * w + xi;
      nq_i[i] = index[off];
      nq_e[i] = out[off];
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < NI; i++) {
      s_idx[mrow0 + 8 * i][mcol] = nq_i[i];
      s_ent[mrow0 + 8 * i][mcol] = nq_e[i];
    }
  };
  auto flush = [&](int k) {
#pragma unroll
    for (int i = 0; i < NI; i++) {
      const int r = mrow0 + 8 * i;
      const int x = k * C - 3 * r + mcol;
      if (r < rows && x >= 0 && x < w)
        __hip_atomic_store(&out[(size_t)(y0 + r) * w + x], s_outc[r][mcol], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (publishes && l < 5 * C) {  // the last row's predictor state of this chunk: 5 rows x 32 columns
      const int q = l >> 5, x = k * C - 3 * (kDeltaRows - 1) + mcol;
      if (x >= 0 && x < w)
        __hip_atomic_store(&wp_mine[(size_t)q * w + x], s_state[q][mcol], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  fetch(0);
  stage();
  if (nchunks > 1) fetch(1);
  __syncthreads();
  for (int k = 0; k < nchunks; k++) {
    const int s0 = k * C;
    if (band > 0 && (s0 & 63) == 0) {
      const int lo = s0 == 0 ? 0 : s0 + 3, hi = min(w, s0 + 67);  // lane 0 copies column s + 3 during step s
      if (lo < hi) {
        const int need = min(prod_steps, (hi - 1) + 3 * (kDeltaRows - 1) + 1);
        if (avail < need) {
          if (l == 0) {
            int v;
            while ((v = __hip_atomic_load(&progress[band - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) < need)
              __builtin_amdgcn_s_sleep(2);
            s_avail = v;
          }
          __syncthreads();
          avail = s_avail;
        }
        // 2 output rows + TE + 4 E rows of the columns [lo, hi), device-coherent loads
        for (int t = l; t < 7 * (hi - lo); t += kDeltaRows) {
          const int r = t % 7, col = lo + t / 7;
          if (r < 2) {
            s_above[r][col & 127] =
                __hip_atomic_load(&out[(size_t)(y0 - 1 - r) * w + col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } else {
            const int32_t vv = __hip_atomic_load(&wp_prev[(size_t)(r - 2) * w + col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (r == 2) s_above_te[col & 127] = vv;
            else s_above_e[r - 3][col & 127] = (uint32_t)vv;
          }
        }
        __syncthreads();
      }
    }
    if (band > 0 && s0 == 0) {  // the rings' view of the row(s) above, for step 0: columns 0..2 / column 0
      if (l < 3) {
        s_ring[1][l] = s_above[0][l];
        s_te[1][l] = s_above_te[l];
#pragma unroll
        for (int q = 0; q < 4; q++) s_e[q][1][l] = s_above_e[q][l];
      }
      if (l == 0) s_ring[0][0] = s_above[1][0];
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    const int send = min(C, nsteps - s0);
    for (int j = 0; j < send; j++) {
      const int s = s0 + j;
      const int x = s - 3 * l;
      const bool active = live && x >= 0 && x < w;
      // one batch of LDS reads; what does not exist is read from valid addresses and discarded by the selects
      const int32_t idx = s_idx[l][j];
      const int32_t entry = s_ent[l][j];
      const int cm = (x - 1) & 7, c0 = x & 7, cp = (x + 1) & 7;
      const int32_t t_m1 = s_ring[l + 1][cm], t_0 = s_ring[l + 1][c0], t_p1 = s_ring[l + 1][cp], tt_0 = s_ring[l][c0];
      const int32_t ta_m1 = s_te[l + 1][cm], ta_0 = s_te[l + 1][c0], ta_p1 = s_te[l + 1][cp];
      uint32_t ea_m1[4], ea_0[4], ea_p1[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        ea_m1[q] = s_e[q][l + 1][cm];
        ea_0[q] = s_e[q][l + 1][c0];
        ea_p1[q] = s_e[q][l + 1][cp];
      }
      const bool has_top = y > 0;
      // PredictionData::get_rows, modular/predict.rs:96-128
      const int32_t left = x > 0 ? left_v : (has_top ? t_0 : 0);
      const int32_t top = has_top ? t_0 : left;
      const int32_t topleft = (x > 0 && has_top) ? t_m1 : left;
      const int32_t topright = (x + 1 < w && has_top) ? t_p1 : top;
      const int32_t toptop = y > 1 ? tt_0 : top;
      const bool at_right = !(x + 1 < w), at_left = !(x > 0);  // pos_ne == x, pos_nw == x
      // weights from the error sums (:340-377)
      uint32_t wk[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint32_t en = (has_top ? ea_0[q] : 0u) + (x > 0 ? e1[q] : 0u);
        const uint32_t ene = at_right ? en : (has_top ? ea_p1[q] : 0u);
        const uint32_t enw = at_left ? en : (has_top ? ea_m1[q] : 0u) + (x > 1 ? e2[q] : 0u);
        const uint32_t err = en + ene + enw;
        int shift = 63 - __clzll((unsigned long long)err + 1ull) - 5;
        shift = shift < 0 ? 0 : shift;
        wk[q] = 4u + ((P.w[q] * s_div[err >> shift]) >> shift);
      }
      const int64_t te_w = x > 0 ? (int64_t)te1 : 0;
      const int64_t te_n = has_top ? (int64_t)ta_0 : 0;
      const int64_t te_nw = has_top ? (int64_t)(at_left ? ta_0 : ta_m1) : 0;
      const int64_t te_ne = has_top ? (int64_t)(at_right ? ta_0 : ta_p1) : 0;
      const int64_t sum_wn = te_n + te_w;
      const int64_t n = (int64_t)top << 3, wv = (int64_t)left << 3, ne = (int64_t)topright << 3;
      const int64_t nw = (int64_t)topleft << 3, nn = (int64_t)toptop << 3;
      int64_t pk[4];
      pk[0] = wv + ne - n;
      pk[1] = n - (((sum_wn + te_ne) * (int64_t)P.p1c) >> 5);
      pk[2] = wv - (((sum_wn + te_nw) * (int64_t)P.p2c) >> 5);
      pk[3] = n - ((te_nw * (int64_t)P.p3c[0] + te_n * (int64_t)P.p3c[1] + te_ne * (int64_t)P.p3c[2] +
                    (nn - n) * (int64_t)P.p3c[3] + (nw - wv) * (int64_t)P.p3c[4]) >> 5);
      const int log_weight = 63 - __clzll((unsigned long long)wk[0] + wk[1] + wk[2] + wk[3]);
      const int64_t w0s = (int64_t)(wk[0] >> (log_weight - 4)), w1s = (int64_t)(wk[1] >> (log_weight - 4));
      const int64_t w2s = (int64_t)(wk[2] >> (log_weight - 4)), w3s = (int64_t)(wk[3] >> (log_weight - 4));
      const int64_t weight_sum = w0s + w1s + w2s + w3s;
      const int64_t sum = (weight_sum >> 1) - 1 + w0s * pk[0] + w1s * pk[1] + w2s * pk[2] + w3s * pk[3];
      int64_t pred = (sum * (int64_t)s_div[(weight_sum - 1) & 63]) >> 24;
      if (((te_n ^ te_w) | (te_n ^ te_nw)) <= 0) {
        const int64_t mx = max(wv, max(ne, n)), mn = min(wv, min(ne, n));
        pred = max(mn, min(mx, pred));
      }
      const int64_t wp_pred = (pred + 3) >> 3;
      const int32_t val = idx < num_deltas ? (int32_t)(uint32_t)(uint64_t)(wp_pred + (int64_t)entry) : entry;
      // update_errors (:472-517)
      const int64_t v = (int64_t)val << 3;
      const int32_t te = (int32_t)(pred - v);
      uint32_t e[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int64_t dd = pk[q] - v;
        e[q] = (uint32_t)(((dd < 0 ? -dd : dd) + 3) >> 3);
      }
      if (active) {
        s_outc[l][j] = val;
        s_ring[l + 2][c0] = val;
        s_te[l + 2][c0] = te;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          s_e[q][l + 2][c0] = e[q];
          e2[q] = e1[q];
          e1[q] = e[q];
        }
        if (publishes && l == kDeltaRows - 1) {
          s_state[0][j] = te;
#pragma unroll
          for (int q = 0; q < 4; q++) s_state[1 + q][j] = (int32_t)e[q];
        }
        te1 = te;
        left_v = val;
      }
      if (band > 0 && l == 0) {  // the rows above, one step ahead (columns past the row's end are never selected)
        const int ca = (s + 3) & 127, cr = (s + 3) & 7;
        s_ring[1][cr] = s_above[0][ca];
        s_te[1][cr] = s_above_te[ca];
#pragma unroll
        for (int q = 0; q < 4; q++) s_e[q][1][cr] = s_above_e[q][ca];
        s_ring[0][(s + 1) & 7] = s_above[1][(s + 1) & 127];
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (publishes && k > 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's write-through stores of chunk k - 1 have landed
      __syncthreads();
      if (l == 0) __hip_atomic_store(&progress[band], s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    flush(k);
    if (k + 1 < nchunks) stage();
    if (k + 2 < nchunks) fetch(k + 2);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  if (publishes) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (l == 0) __hip_atomic_store(&progress[band], nsteps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// LDS_PAL: the explicit palette (num_colors x nb_channels entries, <= kPalLdsEntries) is staged in LDS
// once per workgroup (persistent grid), so the per-pixel gathers never leave the CU.
constexpr int kPalLdsEntries = 12288;  // 48 KB
template <bool LDS_PAL>
__global__ __launch_bounds__(256) void k5_palette(const int32_t* __restrict__ index, size_t n,
                                                  const int32_t* __restrict__ palette_g, int num_colors, size_t pstride_g,
                                                  int nb_channels, int bit_depth, int32_t* __restrict__ out,
                                                  size_t ostride, size_t nvec) {
  __shared__ int32_t s_pal[LDS_PAL ? kPalLdsEntries : 1];
  const int32_t* palette = palette_g;
  size_t pstride = pstride_g;
  if constexpr (LDS_PAL) {
    for (int i = threadIdx.x; i < num_colors * nb_channels; i += 256)
      s_pal[i] = palette_g[(size_t)(i / num_colors) * pstride_g + (i % num_colors)];
    __syncthreads();
    palette = s_pal;
    pstride = (size_t)num_colors;
  }
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
#ifndef JXLH_PAL_NT
#define JXLH_PAL_NT true  // index read once, planes written once: 0.229 -> 0.224 ms at 8192^2
#endif
    const int4 idx = gload_i4<JXLH_PAL_NT>(index + 4 * i);
    for (int c = 0; c < nb_channels; c++) {
      int4 v;
      v.x = palette_value(palette, pstride, idx.x, c, num_colors, bit_depth);
      v.y = palette_value(palette, pstride, idx.y, c, num_colors, bit_depth);
      v.z = palette_value(palette, pstride, idx.z, c, num_colors, bit_depth);
      v.w = palette_value(palette, pstride, idx.w, c, num_colors, bit_depth);
      if ((ostride & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
        gstore_i4<JXLH_PAL_NT>(out + (size_t)c * ostride + 4 * i, v);
      } else {  // channel planes are only 4-byte aligned
        int32_t* o = out + (size_t)c * ostride + i * 4;
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
      }
    }
  }
  for (size_t i = nvec * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int32_t idx = index[i];
    for (int c = 0; c < nb_channels; c++)
      out[(size_t)c * ostride + i] = palette_value(palette, pstride, idx, c, num_colors, bit_depth);
  }
}

// smooth_tendency_impl (squeeze.rs:107-141), a = prev, b = avg, c = next_avg.
// The reference clamps with two parity tricks:
//     if (x > 2|a-b| + (x & 1)) x = 2|a-b| + 1;      if (x + (x & 1) > 2|b-c|) x = 2|b-c|;
// Both bounds are even, so whatever the parity of x they reduce to x = min(x, 2|a-b| + 1) and
// x = min(x, 2|b-c|) (x even: x > t <=> x >= t + 2; x odd: x > t + 1 <=> x >= t + 3; x == t + 1 is a
// fixed point) -- also for wrapped operands, since t + 1 and x + 1 cannot overflow (t is even, x is a
// quarter of a 31-bit sum).  That turns eight dependent compare/select operations of the serial chain
// into one v_min3_i32; the sign is applied as (x ^ s) - s.
__device__ __forceinline__ int32_t smooth_tendency(int32_t a, int32_t b, int32_t c) {
  const int32_t a_b = wsub(a, b), b_c = wsub(b, c), a_c = wsub(a, c);
  const int32_t abs_a_b = max(a_b, wsub(0, a_b));
  const int32_t abs_b_c = max(b_c, wsub(0, b_c));
  const int32_t abs_a_c = max(a_c, wsub(0, a_c));
  const bool skip = (b_c != 0) && (a_b != 0) && ((a_b ^ b_c) < 0);
  const int32_t abs_a_b_3 = __mulhi(abs_a_b, 0x55555556);
  int32_t x = wadd(wadd(2, abs_a_c), abs_a_b_3) >> 2;
  const int32_t t1 = (int32_t)(((uint32_t)abs_a_b << 1) + 1u);
  const int32_t u = (int32_t)((uint32_t)abs_b_c << 1);
  x = min(min(x, t1), u);
  if (skip) x = 0;
  const int32_t sgn = a_c >> 31;
  return wsub(x ^ sgn, sgn);
}

// unsqueeze_impl (squeeze.rs:171-185) around smooth_tendency, restated for the serial chain.  One wave owns a line, so
// a step costs (instructions on the dependent path) x 8 cycles (tools/valu_latency.hip: 8 cycles between dependent
// VALU instructions of a lone wave, 5 between independent ones); the step is therefore written on the state
//     d = prev_b - avg                      (a_b of smooth_tendency_impl)
// with everything that does not depend on it moved off the path:
//     b_c = avg - next, sm = sign mask of b_c, bc = |b_c|        (per step constants)
//     e   = d with the sign of b_c applied: the tendency is non-zero only for e >= 0 (prev, avg, next monotone:
//           a_b and b_c of one sign -- `skip` of the reference), and then |a_b| = e, |a_c| = e + bc
//     x   = max(0, min3((e + e/3 + bc + 2) >> 2, 2e + 1, 2bc))   (e < 0 makes 2e + 1 negative: the max is the skip)
//     diff = res + sign(b_c) * x
//     h   = diff - trunc(diff / 2) = (diff + 1 + (diff >> 31)) >> 1
//     b = avg - h, a = b + diff (off the path), and the next state d' = b - next = b_c - h.
// Eleven dependent instructions instead of about twenty-four; bit-equal to the form above wherever the reference's
// i32 SIMD arithmetic does not wrap (test_unsqueeze_large_magnitudes: +-2^28).
__device__ __forceinline__ uint32_t xad(uint32_t a, uint32_t b, uint32_t c) { return (a ^ b) + c; }  // v_xad_u32
__device__ __forceinline__ void unsqueeze_step(int32_t avg, int32_t res, int32_t next_avg, int32_t& d, int32_t& a,
                                               int32_t& b) {
  // off the dependent path
  const int32_t b_c = wsub(avg, next_avg);
  const uint32_t sm = (uint32_t)(b_c >> 31), nsm = (uint32_t)b_c >> 31;
  const uint32_t bc = xad((uint32_t)b_c, sm, nsm);
  const uint32_t k2 = bc + 2u, u = bc << 1, rs = (uint32_t)res + nsm;
  // the chain
  const int32_t e = (int32_t)xad((uint32_t)d, sm, nsm);
  const int32_t e3 = __mulhi(e, 0x55555556);
  const int32_t s = (int32_t)((uint32_t)e + (uint32_t)e3 + k2) >> 2;
  const int32_t t1 = (int32_t)(((uint32_t)e << 1) + 1u);
  const int32_t x = max(min(min(s, t1), (int32_t)u), 0);
  const int32_t diff = (int32_t)xad((uint32_t)x, sm, rs);
  const int32_t h = (int32_t)((uint32_t)diff + (uint32_t)(diff >> 31) + 1u) >> 1;
  d = wsub(b_c, h);
  b = wsub(avg, h);
  a = wadd(b, diff);
}

// One lane per line; up to 3 planes (the channels of one squeeze step) per launch via blockIdx.y.
// Element i of line l lives at p[l*line_pitch + i*elem_pitch].  n_res = floor(n_out/2) residuals
// per line, n_avg = n_out - n_res averages.  The recurrence itself is ~100 dependent cycles per
// step; what must not be exposed is memory latency, so the loads of the NEXT block of U steps
// are in flight while the current block runs (register double buffering).  HVEC: horizontal step
// with 16-byte aligned rows -> int4 loads / stores (a lane walks along its row).
struct SqueezePlanes {
  const int32_t* avg[3];
  const int32_t* res[3];
  int32_t* out[3];
};

#ifndef JXLH_SQ_U
#define JXLH_SQ_U 16
#endif
template <bool HVEC>
__global__ __launch_bounds__(64) void k6_unsqueeze(const SqueezePlanes pl, size_t avg_lp, size_t avg_ep, size_t res_lp,
                                                   size_t res_ep, size_t out_lp, size_t out_ep, int n_lines,
                                                   int n_out) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= n_lines) return;
  const int32_t* __restrict__ a = pl.avg[blockIdx.y] + (size_t)l * avg_lp;
  const int32_t* __restrict__ r = pl.res[blockIdx.y] + (size_t)l * res_lp;
  int32_t* __restrict__ o = pl.out[blockIdx.y] + (size_t)l * out_lp;
  const int w = n_out / 2;
  if (w == 0) {  // single output sample (squeeze.rs:468-476, :672-675)
    o[0] = a[0];
    return;
  }
  const bool has_tail = n_out & 1;
  int32_t cur = a[0];
  int32_t d = 0;  // prev - avg; the first `prev` is avg[0] itself (squeeze.rs:411-414, :591-594)
  constexpr int U = JXLH_SQ_U;  // steps whose inputs are requested ahead (two such blocks are in flight)
  // main body: next_avg = avg[i+1] exists for i < w-1 (or i < w with a tail)
  const int n_main = has_tail ? w : w - 1;
  const int n_blocks = n_main / U;
  int32_t na[U], rr[U], nb[U], rb[U];
  auto load_block = [&](int i0, int32_t(&xa)[U], int32_t(&xr)[U]) {
    if constexpr (HVEC) {
      // avg[i0+1 .. i0+U] is misaligned by one element: fetch avg[i0 .. i0+U+3] as int4 and shift
      int32_t t[U + 4];
#pragma unroll
      for (int k = 0; k < U / 4 + 1; k++) {
        const int4 v = *reinterpret_cast<const int4*>(a + i0 + 4 * k);
        t[4 * k] = v.x; t[4 * k + 1] = v.y; t[4 * k + 2] = v.z; t[4 * k + 3] = v.w;
      }
#pragma unroll
      for (int k = 0; k < U; k++) xa[k] = t[k + 1];
#pragma unroll
      for (int k = 0; k < U / 4; k++) {
        const int4 v = *reinterpret_cast<const int4*>(r + i0 + 4 * k);
        xr[4 * k] = v.x; xr[4 * k + 1] = v.y; xr[4 * k + 2] = v.z; xr[4 * k + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int k = 0; k < U; k++) {
        xa[k] = a[(size_t)(i0 + k + 1) * avg_ep];
        xr[k] = r[(size_t)(i0 + k) * res_ep];
      }
    }
  };
  auto run_block = [&](int i0, const int32_t(&xa)[U], const int32_t(&xr)[U]) {
    int32_t va[U], vb[U];
#pragma unroll
    for (int k = 0; k < U; k++) {
      unsqueeze_step(cur, xr[k], xa[k], d, va[k], vb[k]);
      cur = xa[k];
    }
    if constexpr (HVEC) {
#pragma unroll
      for (int k = 0; k < U; k += 2)
        *reinterpret_cast<int4*>(o + 2 * (i0 + k)) = make_int4(va[k], vb[k], va[k + 1], vb[k + 1]);
    } else {
#pragma unroll
      for (int k = 0; k < U; k++) {
        o[(size_t)(2 * (i0 + k)) * out_ep] = va[k];
        o[(size_t)(2 * (i0 + k) + 1) * out_ep] = vb[k];
      }
    }
  };
  // HVEC over-reads avg by up to 3 elements past i0+U: keep the last block(s) for the scalar tail
  const int vec_blocks = HVEC ? max(0, (n_main - 4) / U) : n_blocks;
  int i = 0;
  if (vec_blocks > 0) {
    load_block(0, na, rr);
    int blk = 0;
    for (; blk + 2 <= vec_blocks; blk += 2) {
      load_block((blk + 1) * U, nb, rb);
      run_block(blk * U, na, rr);
      if (blk + 2 < vec_blocks) load_block((blk + 2) * U, na, rr);
      run_block((blk + 1) * U, nb, rb);
    }
    if (blk < vec_blocks) run_block(blk * U, na, rr);
    i = vec_blocks * U;
  }
  for (; i < n_main; i++) {
    const int32_t nxt = a[(size_t)(i + 1) * avg_ep];
    int32_t va, vb;
    unsqueeze_step(cur, r[(size_t)i * res_ep], nxt, d, va, vb);
    o[(size_t)(2 * i) * out_ep] = va;
    o[(size_t)(2 * i + 1) * out_ep] = vb;
    cur = nxt;
  }
  if (!has_tail) {  // last pair: next_avg = avg itself (squeeze.rs:423-433, :608-616)
    int32_t va, vb;
    unsqueeze_step(cur, r[(size_t)(w - 1) * res_ep], cur, d, va, vb);
    o[(size_t)(2 * w - 2) * out_ep] = va;
    o[(size_t)(2 * w - 1) * out_ep] = vb;
  } else {  // odd size: trailing average is copied (squeeze.rs:434-437, :641-643)
    o[(size_t)(2 * w) * out_ep] = cur;
  }
}

}  // namespace

void launch_rct(hipStream_t s, int32_t* p0, int32_t* p1, int32_t* p2, size_t n, int op, int perm) {
  if (n == 0) return;
  const unsigned grid = (unsigned)min((size_t)8192, (n / 4 + 255) / 256 + 1);
  // sub-ranges of planes (band-sharded runs) may start anywhere: vector accesses only for 16-byte aligned planes
  const bool aligned = ((reinterpret_cast<uintptr_t>(p0) | reinterpret_cast<uintptr_t>(p1) | reinterpret_cast<uintptr_t>(p2)) & 15) == 0;
  const size_t nvec = aligned ? n / 4 : 0;
  switch (op) {
    case 0: hipLaunchKernelGGL(k4_rct<0>, dim3(grid), dim3(256), 0, s, p0, p1, p2, n, perm, nvec); break;
    case 1: hipLaunchKernelGGL(k4_rct<1>, dim3(grid), dim3(256), 0, s, p0, p1, p2, n, perm, nvec); break;
    case 2: hipLaunchKernelGGL(k4_rct<2>, dim3(grid), dim3(256), 0, s, p0, p1, p2, n, perm, nvec); break;
    case 3: hipLaunchKernelGGL(k4_rct<3>, dim3(grid), dim3(256), 0, s, p0, p1, p2, n, perm, nvec); break;
    case 4: hipLaunchKernelGGL(k4_rct<4>, dim3(grid), dim3(256), 0, s, p0, p1, p2, n, perm, nvec); break;
    case 5: hipLaunchKernelGGL(k4_rct<5>, dim3(grid), dim3(256), 0, s, p0, p1, p2, n, perm, nvec); break;
    default: hipLaunchKernelGGL(k4_rct<6>, dim3(grid), dim3(256), 0, s, p0, p1, p2, n, perm, nvec); break;
  }
}

void launch_rct_rows(hipStream_t s, int32_t* p0, int32_t* p1, int32_t* p2, uint32_t w, uint32_t h, size_t stride, int op,
                     int perm) {
  if (w == 0 || h == 0) return;
  const dim3 grid(min(64u, (w + 255) / 256), min(h, 16384u));  // ONE launch whatever the row count
  switch (op) {
    case 0: hipLaunchKernelGGL(k4_rct_rows<0>, grid, dim3(256), 0, s, p0, p1, p2, w, h, stride, perm); break;
    case 1: hipLaunchKernelGGL(k4_rct_rows<1>, grid, dim3(256), 0, s, p0, p1, p2, w, h, stride, perm); break;
    case 2: hipLaunchKernelGGL(k4_rct_rows<2>, grid, dim3(256), 0, s, p0, p1, p2, w, h, stride, perm); break;
    case 3: hipLaunchKernelGGL(k4_rct_rows<3>, grid, dim3(256), 0, s, p0, p1, p2, w, h, stride, perm); break;
    case 4: hipLaunchKernelGGL(k4_rct_rows<4>, grid, dim3(256), 0, s, p0, p1, p2, w, h, stride, perm); break;
    case 5: hipLaunchKernelGGL(k4_rct_rows<5>, grid, dim3(256), 0, s, p0, p1, p2, w, h, stride, perm); break;
    default: hipLaunchKernelGGL(k4_rct_rows<6>, grid, dim3(256), 0, s, p0, p1, p2, w, h, stride, perm); break;
  }
}

void launch_palette(hipStream_t s, const int32_t* index, size_t n, const int32_t* palette, int num_colors,
                    size_t palette_stride, int nb_channels, int bit_depth, int32_t* out, size_t out_channel_stride) {
  if (n == 0) return;
  const size_t ostride = out_channel_stride ? out_channel_stride : n;
  const size_t nvec = (reinterpret_cast<uintptr_t>(index) & 15) == 0 ? n / 4 : 0;  // int4 index loads need alignment
  if (num_colors > 0 && (size_t)num_colors * nb_channels <= (size_t)kPalLdsEntries) {
    // persistent grid (the palette is staged once per workgroup): 3 workgroups of 48 KB LDS per CU
    const unsigned grid = (unsigned)min((size_t)768, (n / 4 + 255) / 256 + 1);
    hipLaunchKernelGGL(k5_palette<true>, dim3(grid), dim3(256), 0, s, index, n, palette, num_colors, palette_stride,
                       nb_channels, bit_depth, out, ostride, nvec);
  } else {
    const unsigned grid = (unsigned)min((size_t)8192, (n + 255) / 256);
    hipLaunchKernelGGL(k5_palette<false>, dim3(grid), dim3(256), 0, s, index, n, palette, num_colors, palette_stride,
                       nb_channels, bit_depth, out, ostride, nvec);
  }
}

// progress: nb_channels * palette_delta_bands(h) ints of device scratch
int palette_delta_bands(int h) { return (h + kDeltaRows - 1) / kDeltaRows; }
// Weighted predictor: header = p1c, p2c, p3ca..p3ce, w0..w3; wp_rows: nb_channels * bands * 5 * w ints of scratch
void launch_palette_wp(hipStream_t s, const int32_t* index, int w, int h, const int32_t* palette, int num_colors,
                       int num_deltas, size_t palette_stride, int nb_channels, int bit_depth, const uint32_t header[11],
                       int32_t* out, int* progress, int32_t* wp_rows) {
  if (w <= 0 || h <= 0) return;
  launch_palette(s, index, (size_t)w * h, palette, num_colors + num_deltas, palette_stride, nb_channels, bit_depth, out);
  const int nbands = palette_delta_bands(h);
  (void)hipMemsetAsync(progress, 0, sizeof(int) * (size_t)nb_channels * nbands, s);
  WpParams P;
  P.p1c = (int32_t)header[0];
  P.p2c = (int32_t)header[1];
  for (int i = 0; i < 5; i++) P.p3c[i] = (int32_t)header[2 + i];
  for (int i = 0; i < 4; i++) P.w[i] = header[7 + i];
  hipLaunchKernelGGL(k5_palette_wp, dim3(nb_channels, nbands), dim3(kDeltaRows), 0, s, index, w, h, num_deltas, out,
                     progress, wp_rows, P);
}

void launch_palette_delta(hipStream_t s, const int32_t* index, int w, int h, const int32_t* palette, int num_colors,
                          int num_deltas, size_t palette_stride, int nb_channels, int bit_depth, int predictor,
                          int32_t* out, int* progress) {
  if (w <= 0 || h <= 0) return;
  // every pixel's palette entry, in parallel (get_palette_value with palette_size = num_colors + num_deltas) ...
  launch_palette(s, index, (size_t)w * h, palette, num_colors + num_deltas, palette_stride, nb_channels, bit_depth, out);
  if (num_deltas <= 0 && predictor == 0) return;
  // ... then the wavefront adds the predictions (entries below num_deltas, incl. the implicit negative indices)
  const int nbands = palette_delta_bands(h);
  (void)hipMemsetAsync(progress, 0, sizeof(int) * (size_t)nb_channels * nbands, s);
  const dim3 grid(nb_channels, nbands), block(kDeltaRows);
#define JXLH_DELTA(P) \
  case P: hipLaunchKernelGGL(k5_palette_delta<P>, grid, block, 0, s, index, w, h, num_deltas, out, progress); break
  switch (predictor) {
    JXLH_DELTA(0); JXLH_DELTA(1); JXLH_DELTA(2); JXLH_DELTA(3); JXLH_DELTA(4); JXLH_DELTA(5); JXLH_DELTA(7);
    JXLH_DELTA(8); JXLH_DELTA(9); JXLH_DELTA(10); JXLH_DELTA(11); JXLH_DELTA(12); JXLH_DELTA(13);
    default: break;
  }
#undef JXLH_DELTA
}

namespace {
// ConvertI32ToU8Stage x3 (render/stages/convert.rs:672-691) + interleave: 4 pixels per thread
template <int CH>
__global__ __launch_bounds__(256) void k_i32_to_rgb8(const int32_t* __restrict__ p0, const int32_t* __restrict__ p1,
                                                     const int32_t* __restrict__ p2, size_t stride, int w, int h,
                                                     int32_t mult, int32_t maxv, uint8_t* __restrict__ out,
                                                     size_t out_stride) {
  const int x4 = (blockIdx.x * 256 + threadIdx.x) * 4;
  const int y = blockIdx.y;
  if (x4 >= w || y >= h) return;
  const int32_t* __restrict__ pl[3] = {p0, p1, p2};
  uint32_t q[4][3];
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int32_t v = x4 + i < w ? pl[c][(size_t)y * stride + x4 + i] : 0;
      const int32_t scaled = (int32_t)((uint32_t)v * (uint32_t)mult);
      const int32_t zeroclip = scaled < 0 ? 0 : scaled;
      q[i][c] = (uint32_t)(scaled > maxv ? maxv : zeroclip) & 0xffu;
    }
  uint8_t* o = out + (size_t)y * out_stride + (size_t)x4 * CH;
#pragma unroll
  for (int i = 0; i < 4; i++)
    if (x4 + i < w) {
      o[i * CH] = (uint8_t)q[i][0];
      o[i * CH + 1] = (uint8_t)q[i][1];
      o[i * CH + 2] = (uint8_t)q[i][2];
      if constexpr (CH == 4) o[i * CH + 3] = 255;
    }
}

// ConvertModularToF32Stage, integer samples (convert.rs:488-533)
__global__ void k_modular_to_f32(const int32_t* __restrict__ in, size_t n, float scale, float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (float)in[i] * scale;
}
// ConvertModularToF32Stage, floating-point samples (convert.rs:416-486 int_to_float / int_to_float_generic): a `bits`-bit
// float with `exp_bits` exponent bits stored in an integer -> binary32.  The generic form covers the reference's two
// fast paths as well: binary32 passes through bit for bit, binary16 widens exactly like the hardware conversion
// (signalling NaNs keep their payload here; the reference's f16 SIMD path quiets them).
__global__ void k_float_samples_to_f32(const int32_t* __restrict__ in, size_t n, uint32_t bits, uint32_t exp_bits,
                                       float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int exp_bias = (1 << (exp_bits - 1)) - 1;
  const uint32_t sign_shift = bits - 1, mant_bits = bits - exp_bits - 1, mant_shift = 23 - mant_bits;
  uint32_t f = (uint32_t)in[i];
  const bool signbit = (f >> sign_shift) != 0;
  f &= (sign_shift >= 32 ? 0xffffffffu : (1u << sign_shift) - 1u);
  uint32_t r;
  if (f == 0) {
    r = signbit ? 0x80000000u : 0u;
  } else {
    int exp = (int)(f >> mant_bits);
    uint32_t mantissa = f & ((1u << mant_bits) - 1u);
    if (exp == (1 << exp_bits) - 1) {  // NaN or infinity
      r = (signbit ? 0x80000000u : 0u) | 0xffu << 23 | mantissa << mant_shift;
    } else {
      mantissa <<= mant_shift;
      if (exp == 0 && exp_bits < 8) {  // subnormal: normalise
        while ((mantissa & 0x800000u) == 0) {
          mantissa <<= 1;
          exp -= 1;
        }
        exp += 1;
        mantissa &= 0x7fffffu;  // the leading 1 is implicit now
      }
      exp -= exp_bias;
      exp += 127;
      r = (signbit ? 0x80000000u : 0u) | (uint32_t)exp << 23 | mantissa;
    }
  }
  out[i] = __uint_as_float(r);
}
// ConvertModularXYBToF32Stage (convert.rs:306-343)
__global__ void k_modular_xyb_to_f32(const int32_t* __restrict__ y, const int32_t* __restrict__ x,
                                     const int32_t* __restrict__ b, size_t n, float sx, float sy, float sb,
                                     float* __restrict__ ox, float* __restrict__ oy, float* __restrict__ ob) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float fy = (float)y[i];
  ox[i] = (float)x[i] * sx;
  oy[i] = fy * sy;
  ob[i] = ((float)b[i] + fy) * sb;
}
}  // namespace

void launch_i32_to_rgb8(hipStream_t s, const int32_t* const planes[3], size_t stride, int w, int h, int32_t mult,
                        int32_t maxv, int channels, uint8_t* out, size_t out_stride) {
  if (w <= 0 || h <= 0) return;
  const dim3 grid(((w + 3) / 4 + 255) / 256, h);
  if (channels == 3)
    hipLaunchKernelGGL(k_i32_to_rgb8<3>, grid, dim3(256), 0, s, planes[0], planes[1], planes[2], stride, w, h, mult, maxv,
                       out, out_stride);
  else
    hipLaunchKernelGGL(k_i32_to_rgb8<4>, grid, dim3(256), 0, s, planes[0], planes[1], planes[2], stride, w, h, mult, maxv,
                       out, out_stride);
}
void launch_modular_to_f32(hipStream_t s, const int32_t* in, size_t n, float scale, float* out) {
  if (n) hipLaunchKernelGGL(k_modular_to_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, n, scale, out);
}
void launch_float_samples_to_f32(hipStream_t s, const int32_t* in, size_t n, uint32_t bits, uint32_t exp_bits, float* out) {
  if (n)
    hipLaunchKernelGGL(k_float_samples_to_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, n, bits, exp_bits, out);
}
void launch_modular_xyb_to_f32(hipStream_t s, const int32_t* y, const int32_t* x, const int32_t* b, size_t n,
                               const float scale[3], float* ox, float* oy, float* ob) {
  if (n)
    hipLaunchKernelGGL(k_modular_xyb_to_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, y, x, b, n, scale[0],
                       scale[1], scale[2], ox, oy, ob);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for the wave's outstanding global
// stores (vmcnt(0)); in the mover / chain kernels below nothing in the workgroup ever reads those back.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The same recurrence with the memory traffic taken off the chain wave.  In k6_unsqueeze a lane fetches and stores
// its own line: for a horizontal step that is 64 different cache lines per memory instruction, and in both directions
// the address arithmetic and the memory instructions sit in the one instruction stream whose length is the step time.
// Here a workgroup is one CHAIN wave (lane = line, 64 lines) and three MOVER waves: the movers stream chunks of
// JXLH_SQT_S steps through LDS -- coalesced along whichever axis is contiguous in memory, transposed by the LDS layout
// for the horizontal step (the CPU does this with register transposes, squeeze.rs:249-283) -- double-buffered against
// the chain wave, which touches only LDS: 128-bit reads / writes for the horizontal layout (line pitch = 4 mod 32
// dwords: conflict-free), one dword per lane and step for the vertical one.
#define JXLH_SQT_S 32                    // steps per chunk
#define JXLH_SQT_PI (JXLH_SQT_S + 4)     // line pitch of the input tiles, horizontal layout
#define JXLH_SQT_PO (2 * JXLH_SQT_S + 4) // ... of the output tile
// One workgroup's share of a tiled step: 64 lines of one plane.  Offsets are 32-bit (the launchers keep planes of 2^31
// samples or more on k6_unsqueeze).
struct TiledLines {
  const int32_t* ga;
  const int32_t* gr;
  int32_t* go;
  uint32_t alp, aep, rlp, rep, olp, oep;  // line / element pitches of the average, residual and output planes
  int n_lines, n_out, l0;
  int vec;  // 16-byte accesses are possible: plane bases and the pitches of the non-contiguous axis are multiples of 16
            // bytes, and every byte offset inside a plane fits 32 bits (tiled_vec_ok)
};
// Dataflow form (k6_unsqueeze_flow): the averages of this step are the outputs of the step before it, which is still
// RUNNING in other workgroups of the same launch.  Every 64-line group of a step keeps one progress word = output
// samples complete (stored and acknowledged) along ITS lines; a consumer reads the words of the groups its next chunk
// touches before it requests the chunk.  Producer and consumer sit on different XCDs, whose L2s are not coherent:
// outputs are written through and averages read at agent scope (as in k5_palette_delta), so "acknowledged" is
// "visible" and the words need no fence.
constexpr int kFlowWordStride = 64;  // ints between two progress words: one 256-byte line each (polls spread over channels)
struct FlowLink {
  const int* dep;      // the producing step's progress words for this plane (kFlowWordStride apart); nullptr: complete
  int dep_same_axis;   // its lines run along this step's lines (two steps of one direction in a row): same group index
  int* mine;           // this group's progress word
  int* error;          // JXLH_ERR_DEVICE if a wait outlasts the deadline (the result is then undefined, but the launch ends)
  unsigned long long deadline_ticks;  // s_memrealtime ticks (100 MHz)
  unsigned long long wait_ticks;      // out: time this lane spent polling; polls: loads of a progress word
  unsigned int polls;
  unsigned long long phase[6];  // JXLH_FLOW_EXP & 64: mover time in stage, publish, fetch, peek, store, barrier
};
#ifndef JXLH_FLOW_EXP  // timing experiments only (results may then be undefined): 1 plain stores, 2 plain loads, 4 publish at the end only, 8 no peek, 64 profile rows 5-10 = mover phases, 16 profile row 2 = the chain wave's time at the loop barrier
#define JXLH_FLOW_EXP 0
#endif
template <bool FLOW>
__device__ __forceinline__ int32_t ld_avg(const int32_t* p) {
  if constexpr (FLOW && !(JXLH_FLOW_EXP & 2)) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}
template <bool FLOW>
__device__ __forceinline__ void st_out(int32_t* p, int32_t v) {
  if constexpr (FLOW && !(JXLH_FLOW_EXP & 1)) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}

template <bool HORIZ, bool FLOW>
__device__ __forceinline__ void unsqueeze_tiled_lines(const TiledLines& T, int32_t* __restrict__ s_avg,
                                                      int32_t* __restrict__ s_res, int32_t* __restrict__ s_out,
                                                      FlowLink& F, int* s_pub) {
  constexpr int S = JXLH_SQT_S, PI = JXLH_SQT_PI, PO = JXLH_SQT_PO;
  constexpr int IN_ELEMS = HORIZ ? 64 * PI : 64 * S, OUT_ELEMS = HORIZ ? 64 * PO : 64 * 2 * S;
  const int tid = threadIdx.x;
  const int l0 = T.l0, n_lines = T.n_lines, n_out = T.n_out;
  const int32_t* __restrict__ ga = T.ga;
  const int32_t* __restrict__ gr = T.gr;
  int32_t* __restrict__ go = T.go;
  // w steps produce 2 w samples; an odd line ends with a copied sample.  EVERY step runs through the chunk pipeline
  // (round 2 left the last n % S steps and the closing pair to the chain lane's own global loads: with power-of-two
  // planes that is 31 steps per level whose memory latency sat in the dependent instruction stream, ~10 us per level):
  // the step without a following average (even lines: next_avg = avg itself, squeeze.rs:423-430) is staged with the
  // average index clamped to the last one, and the last chunk may be partial.
  const int w = n_out / 2;
  const bool has_tail = n_out & 1;
  const int n_avg = n_out - w;
  const int n_chunks = (n_out + 2 * S - 1) / (2 * S);   // chunks of 2 S output samples; the last may hold fewer steps
  auto steps_of = [&](int c) { return max(0, min(S, w - c * S)); };
  const bool chain = tid < 64;
  if (chain) __builtin_amdgcn_s_setprio(3);  // the step time IS this wave's issue latency
  const int m = tid - 64;  // mover index 0..191

  // movers: chunk c of the inputs (next_avg = avg[i0 + 1 + k], res[i0 + k]) -> registers -> LDS, in two halves so that
  // all of a chunk's loads are in flight together (and the previous chunk's stores are issued under them)
  // Offsets are 32-bit (the launcher keeps planes of 2^31 samples or more on k6_unsqueeze) and affine in the slot j.
  // Three mover waves: with the chain wave that is one wave per SIMD, and two workgroups share a CU (a fifth wave
  // doubles up on a SIMD and the second workgroup no longer fits: measured, 384 workgroups took two rounds).
  constexpr int NM = 192, NIN = (64 * S + NM - 1) / NM, NOUT = (64 * 2 * S + NM - 1) / NM;
  // horizontal: the element index is the contiguous axis (32 elements of one line per half wave);
  // vertical: the line index is (64 lines of one element row per wave)
  const int in_r0 = HORIZ ? m / S : m % 64, in_k0 = HORIZ ? m % S : m / 64;
  constexpr int IN_DR = HORIZ ? NM / S : 0, IN_DK = HORIZ ? 0 : NM / 64;
  const int out_r0 = HORIZ ? m / (2 * S) : m % 64, out_k0 = HORIZ ? m % (2 * S) : m / 64;
  constexpr int OUT_DR = HORIZ ? NM / (2 * S) : 0, OUT_DK = HORIZ ? 0 : NM / 64;
  const uint32_t alp = T.alp, aep = T.aep, rlp = T.rlp, rep = T.rep, olp = T.olp, oep = T.oep;
  const uint32_t a_off0 = (uint32_t)(l0 + in_r0) * alp + (uint32_t)(1 + in_k0) * aep;
  const uint32_t r_off0 = (uint32_t)(l0 + in_r0) * rlp + (uint32_t)in_k0 * rep;
  const uint32_t o_off0 = (uint32_t)(l0 + out_r0) * olp + (uint32_t)out_k0 * oep;
  const uint32_t a_dj = IN_DR * alp + IN_DK * aep, r_dj = IN_DR * rlp + IN_DK * rep, o_dj = OUT_DR * olp + OUT_DK * oep;
  const int in_lds0 = HORIZ ? in_r0 * PI + in_k0 : in_k0 * 64 + in_r0;
  constexpr int IN_LDS_DJ = HORIZ ? IN_DR * PI : IN_DK * 64;
  const int out_lds0 = HORIZ ? out_r0 * PO + out_k0 : out_k0 * 64 + out_r0;
  constexpr int OUT_LDS_DJ = HORIZ ? OUT_DR * PO : OUT_DK * 64;
  // ---- dataflow form: wait until the producing step has stored the averages [e_lo, e_hi] of this group's lines
  // (each wave that requests averages asks for itself: lane 0 polls, the wave goes on from the reconvergence point)
  int dep_groups_ok = 0, dep_avail = 0;  // wave-uniform: producer groups seen complete for our lines / samples seen
  bool dep_dead = false;
  auto flow_spin = [&](const int* word, int need) -> int {
    int v = need;
    if ((tid & 63) == 0) {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      int spins = 0;
      F.polls++;
      while ((v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < need) {
        F.polls++;
        // back off: hundreds of workgroups of the later levels wait for most of the launch, and their polls all end
        // at the memory channels that hold the words the running levels publish and peek at
#ifndef JXLH_FLOW_NAP
#define JXLH_FLOW_NAP 96
#endif
        if (++spins < 8) __builtin_amdgcn_s_sleep(4);
        else if (spins < 24) __builtin_amdgcn_s_sleep(JXLH_FLOW_NAP < 24 ? JXLH_FLOW_NAP : 24);
        else __builtin_amdgcn_s_sleep(JXLH_FLOW_NAP);
        if ((spins & 255) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > F.deadline_ticks) {
          atomicExch(F.error, JXLH_ERR_DEVICE);
          v = -1;
          break;
        }
      }
      F.wait_ticks += __builtin_amdgcn_s_memrealtime() - t0;
    }
    asm volatile("" ::: "memory");
    return __builtin_amdgcn_readfirstlane(v);
  };
  // The word the NEXT request will ask about is read one iteration ahead (flow_peek, issued behind the iteration's
  // loads and stores): a poll is a round trip to the coherence point, and in front of a chunk's requests it would sit
  // in every iteration's critical sequence (poll, then loads, then the next iteration's staging: measured 5.4 us per
  // chunk instead of 1.7).  A producer that is ahead -- the steady state -- is then seen without waiting.
  int pk_idx = -1, pk_v = 0;  // pk_v: per lane (every lane loads the same word), made uniform where it is used
  auto flow_peek = [&]() {
    if constexpr (FLOW) {
      if (!F.dep || dep_dead || (JXLH_FLOW_EXP & 8)) return;
      pk_idx = F.dep_same_axis ? (l0 >> 6) : dep_groups_ok;
      if constexpr ((JXLH_FLOW_EXP & 32) != 0) pk_v = *(const volatile int*)(F.dep + pk_idx * kFlowWordStride);
      else pk_v = __hip_atomic_load(F.dep + pk_idx * kFlowWordStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  auto flow_wait = [&](int e_lo, int e_hi) {
    if constexpr (FLOW) {
      if (!F.dep || dep_dead) return;
      if (F.dep_same_axis) {
        if (pk_idx >= 0) dep_avail = max(dep_avail, __builtin_amdgcn_readfirstlane(pk_v));
        if (dep_avail < e_hi + 1) {
          dep_avail = flow_spin(F.dep + (l0 >> 6) * kFlowWordStride, e_hi + 1);
          dep_dead = dep_avail < 0;
        }
      } else {
        const int need = min(l0 + 64, n_lines);
        dep_groups_ok = max(dep_groups_ok, e_lo >> 6);  // (a wave that joins late -- the scalar tail after vector chunks)
        while (!dep_dead && dep_groups_ok <= (e_hi >> 6)) {
          if (!(pk_idx == dep_groups_ok && __builtin_amdgcn_readfirstlane(pk_v) >= need))
            dep_dead = flow_spin(F.dep + dep_groups_ok * kFlowWordStride, need) < 0;
          dep_groups_ok++;
        }
      }
      pk_idx = -1;
      // nothing the compiler places behind this point may move in front of the reads of the progress word above -- on the
      // path that saw the producer ahead through the peeked word as well as on the spinning one (the
      // hardware side: the word's value has returned before it is compared, and the data was acknowledged at the
      // coherence point before the producer raised the word)
      __atomic_signal_fence(__ATOMIC_SEQ_CST);
    }
  };
  // ... and report: `done` output samples of every line of the group are stored AND acknowledged (the caller's wave
  // has waited for vmcnt(0)); the last of the three mover waves to say so raises the word
  auto flow_publish = [&](int done) {
    if constexpr (FLOW) {
      if ((tid & 63) == 0 && (atomicAdd(s_pub, 1) % 3) == 2)
        __hip_atomic_store(F.mine, done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  auto fetch_chunk = [&](int c, int32_t(&va)[NIN], int32_t(&vr)[NIN]) {
    flow_wait(c * S, min(c * S + S, n_avg - 1));
    if (c * S + S <= w - 1) {  // every next average and residual of the chunk exists: affine offsets
      const uint32_t ca = a_off0 + (uint32_t)(c * S) * aep, cr = r_off0 + (uint32_t)(c * S) * rep;
#pragma unroll
      for (int j = 0; j < NIN; j++) {
        const bool ok = l0 + in_r0 + j * IN_DR < n_lines && m + j * NM < 64 * S;
        va[j] = ok ? ld_avg<FLOW>(ga + (ca + j * a_dj)) : 0;
        vr[j] = ok ? gr[cr + j * r_dj] : 0;
      }
    } else {  // the line's end: clamp the element indices (next_avg of the last step = the last average)
#pragma unroll
      for (int j = 0; j < NIN; j++) {
        const int row = in_r0 + j * IN_DR, k = in_k0 + j * IN_DK;
        const bool ok = l0 + row < n_lines && m + j * NM < 64 * S && c * S + k < w;
        const int ia = min(c * S + 1 + k, n_avg - 1), ir = min(c * S + k, max(w - 1, 0));
        va[j] = ok ? ld_avg<FLOW>(ga + ((uint32_t)(l0 + row) * alp + (uint32_t)ia * aep)) : 0;
        vr[j] = ok ? gr[(uint32_t)(l0 + row) * rlp + (uint32_t)ir * rep] : 0;
      }
    }
  };
  auto stage_chunk = [&](int c, const int32_t(&va)[NIN], const int32_t(&vr)[NIN]) {
#pragma unroll
    for (int j = 0; j < NIN; j++) {
      if (m + j * NM < 64 * S) {
        s_avg[(c & 1) * IN_ELEMS + in_lds0 + j * IN_LDS_DJ] = va[j];
        s_res[(c & 1) * IN_ELEMS + in_lds0 + j * IN_LDS_DJ] = vr[j];
      }
    }
  };
  auto store_chunk = [&](int c) {
    const int32_t* so = s_out + (c & 1) * OUT_ELEMS;
    const uint32_t co = o_off0 + (uint32_t)(2 * c * S) * oep;
    const int count = min(2 * S, n_out - 2 * c * S);  // samples of the chunk (the last one may be partial)
    int32_t v[NOUT];
#pragma unroll
    for (int j = 0; j < NOUT; j++) v[j] = m + j * NM < 64 * 2 * S ? so[out_lds0 + j * OUT_LDS_DJ] : 0;
#pragma unroll
    for (int j = 0; j < NOUT; j++)
      if (l0 + out_r0 + j * OUT_DR < n_lines && m + j * NM < 64 * 2 * S && out_k0 + j * OUT_DK < count)
        st_out<FLOW>(go + (co + j * o_dj), v[j]);
  };


  // ---- vector movers.  The dword movers above spend ~2.1 us of instruction issue per chunk (44 memory instructions
  // per lane, each behind its own bounds test and address arithmetic: measured per phase, profiles/r05_i_*) against
  // the chain wave's 1.6-1.7 us: every tiled step was bound by its MOVERS.  Where the planes allow 16-byte accesses,
  // the group is complete and every step of the chunk has a following average (all but the last one or two chunks of
  // a line), a chunk moves as 16 + 16 + 16 buffer instructions of 128 bits with offsets affine in the slot, split by
  // role as in k6_unsqueeze_rct: wave 1 loads (only loads outstanding: its wait at the staging is for requests a whole
  // iteration old), waves 2 and 3 store (only stores outstanding: exactly eight per iteration, so "all but the newest
  // eight acknowledged" is a precise vmcnt(8) -- what the dataflow form publishes on).
  const bool vec_ok = T.vec && l0 + 64 <= n_lines;
  const int n_fast = vec_ok && w >= 1 ? (w - 1) / S : 0;  // chunks c with c S + S <= w - 1 (a prefix of the line)
  const bool vloader = tid >= 64 && tid < 128;
  const int ms = tid - 128;  // vector storer index 0..127
  constexpr int kAuxCoherent = 16;  // sc1: agent scope (write-through store / load at the coherence point)
  constexpr int kAuxLd = (FLOW && !(JXLH_FLOW_EXP & 2)) ? kAuxCoherent : 0, kAuxSt = (FLOW && !(JXLH_FLOW_EXP & 1)) ? kAuxCoherent : 0;
  // (the bases are the same for the whole workgroup; said explicitly, or the dataflow kernel -- where they come out of a
  // level table indexed by the ticket -- gets a readfirstlane loop around every buffer instruction)
  auto uniform_ptr = [](const int32_t* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    typedef __attribute__((address_space(1))) int32_t global_i32;
    return (int32_t*)(global_i32*)((uint64_t)hi << 32 | lo);
  };
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(ga + (size_t)l0 * alp), 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(gr + (size_t)l0 * rlp), 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(go + (size_t)l0 * olp), 0, -1, 0x00020000);
  // loader slot j = 0..7: horizontal (line (m >> 3) + 8 j, quad m & 7), vertical (row (m >> 4) + 4 j, quad m & 15)
  const uint32_t vl_a0 = HORIZ ? ((uint32_t)(m >> 3) * alp + 4u * (m & 7)) * 4u : ((uint32_t)(m >> 4) * aep + 4u * (m & 15)) * 4u;
  const uint32_t vl_r0 = HORIZ ? ((uint32_t)(m >> 3) * rlp + 4u * (m & 7)) * 4u : ((uint32_t)(m >> 4) * rep + 4u * (m & 15)) * 4u;
  const uint32_t vl_adj = HORIZ ? 32u * alp : 16u * aep, vl_rdj = HORIZ ? 32u * rlp : 16u * rep;
  const int vl_lds0 = HORIZ ? (m >> 3) * PI + 4 * (m & 7) : (m >> 4) * 64 + 4 * (m & 15);
  constexpr int VL_LDS_DJ = HORIZ ? 8 * PI : 4 * 64;
  // storer slot j = 0..7: horizontal (line (ms >> 4) + 8 j, quad ms & 15), vertical (row (ms >> 4) + 8 j, quad ms & 15)
  const uint32_t vs_o0 = HORIZ ? ((uint32_t)(ms >> 4) * olp + 4u * (ms & 15)) * 4u : ((uint32_t)(ms >> 4) * oep + 4u * (ms & 15)) * 4u;
  const uint32_t vs_odj = HORIZ ? 32u * olp : 32u * oep;
  const int vs_lds0 = HORIZ ? (ms >> 4) * PO + 4 * (ms & 15) : (ms >> 4) * 64 + 4 * (ms & 15);
  constexpr int VS_LDS_DJ = HORIZ ? 8 * PO : 8 * 64;
  jxlh_i32x4 qa[8], qr[8];
  int32_t qt = 0;  // horizontal: the 33rd average of the line (element c S + 32), one line per lane
  auto fetch_fast = [&](int c) {
    flow_wait(c * S, min(c * S + S, n_avg - 1));
    // horizontal: the aligned quads avg[c S + 4 q ..] (next_avg[k] = avg[c S + 1 + k] is one element further: the
    // staging shifts); vertical: rows c S + 1 + row
    const uint32_t sa = HORIZ ? (uint32_t)(c * S) * 4u : (uint32_t)(c * S + 1) * aep * 4u;
    const uint32_t sr = HORIZ ? (uint32_t)(c * S) * 4u : (uint32_t)(c * S) * rep * 4u;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      qa[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, vl_a0 + j * vl_adj, sa, kAuxLd);
      qr[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_r, vl_r0 + j * vl_rdj, sr, 0);
    }
    if constexpr (HORIZ) qt = __builtin_amdgcn_raw_buffer_load_b32(rs_a, (uint32_t)m * alp * 4u, sa + 4u * S, kAuxLd);
  };
  auto stage_fast = [&](int c) {
    int32_t* da = s_avg + (c & 1) * IN_ELEMS + vl_lds0;
    int32_t* dr = s_res + (c & 1) * IN_ELEMS + vl_lds0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      *reinterpret_cast<jxlh_i32x4*>(dr + j * VL_LDS_DJ) = qr[j];
      if constexpr (HORIZ) {
        int32_t* q = da + j * VL_LDS_DJ - 1;  // position k = 4 q + t - 1 of the line
        if ((m & 7) != 0) q[0] = qa[j].x;
        q[1] = qa[j].y;
        q[2] = qa[j].z;
        q[3] = qa[j].w;
      } else {
        *reinterpret_cast<jxlh_i32x4*>(da + j * VL_LDS_DJ) = qa[j];
      }
    }
    if constexpr (HORIZ) s_avg[(c & 1) * IN_ELEMS + m * PI + S - 1] = qt;
  };
  auto store_fast = [&](int c) {
    const int32_t* so = s_out + (c & 1) * OUT_ELEMS + vs_lds0;
    const uint32_t soff = HORIZ ? (uint32_t)(2 * c * S) * 4u : (uint32_t)(2 * c * S) * oep * 4u;
    jxlh_i32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = *reinterpret_cast<const jxlh_i32x4*>(so + j * VS_LDS_DJ);
#pragma unroll
    for (int j = 0; j < 8; j++) __builtin_amdgcn_raw_buffer_store_b128(v[j], rs_o, vs_o0 + j * vs_odj, soff, kAuxSt);
  };

  const int l = l0 + tid;  // chain lanes
  int32_t cur = 0, d = 0;
  if (chain) {
    flow_wait(0, 0);
    if (l < n_lines) cur = ld_avg<FLOW>(ga + (uint32_t)l * alp);
  }
  // mover schedule, iteration c: stage chunk c + 1 (fetched during iteration c - 1: its latency is a whole iteration
  // old), fetch chunk c + 2 into registers, drain the outputs of chunk c - 1
  int32_t pa[NIN], pr[NIN];
  if (!chain && steps_of(0) > 0) {
    if (0 < n_fast) {
      if (vloader) {
        fetch_fast(0);
        stage_fast(0);
      }
    } else {
      fetch_chunk(0, pa, pr);
      stage_chunk(0, pa, pr);
    }
    if (steps_of(1) > 0) {
      if (1 < n_fast) {
        if (vloader) fetch_fast(1);
      } else {
        fetch_chunk(1, pa, pr);
      }
    }
    if (steps_of(2) > 0 && (vloader || 2 >= n_fast)) flow_peek();
  }
  lds_barrier();
  unsigned long long tp = 0;
  auto mark = [&](int ph) {
    if constexpr (FLOW && (JXLH_FLOW_EXP & 64)) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const unsigned long long now = __builtin_amdgcn_s_memrealtime();
      if (ph >= 0) F.phase[ph] += now - tp;
      tp = now;
    }
  };
  // the chain wave's chunk c: LDS in, LDS out
  auto chain_chunk = [&](int c) {
    const int32_t* ia = s_avg + (c & 1) * IN_ELEMS;
    const int32_t* ir = s_res + (c & 1) * IN_ELEMS;
    int32_t* oa = s_out + (c & 1) * OUT_ELEMS;
    const int sc = steps_of(c);
    if (sc == S) {
      int32_t xa[S], xr[S];
      if constexpr (HORIZ) {
#pragma unroll
        for (int j = 0; j < S / 4; j++) {
          const int4 va = *reinterpret_cast<const int4*>(ia + tid * PI + 4 * j);
          const int4 vr = *reinterpret_cast<const int4*>(ir + tid * PI + 4 * j);
          xa[4 * j] = va.x; xa[4 * j + 1] = va.y; xa[4 * j + 2] = va.z; xa[4 * j + 3] = va.w;
          xr[4 * j] = vr.x; xr[4 * j + 1] = vr.y; xr[4 * j + 2] = vr.z; xr[4 * j + 3] = vr.w;
        }
      } else {
#pragma unroll
        for (int k = 0; k < S; k++) {
          xa[k] = ia[k * 64 + tid];
          xr[k] = ir[k * 64 + tid];
        }
      }
#pragma unroll
      for (int k = 0; k < S; k += 2) {
        int32_t a0, b0, a1, b1;
        unsqueeze_step(cur, xr[k], xa[k], d, a0, b0);
        unsqueeze_step(xa[k], xr[k + 1], xa[k + 1], d, a1, b1);
        cur = xa[k + 1];
        if constexpr (HORIZ) {
          *reinterpret_cast<int4*>(oa + tid * PO + 2 * k) = make_int4(a0, b0, a1, b1);
        } else {
          oa[(2 * k) * 64 + tid] = a0;
          oa[(2 * k + 1) * 64 + tid] = b0;
          oa[(2 * k + 2) * 64 + tid] = a1;
          oa[(2 * k + 3) * 64 + tid] = b1;
        }
      }
    } else {  // the line's last chunk: fewer steps, one by one; an odd line's copied sample behind them
      for (int k = 0; k < sc; k++) {
        const int32_t nxt = HORIZ ? ia[tid * PI + k] : ia[k * 64 + tid];
        const int32_t rs = HORIZ ? ir[tid * PI + k] : ir[k * 64 + tid];
        int32_t va, vb;
        unsqueeze_step(cur, rs, nxt, d, va, vb);
        cur = nxt;
        if constexpr (HORIZ) {
          oa[tid * PO + 2 * k] = va;
          oa[tid * PO + 2 * k + 1] = vb;
        } else {
          oa[(2 * k) * 64 + tid] = va;
          oa[(2 * k + 1) * 64 + tid] = vb;
        }
      }
      if (has_tail) {  // n_out odd: sample 2 w = avg[w] (squeeze.rs:434-437), always in the last chunk
        if constexpr (HORIZ) oa[tid * PO + 2 * sc] = cur;
        else oa[(2 * sc) * 64 + tid] = cur;
      }
    }
    };
  // the movers' iteration c in general (any mix of vector and dword chunks, the line's end)
  auto mover_general = [&](int c) {
    if (steps_of(c + 1) > 0) {
      if (c + 1 < n_fast) {
        if (vloader) stage_fast(c + 1);
      } else {
        stage_chunk(c + 1, pa, pr);
      }
    }
    mark(0);
    if constexpr (FLOW) {
      // Report chunk c - 3, stored during iteration c - 2.  No wait of its own: vector memory operations of a wave
      // complete in the order they were issued, the staging above has just consumed loads issued AFTER those stores
      // (iteration c - 1), so they are acknowledged.  A wait for the stores themselves -- vmcnt(0) here -- would
      // also wait for the stores of iteration c - 1, a write-through round trip that is longer than the chain
      // wave's chunk: measured 3.0-3.2 us per chunk instead of 1.7.  Where that argument has a hole (nothing staged
      // at a line's end; a group with fewer than 64 lines, where a wave may hold stores but no loads) the wait is
      // explicit.
      // Vector chunks: chunks c - 3 and c - 2 both went out as the storer waves' eight stores per iteration, so
      // "all but the newest eight acknowledged" is exact (the loader wave has no stores to wait for).
      if (c >= 3 && !(JXLH_FLOW_EXP & 4)) {
        if (c - 2 < n_fast) {
          if (!vloader) __builtin_amdgcn_s_waitcnt(0x0f78);  // vmcnt(8)
        } else if (n_fast > 0 || steps_of(c + 1) == 0 || l0 + 64 > n_lines) {
          __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
        }
        asm volatile("" ::: "memory");
        flow_publish(min(n_out, 2 * S * (c - 2)));
      }
    }
    mark(1);
    if (steps_of(c + 2) > 0) {
      if (c + 2 < n_fast) {
        if (vloader) fetch_fast(c + 2);
      } else {
        fetch_chunk(c + 2, pa, pr);
      }
    }
    mark(2);
    // (before the stores: reading it back must not wait for them)
    if (steps_of(c + 3) > 0 && (vloader || c + 3 >= n_fast)) flow_peek();
    mark(3);
    if (c >= 1) {
      if (c - 1 < n_fast) {
        if (!vloader) store_fast(c - 1);
      } else {
        store_chunk(c - 1);
      }
    }
    mark(4);
  };
  // ... and while chunks c - 1, c + 1 and c + 2 are all vector chunks: the same schedule with nothing but the vector
  // movers in the loop.  Kept apart because the compiler's wait insertion does not follow which path a wave took: with
  // the dword movers' registers and requests in the same loop it put a wait for ALL outstanding memory operations
  // in front of the loader's next request (0.9 us per iteration, profiles/r05_i_*).
  auto mover_steady = [&](int c) {
    if (vloader) {
      stage_fast(c + 1);
      mark(0);
      if constexpr (FLOW) {
        if (c >= 3 && !(JXLH_FLOW_EXP & 4)) flow_publish(min(n_out, 2 * S * (c - 2)));
      }
      mark(1);
      fetch_fast(c + 2);
      mark(2);
      if (steps_of(c + 3) > 0) flow_peek();
      mark(3);
      mark(4);
    } else {
      if constexpr (FLOW) {
        if (c >= 3 && !(JXLH_FLOW_EXP & 4)) {
          __builtin_amdgcn_s_waitcnt(0x0f78);  // vmcnt(8): everything but the previous iteration's eight stores
          asm volatile("" ::: "memory");
          flow_publish(min(n_out, 2 * S * (c - 2)));
        }
      }
      if (c >= 1) store_fast(c - 1);
    }
  };
  auto end_of_iteration = [&]() {
    if constexpr (FLOW && (JXLH_FLOW_EXP & 64)) {
      lds_barrier();
      if (!chain) mark(5);
    } else if constexpr (FLOW && (JXLH_FLOW_EXP & 16)) {  // experiment: the chain wave's time at the barrier
      const unsigned long long tb = __builtin_amdgcn_s_memrealtime();
      lds_barrier();
      if (chain) F.wait_ticks += __builtin_amdgcn_s_memrealtime() - tb;
    } else {
      lds_barrier();
    }
  };
  int c = 0;
  for (; c + 2 < n_fast; c++) {
    mark(-1);
    if (chain) chain_chunk(c);
    else mover_steady(c);
    end_of_iteration();
  }
  for (; c < n_chunks; c++) {
    mark(-1);
    if (chain) chain_chunk(c);
    else mover_general(c);
    end_of_iteration();
  }
  if (!chain && n_chunks > 0) {
    if (n_chunks - 1 < n_fast) {
      if (!vloader) store_fast(n_chunks - 1);
    } else {
      store_chunk(n_chunks - 1);
    }
    if constexpr (FLOW) {
      __builtin_amdgcn_s_waitcnt(0x0f70);
      flow_publish(n_out);
    }
  }
}


template <bool HORIZ>
__global__ __launch_bounds__(256) void k6_unsqueeze_tiled(const SqueezePlanes pl, size_t avg_lp, size_t avg_ep,
                                                          size_t res_lp, size_t res_ep, size_t out_lp, size_t out_ep,
                                                          int n_lines, int n_out, int vec) {
  constexpr int S = JXLH_SQT_S, PI = JXLH_SQT_PI, PO = JXLH_SQT_PO;
  constexpr int IN_ELEMS = HORIZ ? 64 * PI : 64 * S, OUT_ELEMS = HORIZ ? 64 * PO : 64 * 2 * S;
  __shared__ __attribute__((aligned(16))) int32_t s_avg[2 * IN_ELEMS];
  __shared__ __attribute__((aligned(16))) int32_t s_res[2 * IN_ELEMS];
  __shared__ __attribute__((aligned(16))) int32_t s_out[2 * OUT_ELEMS];
  TiledLines T;
  T.ga = pl.avg[blockIdx.y];
  T.gr = pl.res[blockIdx.y];
  T.go = pl.out[blockIdx.y];
  T.alp = (uint32_t)avg_lp; T.aep = (uint32_t)avg_ep; T.rlp = (uint32_t)res_lp; T.rep = (uint32_t)res_ep;
  T.olp = (uint32_t)out_lp; T.oep = (uint32_t)out_ep;
  T.n_lines = n_lines; T.n_out = n_out; T.l0 = blockIdx.x * 64;
  T.vec = vec;
  FlowLink F{};
  unsqueeze_tiled_lines<HORIZ, false>(T, s_avg, s_res, s_out, F, nullptr);
}

// ---- The streamed levels of a squeeze chain as ONE launch (dataflow).  Run level by level, a chain costs the SUM of
// its levels' line lengths in dependent steps (16 368 for 8192^2: every level waits for the whole level before it).
// But a step only needs the averages NEAR its own position: row group g of a horizontal step can start as soon as the
// vertical step before it has finished rows [64 g, 64 g + 64) -- in all its column groups, which advance together --
// and a column group of the NEXT vertical step follows the row groups of this one at half their speed.  The critical
// path is then monotone in both image axes: about one line of the finest horizontal level plus one of the finest
// vertical one, not the sum over the levels; coarse levels finish under the start of the fine ones.
// Workgroups take tickets (an atomic counter) and tickets are handed out level by level, lowest group first: a
// workgroup only ever waits for lower tickets, which are running or done -- no residency assumption, no deadlock.
// Each level writes its own plane set (no ping-pong: level i + 2 would overwrite what level i + 1 still reads).
constexpr int kFlowMaxLevels = 16;
struct FlowLevel {
  const int32_t* avg[3];
  const int32_t* res[3];
  int32_t* out[3];
  uint32_t avg_lp, avg_ep, res_lp, res_ep, out_lp, out_ep;
  int n_lines, n_out;
  int horiz;
  int first_wg;  // ticket of the level's first workgroup; workgroup = group * n_planes + plane
  int groups;    // 64-line groups per plane
  int flag0;     // index of the level's first progress word (plane-major)
  int dep_same_axis;
  int vec;  // TiledLines::vec
};
struct FlowArgs {
  int n_levels, n_planes;
  int* ticket;    // zero at launch
  int* progress;  // zero at launch
  int* error;
  unsigned long long deadline_ticks;
  // optional profile (nullptr: none), five rows of kFlowMaxLevels: first start (min, preset to ~0), last end (max), time
  // the first mover wave of every workgroup spent polling (sum), its polls (sum), workgroup lifetimes (sum);
  // s_memrealtime ticks
  unsigned long long* prof;
  FlowLevel lv[kFlowMaxLevels];
};
__global__ __launch_bounds__(256) void k6_unsqueeze_flow(const FlowArgs A) {
  constexpr int PI = JXLH_SQT_PI, PO = JXLH_SQT_PO;
  __shared__ __attribute__((aligned(16))) int32_t s_avg[2 * 64 * PI];
  __shared__ __attribute__((aligned(16))) int32_t s_res[2 * 64 * PI];
  __shared__ __attribute__((aligned(16))) int32_t s_out[2 * 64 * PO];
  __shared__ int s_ticket, s_pub;
  if (threadIdx.x == 0) {
    s_ticket = atomicAdd(A.ticket, 1);
    s_pub = 0;
  }
  __syncthreads();
  const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
  const int t = __builtin_amdgcn_readfirstlane(s_ticket);
  int li = 0;
  for (int i = 1; i < A.n_levels; i++) li = t >= A.lv[i].first_wg ? i : li;
  const FlowLevel& L = A.lv[li];
  // everything below is the same for the whole workgroup; the compiler does not see it (the level is found by the
  // ticket, the ticket comes out of LDS) and would do the address arithmetic per lane: said explicitly
  auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane(v); };
  auto uni_ptr = [&](const int32_t* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    // (through a global-address-space pointer: rebuilt from an integer it would be a generic one, and every access
    // through it a flat_ instruction, which counts on both memory counters)
    typedef __attribute__((address_space(1))) int32_t global_i32;
    return (int32_t*)(global_i32*)((uint64_t)uni((uint32_t)(v >> 32)) << 32 | uni((uint32_t)v));
  };
  const int local = t - (int)uni(L.first_wg), n_planes = (int)uni(A.n_planes);
  const int plane = (int)uni(local % n_planes), g = (int)uni(local / n_planes);
  TiledLines T;
  T.ga = uni_ptr(L.avg[plane]);
  T.gr = uni_ptr(L.res[plane]);
  T.go = uni_ptr(L.out[plane]);
  T.alp = uni(L.avg_lp); T.aep = uni(L.avg_ep); T.rlp = uni(L.res_lp); T.rep = uni(L.res_ep);
  T.olp = uni(L.out_lp); T.oep = uni(L.out_ep);
  T.n_lines = (int)uni(L.n_lines); T.n_out = (int)uni(L.n_out); T.l0 = g * 64;
  T.vec = (int)uni(L.vec);
  FlowLink F;
  F.dep = li > 0 ? uni_ptr(A.progress + (A.lv[li - 1].flag0 + plane * A.lv[li - 1].groups) * kFlowWordStride) : nullptr;
  F.dep_same_axis = (int)uni(L.dep_same_axis);
  F.mine = uni_ptr(A.progress + (L.flag0 + plane * L.groups + g) * kFlowWordStride);
  F.error = A.error;
  F.deadline_ticks = A.deadline_ticks;
  F.wait_ticks = 0;
  F.polls = 0;
  for (int i = 0; i < 6; i++) F.phase[i] = 0;
  if (uni(L.horiz)) unsqueeze_tiled_lines<true, true>(T, s_avg, s_res, s_out, F, &s_pub);
  else unsqueeze_tiled_lines<false, true>(T, s_avg, s_res, s_out, F, &s_pub);
  if ((JXLH_FLOW_EXP & 16) && A.prof && threadIdx.x == 0) atomicAdd(&A.prof[2 * kFlowMaxLevels + li], F.wait_ticks);
  if (A.prof && threadIdx.x == 64) {
    atomicMin(&A.prof[li], t_start);
    atomicMax(&A.prof[kFlowMaxLevels + li], (unsigned long long)__builtin_amdgcn_s_memrealtime());
    if (!(JXLH_FLOW_EXP & 16)) atomicAdd(&A.prof[2 * kFlowMaxLevels + li], F.wait_ticks);
    atomicAdd(&A.prof[3 * kFlowMaxLevels + li], (unsigned long long)F.polls);
    atomicAdd(&A.prof[4 * kFlowMaxLevels + li], (unsigned long long)__builtin_amdgcn_s_memrealtime() - t_start);
    if constexpr ((JXLH_FLOW_EXP & 64) != 0)
      for (int i = 0; i < 6; i++) atomicAdd(&A.prof[(5 + i) * kFlowMaxLevels + li], F.phase[i]);
  }
}

// The last step of a colour image's squeeze chain is an unsqueeze of three channels at full size (vertical for square
// and tall images, horizontal for wide ones: default_squeeze, squeeze.rs:71-105), and the transform that follows it in
// the inverse chain is the RCT on the same three channels: a second full read and write of the image.  Fused form:
// the chain wave's lanes are 3 planes x NL lines, so that the three channel values of every output sample meet in one
// workgroup's LDS tile, and the movers apply rct_op while they drain it.  NL = 21 lines for the horizontal step (63
// lanes); 16 columns for the vertical one (48 lanes), so that the per-plane row segments the movers touch are whole
// 64-byte sectors.  The remainder of a line (what k6_unsqueeze_tiled leaves to the chain lane's own stores) goes
// through the tile as well, as a partial chunk.
// NCW = chain waves: 2 widens the vertical tile to 32 columns per plane (96 chain lanes in two waves), so that the
// movers touch whole 128-byte lines of every plane row; vector movers only.
template <bool HORIZ, bool VEC, int NCW = 1>
__global__ __launch_bounds__(64 * (3 * NCW + 1)) void k6_unsqueeze_rct(const SqueezePlanes pl, uint32_t avg_lp, uint32_t avg_ep,
                                                        uint32_t res_lp, uint32_t res_ep, uint32_t out_lp,
                                                        uint32_t out_ep, int n_lines, int n_out, int op, int perm) {
  static_assert(NCW == 1 || (VEC && !HORIZ), "two chain waves: vertical step with the vector movers");
  constexpr int S = JXLH_SQT_S, PI = JXLH_SQT_PI, PO = JXLH_SQT_PO, NL = HORIZ ? 21 : 16 * NCW, NM = 192;
  constexpr int RP = 64 * NCW;   // tile rows (= chain lanes) of the vertical layout
  constexpr int QN = NL / 4;     // 4-column quads per plane row (vector movers)
  constexpr int NLW = NCW;       // loader waves of the vector movers (the wide tile has twice the slots)
  constexpr int NIN = VEC ? 4 * ((S * 3 * QN + 64 * NLW - 1) / (64 * NLW)) : (64 * S + NM - 1) / NM, NPIX = (NL * 2 * S + NM - 1) / NM;
  constexpr int IN_ELEMS = HORIZ ? 64 * PI : RP * S, OUT_ELEMS = HORIZ ? 64 * PO : RP * 2 * S;
  __shared__ __attribute__((aligned(16))) int32_t s_avg[2][IN_ELEMS];
  __shared__ __attribute__((aligned(16))) int32_t s_res[2][IN_ELEMS];
  __shared__ __attribute__((aligned(16))) int32_t s_out[2][OUT_ELEMS];
  const int tid = threadIdx.x;
  const int l0 = blockIdx.x * NL;
  // every step through the chunk pipeline, the last chunk possibly partial (see k6_unsqueeze_tiled)
  const int w = n_out / 2;
  const bool has_tail = n_out & 1;
  const int n_avg = n_out - w;
  const int n_chunks = (n_out + 2 * S - 1) / (2 * S);
  auto steps_of = [&](int c) { return max(0, min(S, w - c * S)); };
  const bool chain = tid < 64 * NCW;
  if (chain) __builtin_amdgcn_s_setprio(3);  // the step time IS this wave's issue latency
  // Mover roles.  The scalar movers (m = 0..191) both load and store.  The vector movers are split, NCW waves loading
  // and NCW + 1 waves storing: a wave with loads and stores outstanding waits for the stores' acknowledgements whenever it
  // needs a loaded value (one in-order vmcnt), which with a handful of wide instructions per chunk is all it does.
  constexpr int MB = 64 * NCW;  // first mover thread
  const bool loader = VEC ? (tid >= MB && tid < MB + 64 * NLW) : !chain, storer = VEC ? tid >= MB + 64 * NLW : !chain;
  const int m = VEC ? (tid < MB + 64 * NLW ? tid - MB : tid - MB - 64 * NLW) : tid - 64;
  constexpr int NML = VEC ? 64 * NLW : 192, NMS = VEC ? 64 * (NCW + 1) : 192;  // storer waves: 2 (3 for the wide tile)
  // perm: which output plane receives w0 / w1 / w2 (as k4_rct)
  int32_t *o0, *o1, *o2;
  switch (perm) {
    default:
    case 0: o0 = pl.out[0]; o1 = pl.out[1]; o2 = pl.out[2]; break;
    case 1: o0 = pl.out[1]; o1 = pl.out[2]; o2 = pl.out[0]; break;
    case 2: o0 = pl.out[2]; o1 = pl.out[0]; o2 = pl.out[1]; break;
    case 3: o0 = pl.out[0]; o1 = pl.out[2]; o2 = pl.out[1]; break;
    case 4: o0 = pl.out[1]; o1 = pl.out[0]; o2 = pl.out[2]; break;
    case 5: o0 = pl.out[2]; o1 = pl.out[1]; o2 = pl.out[0]; break;
  }
  // plane p of a channel triple as base + masked byte offsets (a select chain over three pointers is turned into a
  // table in scratch memory by the compiler, and a scratch load in front of every global load)
  const int64_t a_d1 = (const char*)pl.avg[1] - (const char*)pl.avg[0], a_d2 = (const char*)pl.avg[2] - (const char*)pl.avg[0];
  const int64_t r_d1 = (const char*)pl.res[1] - (const char*)pl.res[0], r_d2 = (const char*)pl.res[2] - (const char*)pl.res[0];
  auto plane_ptr = [](const int32_t* base, int64_t d1, int64_t d2, int p) {
    const int64_t off = (-(int64_t)(p == 1) & d1) | (-(int64_t)(p == 2) & d2);
    return (const int32_t*)((const char*)base + off);
  };
  // tile row r = plane * NL + line; rows from 3 NL on are unused
  auto row_of = [&](int r, int& p, int& q) {
    p = r / NL;
    q = r - p * NL;
    return r < 3 * NL && l0 + q < n_lines;
  };
  // tile position of (row r, element k): one line per row of PI / PO dwords for the horizontal step (the chain lane
  // reads its row with 128-bit accesses), one element row of 64 lanes for the vertical one
  auto in_idx = [](int r, int k) { return HORIZ ? r * PI + k : k * RP + r; };
  auto out_idx = [](int r, int k) { return HORIZ ? r * PO + k : k * RP + r; };
  // VEC (vertical step, 16-byte aligned planes, column count a multiple of 4): a mover slot is four columns of one
  // plane's element row -- 16 row segments of 64 bytes per wave instruction instead of three (or four, when storing),
  // which is what this step's throughput hangs on once three planes stream through every workgroup.  The register
  // arrays then hold int4 slots: (k, plane, quad) for the inputs, (k, quad) for the outputs.
  static_assert(!(VEC && HORIZ), "the vector movers are for the vertical step");
  constexpr int NVIN = (S * 3 * QN + NML - 1) / NML, NVOUT = (2 * S * QN + NMS - 1) / NMS;
  auto fetch_chunk = [&](int c, int32_t(&va)[NIN], int32_t(&vr)[NIN]) {
    if constexpr (VEC) {
#pragma unroll
      for (int j = 0; j < NVIN; j++) {
        const int f = m + j * NML, k = f / (3 * QN), rem = f - k * (3 * QN), p = rem / QN, qq = rem % QN;
        // slots past the tile (the last j) and quads past the last column read a valid address instead of being
        // predicated (a select between a load and a constant becomes a load through a selected POINTER, via scratch);
        // what they fetch is never staged / never stored
        const int kk = min(k, S - 1);
        const uint32_t col = (uint32_t)(l0 + 4 * qq < n_lines ? l0 + 4 * qq : l0);
        const int32_t* ap = plane_ptr(pl.avg[0], a_d1, a_d2, p);
        const int32_t* rp = plane_ptr(pl.res[0], r_d1, r_d2, p);
        // the line's end: next_avg of the last step of an even line is the last average itself (index clamped);
        // steps past the line fetch valid rows that are never used
        const int ia = min(c * S + 1 + kk, n_avg - 1), ir = min(c * S + kk, max(w - 1, 0));
#ifndef JXLH_SQRCT_NT
#define JXLH_SQRCT_NT false
#endif
        const int4 xa = gload_i4<JXLH_SQRCT_NT>(ap + (uint32_t)ia * avg_ep + col);
        const int4 xr = gload_i4<JXLH_SQRCT_NT>(rp + (uint32_t)ir * res_ep + col);
        va[4 * j] = xa.x; va[4 * j + 1] = xa.y; va[4 * j + 2] = xa.z; va[4 * j + 3] = xa.w;
        vr[4 * j] = xr.x; vr[4 * j + 1] = xr.y; vr[4 * j + 2] = xr.z; vr[4 * j + 3] = xr.w;
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < NIN; j++) {
      const int f = m + j * NM, r = HORIZ ? f / S : f % 64, k = HORIZ ? f % S : f / 64;
      int p, q;
      const bool ok = row_of(r, p, q) && f < 64 * S && c * S + k < w;
      const int32_t* ap = plane_ptr(pl.avg[0], a_d1, a_d2, p);
      const int32_t* rp = plane_ptr(pl.res[0], r_d1, r_d2, p);
      const int ia = min(c * S + 1 + k, n_avg - 1), ir = min(c * S + k, max(w - 1, 0));
      va[j] = ok ? ap[(uint32_t)(l0 + q) * avg_lp + (uint32_t)ia * avg_ep] : 0;
      vr[j] = ok ? rp[(uint32_t)(l0 + q) * res_lp + (uint32_t)ir * res_ep] : 0;
    }
  };
  auto stage_chunk = [&](int c, const int32_t(&va)[NIN], const int32_t(&vr)[NIN]) {
    if constexpr (VEC) {
#pragma unroll
      for (int j = 0; j < NVIN; j++) {
        const int f = m + j * NML, k = f / (3 * QN), rem = f - k * (3 * QN), p = rem / QN, qq = rem % QN;
        if (f < S * 3 * QN) {
          const int idx = k * RP + p * NL + 4 * qq;
          *reinterpret_cast<int4*>(&s_avg[c & 1][idx]) = make_int4(va[4 * j], va[4 * j + 1], va[4 * j + 2], va[4 * j + 3]);
          *reinterpret_cast<int4*>(&s_res[c & 1][idx]) = make_int4(vr[4 * j], vr[4 * j + 1], vr[4 * j + 2], vr[4 * j + 3]);
        }
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < NIN; j++) {
      const int f = m + j * NM, r = HORIZ ? f / S : f % 64, k = HORIZ ? f % S : f / 64;
      if (f < 64 * S) {
        s_avg[c & 1][in_idx(r, k)] = va[j];
        s_res[c & 1][in_idx(r, k)] = vr[j];
      }
    }
  };
  // drain `count` (<= 2 S) output samples per line of chunk c, through the RCT
  auto rct4 = [&](const int4& v0, const int4& v1, const int4& v2, int4& x, int4& y, int4& z) {
    switch (op) {
#define JXLH_RCT4(OP)                          \
  case OP:                                     \
    rct_op<OP>(v0.x, v1.x, v2.x, x.x, y.x, z.x); \
    rct_op<OP>(v0.y, v1.y, v2.y, x.y, y.y, z.y); \
    rct_op<OP>(v0.z, v1.z, v2.z, x.z, y.z, z.z); \
    rct_op<OP>(v0.w, v1.w, v2.w, x.w, y.w, z.w); \
    break;
      JXLH_RCT4(0) JXLH_RCT4(1) JXLH_RCT4(2) JXLH_RCT4(3) JXLH_RCT4(4) JXLH_RCT4(5)
      default:
      JXLH_RCT4(6)
#undef JXLH_RCT4
    }
  };
  auto store_chunk = [&](int c, int count) {
    const int32_t* so = s_out[c & 1];
    if constexpr (VEC) {
#pragma unroll
      for (int j = 0; j < NVOUT; j++) {
        const int f = m + j * NMS, k = f / QN, qq = f % QN;
        if (f < 2 * S * QN && l0 + 4 * qq < n_lines && k < count) {
          const int4 v0 = *reinterpret_cast<const int4*>(so + k * RP + 4 * qq);
          const int4 v1 = *reinterpret_cast<const int4*>(so + k * RP + NL + 4 * qq);
          const int4 v2 = *reinterpret_cast<const int4*>(so + k * RP + 2 * NL + 4 * qq);
          int4 x, y, z;
          rct4(v0, v1, v2, x, y, z);
          const uint32_t off = (uint32_t)(2 * c * S + k) * out_ep + (uint32_t)(l0 + 4 * qq);
          gstore_i4<JXLH_SQRCT_NT>(o0 + off, x);
          gstore_i4<JXLH_SQRCT_NT>(o1 + off, y);
          gstore_i4<JXLH_SQRCT_NT>(o2 + off, z);
        }
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < NPIX; j++) {
      const int f = m + j * NM, q = HORIZ ? f / (2 * S) : f % NL, k = HORIZ ? f % (2 * S) : f / NL;
      if (f < NL * 2 * S && l0 + q < n_lines && k < count) {
        const int32_t v0 = so[out_idx(q, k)], v1 = so[out_idx(NL + q, k)], v2 = so[out_idx(2 * NL + q, k)];
        int32_t x, y, z;
        switch (op) {
          case 0: rct_op<0>(v0, v1, v2, x, y, z); break;
          case 1: rct_op<1>(v0, v1, v2, x, y, z); break;
          case 2: rct_op<2>(v0, v1, v2, x, y, z); break;
          case 3: rct_op<3>(v0, v1, v2, x, y, z); break;
          case 4: rct_op<4>(v0, v1, v2, x, y, z); break;
          case 5: rct_op<5>(v0, v1, v2, x, y, z); break;
          default: rct_op<6>(v0, v1, v2, x, y, z); break;
        }
        const uint32_t off = (uint32_t)(l0 + q) * out_lp + (uint32_t)(2 * c * S + k) * out_ep;
        o0[off] = x;
        o1[off] = y;
        o2[off] = z;
      }
    }
  };

  int cp, cq;
  const bool live = chain && row_of(tid, cp, cq);
  int32_t cur = 0, d = 0;
  if (live) cur = (plane_ptr(pl.avg[0], a_d1, a_d2, cp) + (size_t)(l0 + cq) * avg_lp)[0];
  auto count_of = [&](int c) { return min(2 * S, n_out - 2 * c * S); };
  int32_t pa[NIN], pr[NIN];
  if (loader && steps_of(0) > 0) {
    fetch_chunk(0, pa, pr);
    stage_chunk(0, pa, pr);
    if (steps_of(1) > 0) fetch_chunk(1, pa, pr);
  }
  lds_barrier();
  for (int c = 0; c < n_chunks; c++) {
    if (!chain) {
      if (loader && steps_of(c + 1) > 0) stage_chunk(c + 1, pa, pr);
      if (loader && steps_of(c + 2) > 0) fetch_chunk(c + 2, pa, pr);
      if (storer && c >= 1) store_chunk(c - 1, count_of(c - 1));
    } else {
      const int32_t* ia = s_avg[c & 1];
      const int32_t* ir = s_res[c & 1];
      int32_t* oa = s_out[c & 1];
      const int sc = steps_of(c);
      if (sc == S) {
        int32_t xa[S], xr[S];
        if constexpr (HORIZ) {
#pragma unroll
          for (int j = 0; j < S / 4; j++) {
            const int4 va = *reinterpret_cast<const int4*>(ia + tid * PI + 4 * j);
            const int4 vr = *reinterpret_cast<const int4*>(ir + tid * PI + 4 * j);
            xa[4 * j] = va.x; xa[4 * j + 1] = va.y; xa[4 * j + 2] = va.z; xa[4 * j + 3] = va.w;
            xr[4 * j] = vr.x; xr[4 * j + 1] = vr.y; xr[4 * j + 2] = vr.z; xr[4 * j + 3] = vr.w;
          }
        } else {
#pragma unroll
          for (int k = 0; k < S; k++) {
            xa[k] = ia[k * RP + tid];
            xr[k] = ir[k * RP + tid];
          }
        }
#pragma unroll
        for (int k = 0; k < S; k += 2) {
          int32_t a0, b0, a1, b1;
          unsqueeze_step(cur, xr[k], xa[k], d, a0, b0);
          unsqueeze_step(xa[k], xr[k + 1], xa[k + 1], d, a1, b1);
          cur = xa[k + 1];
          if constexpr (HORIZ) {
            *reinterpret_cast<int4*>(oa + tid * PO + 2 * k) = make_int4(a0, b0, a1, b1);
          } else {
            oa[(2 * k) * RP + tid] = a0;
            oa[(2 * k + 1) * RP + tid] = b0;
            oa[(2 * k + 2) * RP + tid] = a1;
            oa[(2 * k + 3) * RP + tid] = b1;
          }
        }
      } else if (live) {  // the line's last chunk: fewer steps, one by one; an odd line's copied sample behind them
        for (int k = 0; k < sc; k++) {
          const int32_t nxt = ia[in_idx(tid, k)], rs = ir[in_idx(tid, k)];
          int32_t va, vb;
          unsqueeze_step(cur, rs, nxt, d, va, vb);
          cur = nxt;
          oa[out_idx(tid, 2 * k)] = va;
          oa[out_idx(tid, 2 * k + 1)] = vb;
        }
        if (has_tail) oa[out_idx(tid, 2 * sc)] = cur;  // sample 2 w = avg[w] (squeeze.rs:434-437, :468-476)
      }
    }
    lds_barrier();
  }
  if (storer && n_chunks > 0) store_chunk(n_chunks - 1, count_of(n_chunks - 1));
}

// TiledLines::vec: may the tiled kernels move this step with 16-byte buffer accesses?  (JXLH_SQ_VEC=0: never -- tests, A/B)
static int tiled_vec_ok(int horizontal, int n_planes, const int32_t* const avg[], size_t avg_stride, const int32_t* const res[],
                        size_t res_stride, uint32_t out_w, uint32_t out_h, int32_t* const out[], size_t out_stride) {
  static const bool off = getenv("JXLH_SQ_VEC") && getenv("JXLH_SQ_VEC")[0] == '0';
  if (off) return 0;
  if (avg_stride % 4 || res_stride % 4 || out_stride % 4) return 0;
  if (!horizontal && out_w % 4) return 0;  // (lines are columns: a 64-column group must be whole quads -- it is; the planes' rows must be)
  for (int i = 0; i < n_planes && i < 3; i++)
    if ((uintptr_t)avg[i] % 16 || (uintptr_t)res[i] % 16 || (uintptr_t)out[i] % 16) return 0;
  // every byte offset from a plane's first sample fits 32 bits, with room for the kernel's slot arithmetic
  const size_t lim = (size_t)1 << 30;
  return out_stride * (size_t)out_h < lim && avg_stride * (size_t)out_h < lim && res_stride * (size_t)out_h < lim;
}

void launch_unsqueeze(hipStream_t s, int horizontal, int n_planes, const int32_t* const avg[], size_t avg_stride,
                      const int32_t* const res[], size_t res_stride, uint32_t out_w, uint32_t out_h,
                      int32_t* const out[], size_t out_stride) {
  if (out_w == 0 || out_h == 0 || n_planes <= 0) return;
  SqueezePlanes pl{};
  bool aligned = (avg_stride % 4 == 0) && (res_stride % 4 == 0) && (out_stride % 4 == 0);
  for (int i = 0; i < n_planes && i < 3; i++) {
    pl.avg[i] = avg[i];
    pl.res[i] = res[i];
    pl.out[i] = out[i];
    aligned = aligned && ((uintptr_t)avg[i] % 16 == 0) && ((uintptr_t)res[i] % 16 == 0) && ((uintptr_t)out[i] % 16 == 0);
  }
  // long lines go through the mover / chain workgroups; short ones (the early levels of a squeeze chain) keep the
  // one-wave kernel: nothing to stream, and a 256-thread workgroup would idle three waves
  const int n_steps = (int)(horizontal ? out_w : out_h) / 2;
  const size_t span = out_stride * (size_t)out_h;  // the largest of the three planes; the kernel's offsets are 32-bit
  if (n_steps >= 4 * JXLH_SQT_S && span < ((size_t)1 << 31) && avg_stride * (size_t)out_h < ((size_t)1 << 31) &&
      res_stride * (size_t)out_h < ((size_t)1 << 31)) {
    const int n_lines = (int)(horizontal ? out_h : out_w);
    const dim3 grid((n_lines + 63) / 64, n_planes);
    const int vec = tiled_vec_ok(horizontal, n_planes, avg, avg_stride, res, res_stride, out_w, out_h, out, out_stride);
    if (horizontal)
      hipLaunchKernelGGL(k6_unsqueeze_tiled<true>, grid, dim3(256), 0, s, pl, avg_stride, (size_t)1, res_stride,
                         (size_t)1, out_stride, (size_t)1, n_lines, (int)out_w, vec);
    else
      hipLaunchKernelGGL(k6_unsqueeze_tiled<false>, grid, dim3(256), 0, s, pl, (size_t)1, avg_stride, (size_t)1,
                         res_stride, (size_t)1, out_stride, n_lines, (int)out_h, vec);
    return;
  }
  if (horizontal) {
    const int n_lines = (int)out_h;
    const dim3 grid((n_lines + 63) / 64, n_planes);
    if (aligned) {
      hipLaunchKernelGGL(k6_unsqueeze<true>, grid, dim3(64), 0, s, pl, avg_stride, (size_t)1, res_stride, (size_t)1,
                         out_stride, (size_t)1, n_lines, (int)out_w);
    } else {
      hipLaunchKernelGGL(k6_unsqueeze<false>, grid, dim3(64), 0, s, pl, avg_stride, (size_t)1, res_stride, (size_t)1,
                         out_stride, (size_t)1, n_lines, (int)out_w);
    }
  } else {
    const int n_lines = (int)out_w;
    const dim3 grid((n_lines + 63) / 64, n_planes);
    hipLaunchKernelGGL(k6_unsqueeze<false>, grid, dim3(64), 0, s, pl, (size_t)1, avg_stride, (size_t)1, res_stride,
                       (size_t)1, out_stride, n_lines, (int)out_h);
  }
}

// ---- dataflow launch of consecutive tiled steps (see k6_unsqueeze_flow)
bool unsqueeze_tiled_eligible(int horizontal, uint32_t out_w, uint32_t out_h, size_t avg_stride, size_t res_stride,
                              size_t out_stride) {
  const int n_steps = (int)(horizontal ? out_w : out_h) / 2;
  const size_t lim = (size_t)1 << 31;
  return n_steps >= 4 * JXLH_SQT_S && out_stride * (size_t)out_h < lim && avg_stride * (size_t)out_h < lim &&
         res_stride * (size_t)out_h < lim;
}
int unsqueeze_flow_max_steps() { return kFlowMaxLevels; }
size_t unsqueeze_flow_words(int n_planes, int n_steps, const FlowStep* steps) {
  size_t words = 2 * kFlowWordStride;  // the ticket; slack behind the last word (a peek may look one word past a level's groups)
  for (int i = 0; i < n_steps; i++) {
    const int n_lines = (int)(steps[i].horizontal ? steps[i].out_h : steps[i].out_w);
    words += (size_t)n_planes * ((n_lines + 63) / 64) * kFlowWordStride;
  }
  return words;
}
void launch_unsqueeze_flow(hipStream_t s, int n_planes, int n_steps, const FlowStep* steps, int* scratch, int* error,
                           float deadline_s, unsigned long long* prof) {
  FlowArgs A{};
  A.n_levels = n_steps;
  A.n_planes = n_planes;
  A.ticket = scratch;
  A.progress = scratch + kFlowWordStride;
  A.error = error;
  A.deadline_ticks = (unsigned long long)(deadline_s * 1.0e8);
  int wg = 0, flag = 0;
  for (int i = 0; i < n_steps; i++) {
    const FlowStep& st = steps[i];
    FlowLevel& L = A.lv[i];
    for (int p = 0; p < 3; p++) {
      const int q = p < n_planes ? p : 0;
      L.avg[p] = st.avg[q];
      L.res[p] = st.res[q];
      L.out[p] = st.out[q];
    }
    const bool hz = st.horizontal != 0;
    L.avg_lp = hz ? (uint32_t)st.avg_stride : 1u;
    L.avg_ep = hz ? 1u : (uint32_t)st.avg_stride;
    L.res_lp = hz ? (uint32_t)st.res_stride : 1u;
    L.res_ep = hz ? 1u : (uint32_t)st.res_stride;
    L.out_lp = hz ? (uint32_t)st.out_stride : 1u;
    L.out_ep = hz ? 1u : (uint32_t)st.out_stride;
    L.n_lines = (int)(hz ? st.out_h : st.out_w);
    L.n_out = (int)(hz ? st.out_w : st.out_h);
    L.horiz = hz;
    L.groups = (L.n_lines + 63) / 64;
    L.first_wg = wg;
    L.flag0 = flag;
    L.dep_same_axis = i > 0 && (steps[i - 1].horizontal != 0) == hz;
    L.vec = tiled_vec_ok(st.horizontal, n_planes, st.avg, st.avg_stride, st.res, st.res_stride, st.out_w, st.out_h, st.out,
                         st.out_stride);
    wg += L.groups * n_planes;
    flag += L.groups * n_planes;
  }
  (void)hipMemsetAsync(scratch, 0, sizeof(int) * (size_t)(1 + flag) * kFlowWordStride, s);
  A.prof = prof;
  if (prof) {
    (void)hipMemsetAsync(prof, 0xff, sizeof(unsigned long long) * kFlowMaxLevels, s);
    (void)hipMemsetAsync(prof + kFlowMaxLevels, 0, sizeof(unsigned long long) * 10 * kFlowMaxLevels, s);
  }
  static const int lds_pad = getenv("JXLH_FLOW_LDS_PAD") ? atoi(getenv("JXLH_FLOW_LDS_PAD")) : 0;  // experiments: residency
  hipLaunchKernelGGL(k6_unsqueeze_flow, dim3(wg), dim3(256), (size_t)lds_pad, s, A);
}

// Unsqueeze of three channels + inverse RCT on them, one pass (planes below 2^31 samples; the caller falls back to
// the two separate launches otherwise).
bool launch_unsqueeze_rct(hipStream_t s, int horizontal, const int32_t* const avg[3], size_t avg_stride,
                          const int32_t* const res[3], size_t res_stride, uint32_t out_w, uint32_t out_h,
                          int32_t* const out[3], size_t out_stride, int op, int perm) {
  if (out_w == 0 || out_h == 0) return true;
  const size_t lim = (size_t)1 << 31;
  if (out_stride * (size_t)out_h >= lim || avg_stride * (size_t)out_h >= lim || res_stride * (size_t)out_h >= lim)
    return false;
  SqueezePlanes pl{};
  for (int i = 0; i < 3; i++) {
    pl.avg[i] = avg[i];
    pl.res[i] = res[i];
    pl.out[i] = out[i];
  }
  if (horizontal) {
    const dim3 grid((out_h + 20) / 21);
    hipLaunchKernelGGL((k6_unsqueeze_rct<true, false>), grid, dim3(256), 0, s, pl, (uint32_t)avg_stride, 1u, (uint32_t)res_stride,
                       1u, (uint32_t)out_stride, 1u, (int)out_h, (int)out_w, op, perm);
  } else {
    const dim3 grid((out_w + 15) / 16);
    bool vec = out_w % 4 == 0 && avg_stride % 4 == 0 && res_stride % 4 == 0 && out_stride % 4 == 0;
    for (int i = 0; i < 3; i++)
      vec = vec && ((uintptr_t)avg[i] % 16 == 0) && ((uintptr_t)res[i] % 16 == 0) && ((uintptr_t)out[i] % 16 == 0);
    if (vec && out_w >= 4096 && out_w % 32 == 0) {
      // wide planes: 32 columns per plane and workgroup (whole 128-byte lines), one workgroup per CU is enough
      hipLaunchKernelGGL((k6_unsqueeze_rct<false, true, 2>), dim3((out_w + 31) / 32), dim3(448), 0, s, pl, 1u,
                         (uint32_t)avg_stride, 1u, (uint32_t)res_stride, 1u, (uint32_t)out_stride, (int)out_w,
                         (int)out_h, op, perm);
    } else if (vec)
      hipLaunchKernelGGL((k6_unsqueeze_rct<false, true>), grid, dim3(256), 0, s, pl, 1u, (uint32_t)avg_stride, 1u,
                         (uint32_t)res_stride, 1u, (uint32_t)out_stride, (int)out_w, (int)out_h, op, perm);
    else
      hipLaunchKernelGGL((k6_unsqueeze_rct<false, false>), grid, dim3(256), 0, s, pl, 1u, (uint32_t)avg_stride, 1u,
                         (uint32_t)res_stride, 1u, (uint32_t)out_stride, (int)out_w, (int)out_h, op, perm);
  }
  return true;
}

// The first levels of a squeeze chain are tiny (8x8 -> 16x8 -> 16x16 -> ... ) and each one, as its own launch, costs
// ~10 us however few steps it has: the eight levels up to 128 x 128 of the default chain took 83 us for 240 steps.
// Here one workgroup per plane walks ALL of them: the running average plane lives in LDS (two buffers swapped per
// level), a level's residual plane is staged through LDS with coalesced loads, lanes = lines, and only the last level
// is written out.
struct SqueezeLevels {
  int n_levels;
  int base_w, base_h;
  uint32_t base_stride, out_stride;
  const int32_t* base[3];
  int32_t* out[3];
  struct {
    int horizontal, out_w, out_h;
    uint32_t res_stride;
    const int32_t* res[3];
  } lv[JXLH_SQL_LEVELS];
};
__global__ __launch_bounds__(256) void k6_unsqueeze_levels(const SqueezeLevels L) {
  // Two plane buffers, swapped per level.  A level's output is twice its input, so only the buffer the LAST level
  // writes has to hold a full 128 x 128 plane; the other one (and the residual tile) hold half planes.  Row pitch =
  // width | 1 (odd: lane = row and lane = column are both conflict-free).
  constexpr int kFull = JXLH_SQL_MAX * (JXLH_SQL_MAX + 1), kHalf = JXLH_SQL_MAX * (JXLH_SQL_MAX / 2 + 1);
  __shared__ int32_t s_big[kFull], s_half[kHalf], s_r[kHalf];
  const int tid = threadIdx.x, pl = blockIdx.x;
  // level i reads buf[(i + off) & 1] and writes the other one; the last level must write s_big (index 0)
  const int off = L.n_levels & 1;
  auto buf = [&](int k) { return (k & 1) ? s_half : s_big; };
  int cw = L.base_w, ch = L.base_h;
  // plane <-> thread mapping without divisions: tx = tid % TW, TW = the power of two >= the plane's width, 256 / TW
  // rows per pass (uniform per level); at most 64 passes for a 128-row plane
  auto tile_log2 = [](int width) { return width <= 1 ? 0 : 32 - __builtin_clz((unsigned)(width - 1)); };
  {
    int32_t* b0 = buf(off);
    const int pc = cw | 1, lg = tile_log2(cw), tx = tid & ((1 << lg) - 1), ty = tid >> lg, rpp = 256 >> lg;
    for (int y = ty; y < ch; y += rpp)
      if (tx < cw) b0[y * pc + tx] = L.base[pl][(size_t)y * L.base_stride + tx];
  }
  // a level's residuals are fetched into registers (coalesced, <= 33 per thread) while the PREVIOUS level's recurrence
  // runs, and dropped into the LDS tile at the level's start: the global round trip is off the serial path
  constexpr int kResPasses = 64;
  int32_t rq[kResPasses];
  auto fetch_res = [&](int lv) {
    const int horizontal = L.lv[lv].horizontal, ow = L.lv[lv].out_w, oh = L.lv[lv].out_h;
    const int rw = horizontal ? ow / 2 : ow, rh = horizontal ? oh : oh / 2;
    if (rw == 0 || rh == 0) return;  // a one-sample axis has no residuals (and no plane to read)
    const int32_t* __restrict__ res = L.lv[lv].res[pl];
    const uint32_t rstride = L.lv[lv].res_stride;
    const int lg = tile_log2(rw), tx = tid & ((1 << lg) - 1), ty = tid >> lg, rpp = 256 >> lg;
    const int txc = min(tx, rw - 1);
#pragma unroll
    for (int j = 0; j < kResPasses; j++) {
      const int y = ty + j * rpp;
      if (j * rpp >= rh) break;  // uniform
      rq[j] = res[(size_t)min(y, rh - 1) * rstride + txc];  // past the plane: a valid sample, never staged
    }
  };
  fetch_res(0);
  for (int lv = 0; lv < L.n_levels; lv++) {
    const int horizontal = L.lv[lv].horizontal, ow = L.lv[lv].out_w, oh = L.lv[lv].out_h;
    const int rw = horizontal ? ow / 2 : ow, rh = horizontal ? oh : oh / 2;
    const int pc = cw | 1, po = ow | 1, pr = rw | 1;
    {
      const int lg = tile_log2(rw), tx = tid & ((1 << lg) - 1), ty = tid >> lg, rpp = 256 >> lg;
#pragma unroll
      for (int j = 0; j < kResPasses; j++) {
        const int y = ty + j * rpp;
        if (j * rpp >= rh) break;  // uniform
        if (tx < rw && y < rh) s_r[y * pr + tx] = rq[j];
      }
    }
    __syncthreads();
    if (lv + 1 < L.n_levels) fetch_res(lv + 1);
    const int32_t* cur_buf = buf(lv + off);
    int32_t* nxt_buf = buf(lv + off + 1);
    const int n_lines = horizontal ? oh : ow, n_out = horizontal ? ow : oh;
    if (tid < n_lines) {
      // element i of line `tid`: horizontal = row tid, vertical = column tid
      const int32_t* a = cur_buf + (horizontal ? tid * pc : tid);
      const int32_t* r = s_r + (horizontal ? tid * pr : tid);
      int32_t* o = nxt_buf + (horizontal ? tid * po : tid);
      const int aep = horizontal ? 1 : pc, rep = horizontal ? 1 : pr, oep = horizontal ? 1 : po;
      const int w = n_out / 2;
      if (w == 0) {
        o[0] = a[0];
      } else {
        const bool has_tail = n_out & 1;
        const int n_main = has_tail ? w : w - 1;
        int32_t c0 = a[0], d = 0;
        int i = 0;
        for (; i + 4 <= n_main; i += 4) {  // operands of four steps in one LDS round trip
          int32_t nx[4], rr[4], va[4], vb[4];
#pragma unroll
          for (int k = 0; k < 4; k++) {
            nx[k] = a[(i + k + 1) * aep];
            rr[k] = r[(i + k) * rep];
          }
#pragma unroll
          for (int k = 0; k < 4; k++) {
            unsqueeze_step(c0, rr[k], nx[k], d, va[k], vb[k]);
            c0 = nx[k];
          }
#pragma unroll
          for (int k = 0; k < 4; k++) {
            o[(2 * (i + k)) * oep] = va[k];
            o[(2 * (i + k) + 1) * oep] = vb[k];
          }
        }
        for (; i < n_main; i++) {
          const int32_t nx = a[(i + 1) * aep];
          int32_t va, vb;
          unsqueeze_step(c0, r[i * rep], nx, d, va, vb);
          o[(2 * i) * oep] = va;
          o[(2 * i + 1) * oep] = vb;
          c0 = nx;
        }
        if (!has_tail) {
          int32_t va, vb;
          unsqueeze_step(c0, r[(w - 1) * rep], c0, d, va, vb);
          o[(2 * w - 2) * oep] = va;
          o[(2 * w - 1) * oep] = vb;
        } else {
          o[(2 * w) * oep] = c0;
        }
      }
    }
    __syncthreads();
    cw = ow;
    ch = oh;
  }
  {
    const int32_t* fin = buf(L.n_levels + off);  // = s_big
    const int pc = cw | 1, lg = tile_log2(cw), tx = tid & ((1 << lg) - 1), ty = tid >> lg, rpp = 256 >> lg;
    for (int y = ty; y < ch; y += rpp)
      if (tx < cw) L.out[pl][(size_t)y * L.out_stride + tx] = fin[y * pc + tx];
  }
}

// n_levels <= 16 levels, every plane side <= 128: one launch.  Returns false when the chain does not qualify.
bool launch_unsqueeze_levels(hipStream_t s, int n_planes, int n_levels, const int* horizontal, const uint32_t* out_w,
                             const uint32_t* out_h, const int32_t* const* res, const size_t* res_stride,
                             const int32_t* const base[], size_t base_stride, uint32_t base_w, uint32_t base_h,
                             int32_t* const out[], size_t out_stride) {
  if (n_levels < 1 || n_levels > JXLH_SQL_LEVELS || n_planes < 1 || n_planes > 3) return false;
  if (base_w == 0 || base_h == 0 || base_w > JXLH_SQL_MAX || base_h > JXLH_SQL_MAX) return false;
  SqueezeLevels L{};
  L.n_levels = n_levels;
  L.base_w = (int)base_w;
  L.base_h = (int)base_h;
  L.base_stride = (uint32_t)base_stride;
  L.out_stride = (uint32_t)out_stride;
  for (int p = 0; p < n_planes; p++) {
    L.base[p] = base[p];
    L.out[p] = out[p];
  }
  for (int i = 0; i < n_levels; i++) {
    if (out_w[i] == 0 || out_h[i] == 0 || out_w[i] > JXLH_SQL_MAX || out_h[i] > JXLH_SQL_MAX) return false;
    L.lv[i].horizontal = horizontal[i];
    L.lv[i].out_w = (int)out_w[i];
    L.lv[i].out_h = (int)out_h[i];
    L.lv[i].res_stride = (uint32_t)res_stride[i];
    for (int p = 0; p < n_planes; p++) L.lv[i].res[p] = res[i * 3 + p];
  }
  // every plane written to the half-size buffer (the base when n_levels is odd, and every second level counted from
  // the end) must fit it: rows * (width | 1) <= 128 * 65
  const size_t half_cap = (size_t)JXLH_SQL_MAX * (JXLH_SQL_MAX / 2 + 1);
  auto fits_half = [&](uint32_t pw, uint32_t ph) { return (size_t)ph * (pw | 1u) <= half_cap; };
  if ((n_levels & 1) && !fits_half(base_w, base_h)) return false;
  for (int i = n_levels - 2; i >= 0; i -= 2)
    if (!fits_half(out_w[i], out_h[i])) return false;
  for (int i = 0; i < n_levels; i++) {  // the residual tile has the same capacity
    const uint32_t rw = horizontal[i] ? out_w[i] / 2 : out_w[i], rh = horizontal[i] ? out_h[i] : out_h[i] / 2;
    if (rw && rh && !fits_half(rw, rh)) return false;
  }
  hipLaunchKernelGGL(k6_unsqueeze_levels, dim3(n_planes), dim3(256), 0, s, L);
  return true;
}

// ---------------------------------------------------------------------------------------------------------------
// Smooth unsqueeze: what a squeeze step runs while its residual channel has not arrived (progressive previews;
// transforms/step.rs:138-150 picks the kind, :841-851 dispatches): smooth_h / smooth_v / smooth_2d_unsqueeze
// (squeeze.rs:908-1225).  A pure 5x5 stencil on the average channel -- unlike the regular step there is no
// recurrence, so it is one thread per column of a 64 x 16 tile of average samples, walking four rows with a sliding window; the
// tile's 68 x 20 window is staged through LDS once.
// Arithmetic order is the reference's (four partial sums of <= 4 FMAs from zero, (a + b) + (c + d), +-0.5,
// truncating convert: the scalar / NEON / wasm as_i32; the x86 back-ends' cvtps rounds a second time, see
// DESIGN.md 4).
struct SmTap {
  int n;
  float w;
};
#define SW2 0.62646443f
#define SW10 0.24413736f
#define SW18 0.06118795f
#define SW26 -0.01328634f
#define SW34 -0.03355509f
#define SW50 -0.02015225f
#define SW58 -0.01033307f
#define SW74 -0.00056067f
#define SV1 0.69472290f
#define SV9 0.27861324f
#define SV17 0.07666797f
#define SV25 -0.00778371f
#define SV41 -0.03143468f
#define SV49 -0.02150597f
#define SV65 -0.00434251f
#define SV73 -0.00078780f
// n < 0: the slot is empty (the 2-D kernel's third partial sum has three taps)
__device__ static constexpr SmTap kSm2d[4][16] = {
    {{1, SW58}, {2, SW50}, {3, SW74}, {5, SW58}, {6, SW18}, {7, SW10}, {8, SW34}, {10, SW50},
     {11, SW10}, {12, SW2}, {13, SW26}, {-1, 0.f}, {15, SW74}, {16, SW34}, {17, SW26}, {18, SW50}},
    {{1, SW74}, {2, SW50}, {3, SW58}, {6, SW34}, {7, SW10}, {8, SW18}, {9, SW58}, {11, SW26},
     {12, SW2}, {13, SW10}, {14, SW50}, {-1, 0.f}, {16, SW50}, {17, SW26}, {18, SW34}, {19, SW74}},
    {{5, SW74}, {6, SW34}, {7, SW26}, {8, SW50}, {10, SW50}, {11, SW10}, {12, SW2}, {13, SW26},
     {15, SW58}, {16, SW18}, {17, SW10}, {-1, 0.f}, {18, SW34}, {21, SW58}, {22, SW50}, {23, SW74}},
    {{6, SW50}, {7, SW26}, {8, SW34}, {9, SW74}, {11, SW26}, {12, SW2}, {13, SW10}, {14, SW50},
     {16, SW34}, {17, SW10}, {18, SW18}, {-1, 0.f}, {19, SW58}, {21, SW74}, {22, SW50}, {23, SW58}}};
__device__ static constexpr SmTap kSm1d[2][16] = {
    {{1, SV73}, {2, SV65}, {5, SV65}, {6, SV25}, {7, SV17}, {8, SV41}, {10, SV49}, {11, SV9},
     {12, SV1}, {13, SV25}, {15, SV65}, {16, SV25}, {17, SV17}, {18, SV41}, {21, SV73}, {22, SV65}},
    {{2, SV65}, {3, SV73}, {6, SV41}, {7, SV17}, {8, SV25}, {9, SV65}, {11, SV25}, {12, SV1},
     {13, SV9}, {14, SV49}, {16, SV41}, {17, SV17}, {18, SV25}, {19, SV65}, {22, SV65}, {23, SV73}}};

// RNE: the x86 back-ends' as_i32 (cvtps2dq: round to nearest even, jxl_simd/src/x86_64/avx.rs:580) instead of the
// truncation of the scalar / NEON / wasm ones (scalar.rs:178): what a reference build running on an x86 host produces
template <bool TWO_D, int WHICH, bool RNE>
__device__ __forceinline__ int32_t smooth_eval(const float (&n)[25]) {
  float part[4];
#pragma unroll
  for (int g = 0; g < 4; g++) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      constexpr const SmTap* t = TWO_D ? kSm2d[WHICH] : kSm1d[WHICH & 1];
      const SmTap tp = t[4 * g + k];
      if (tp.n >= 0) acc = __fmaf_rn(n[tp.n], tp.w, acc);
    }
    part[g] = acc;
  }
  const float sum = __fadd_rn(__fadd_rn(part[0], part[1]), __fadd_rn(part[2], part[3]));
  const float biased = __fadd_rn(sum, copysignf(0.5f, sum));
  return RNE ? (int32_t)rintf(biased) : (int32_t)biased;
}

#define JXLH_SM_TX 64
#define JXLH_SM_WAVES 4
#define JXLH_SM_RPT 8                                 // consecutive rows one thread walks with a sliding window
#define JXLH_SM_TY (JXLH_SM_WAVES * JXLH_SM_RPT)      // average rows per workgroup: 20 window rows for 16 (1.25x)
// KIND 0: horizontal (out = 2 in_x), 1: vertical, 2: both.  nx x ny = average samples the rectangle covers.
template <int KIND, bool PAIR, bool RNE>
__global__ __launch_bounds__(JXLH_SM_TX* JXLH_SM_WAVES) void k6_smooth_unsqueeze(
    const int32_t* __restrict__ in, size_t in_stride, int in_w, int in_h, int cx0, int cy0, int32_t* __restrict__ out,
    size_t out_stride, int out_w, int out_h, int nx, int ny) {
  __shared__ float tile[JXLH_SM_TY + 4][JXLH_SM_TX + 4 + 1];
  const int tid = threadIdx.x;
  const int bx = blockIdx.x * JXLH_SM_TX, by = blockIdx.y * JXLH_SM_TY;
  // one wave per window row (row arithmetic is wave-uniform), lane = column; lanes 0..3 also fetch the 4 extra columns.
  // rows mirror ( -1 -> 0, h -> h - 1 ), columns clamp: load_row_to_scratch, step.rs:386-416
  const int wave = __builtin_amdgcn_readfirstlane(tid / JXLH_SM_TX), lane = tid % JXLH_SM_TX;
  for (int r = wave; r < JXLH_SM_TY + 4; r += JXLH_SM_WAVES) {
    int y = cy0 + by + r - 2;
    y = in_h == 1 ? 0 : (y < 0 ? -y - 1 : (y >= in_h ? 2 * in_h - 1 - y : y));
    y = min(max(y, 0), in_h - 1);  // only rows no live thread reads can still be outside
    const int32_t* row = in + (size_t)y * in_stride;
    const int x = cx0 + bx + lane - 2;
    tile[r][lane] = (float)row[min(max(x, 0), in_w - 1)];
    if (lane < 4) tile[r][JXLH_SM_TX + lane] = (float)row[min(max(x + JXLH_SM_TX, 0), in_w - 1)];
  }
  __syncthreads();
  const int ix = bx + lane;
  if (ix >= nx) return;
  const bool both = 2 * ix + 1 < out_w;
  float win[5][5];  // window rows; the first four are loaded once, then one new row per step
#pragma unroll
  for (int r = 0; r < 4; r++)
#pragma unroll
    for (int c = 0; c < 5; c++) win[r][c] = tile[wave * JXLH_SM_RPT + r][lane + c];
#pragma unroll
  for (int k = 0; k < JXLH_SM_RPT; k++) {
    const int ly = wave * JXLH_SM_RPT + k, iy = by + ly;
    if (iy >= ny) return;  // wave-uniform
#pragma unroll
    for (int c = 0; c < 5; c++) win[4][c] = tile[ly + 4][lane + c];
    float n[25];
#pragma unroll
    for (int r = 0; r < 5; r++)
#pragma unroll
      for (int c = 0; c < 5; c++) n[KIND == 1 ? 5 * c + r : 5 * r + c] = win[r][c];
    if (KIND == 2) {
      const int32_t o00 = smooth_eval<true, 0, RNE>(n), o01 = smooth_eval<true, 1, RNE>(n);
      const int32_t o10 = smooth_eval<true, 2, RNE>(n), o11 = smooth_eval<true, 3, RNE>(n);
      int32_t* p0 = out + (size_t)(2 * iy) * out_stride + 2 * ix;
      if (PAIR && both) {  // PAIR: base and stride keep every sample pair 8-byte aligned
        *(int2*)p0 = make_int2(o00, o01);
        if (2 * iy + 1 < out_h) *(int2*)(p0 + out_stride) = make_int2(o10, o11);
      } else {
        p0[0] = o00;
        if (both) p0[1] = o01;
        if (2 * iy + 1 < out_h) {
          p0[out_stride] = o10;
          if (both) p0[out_stride + 1] = o11;
        }
      }
    } else if (KIND == 0) {
      int32_t* p = out + (size_t)iy * out_stride + 2 * ix;
      const int32_t e = smooth_eval<false, 0, RNE>(n), o = smooth_eval<false, 1, RNE>(n);
      if (PAIR && both) {
        *(int2*)p = make_int2(e, o);
      } else {
        p[0] = e;
        if (both) p[1] = o;
      }
    } else {
      int32_t* p = out + (size_t)(2 * iy) * out_stride + ix;
      p[0] = smooth_eval<false, 0, RNE>(n);
      if (2 * iy + 1 < out_h) p[out_stride] = smooth_eval<false, 1, RNE>(n);
    }
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int c = 0; c < 5; c++) win[r][c] = win[r + 1][c];
  }
}

void launch_smooth_unsqueeze(hipStream_t s, int kind, const int32_t* in, size_t in_stride, int in_w, int in_h, int x0,
                             int y0, int32_t* out, size_t out_stride, int out_w, int out_h, bool cvt_rne) {
  const bool fx = kind != 1, fy = kind != 0;
  // the reference returns with the output untouched when the rectangle has no complete pair (squeeze.rs:921-923)
  if ((fx ? out_w / 2 : out_w) == 0 || (fy ? out_h / 2 : out_h) == 0) return;
  const int nx = fx ? (out_w + 1) / 2 : out_w, ny = fy ? (out_h + 1) / 2 : out_h;
  const int cx0 = fx ? x0 / 2 : x0, cy0 = fy ? y0 / 2 : y0;
  const dim3 grid((nx + JXLH_SM_TX - 1) / JXLH_SM_TX, (ny + JXLH_SM_TY - 1) / JXLH_SM_TY);
  const dim3 block(JXLH_SM_TX * JXLH_SM_WAVES);
  const bool pair = (uintptr_t)out % 8 == 0 && out_stride % 2 == 0;
#define JXLH_SM_LAUNCH1(K, P, R)                                                                                    \
  hipLaunchKernelGGL((k6_smooth_unsqueeze<K, P, R>), grid, block, 0, s, in, in_stride, in_w, in_h, cx0, cy0, out,   \
                     out_stride, out_w, out_h, nx, ny)
#define JXLH_SM_LAUNCH(K, P)                  \
  do {                                        \
    if (cvt_rne) JXLH_SM_LAUNCH1(K, P, true); \
    else JXLH_SM_LAUNCH1(K, P, false);        \
  } while (0)
  if (kind == 0) {
    if (pair) JXLH_SM_LAUNCH(0, true); else JXLH_SM_LAUNCH(0, false);
  } else if (kind == 1) {
    JXLH_SM_LAUNCH(1, false);
  } else {
    if (pair) JXLH_SM_LAUNCH(2, true); else JXLH_SM_LAUNCH(2, false);
  }
#undef JXLH_SM_LAUNCH
#undef JXLH_SM_LAUNCH1
}

}  // namespace jxlh
