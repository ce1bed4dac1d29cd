// Internal declarations shared by the HIP kernels and the C-ABI host code.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/jxl_hip.h"

namespace jxlh {
// 16-byte global accesses with a selectable cache policy (NT = streamed once: `nt` loads / stores)
typedef int jxlh_i32x4 __attribute__((ext_vector_type(4)));
typedef float jxlh_f32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ int4 gload_i4(const int* p) {
  const jxlh_i32x4* q = reinterpret_cast<const jxlh_i32x4*>(p);
  const jxlh_i32x4 v = NT ? __builtin_nontemporal_load(q) : *q;
  return make_int4(v.x, v.y, v.z, v.w);
}
template <bool NT>
__device__ __forceinline__ void gstore_i4(int* p, int4 o) {
  jxlh_i32x4 v = {o.x, o.y, o.z, o.w};
  jxlh_i32x4* q = reinterpret_cast<jxlh_i32x4*>(p);
  if (NT) __builtin_nontemporal_store(v, q);
  else *q = v;
}
template <bool NT>
__device__ __forceinline__ float4 gload_f4(const float* p) {
  const jxlh_f32x4* q = reinterpret_cast<const jxlh_f32x4*>(p);
  const jxlh_f32x4 v = NT ? __builtin_nontemporal_load(q) : *q;
  return make_float4(v.x, v.y, v.z, v.w);
}
template <bool NT>
__device__ __forceinline__ void gstore_f4(float* p, float4 o) {
  jxlh_f32x4 v = {o.x, o.y, o.z, o.w};
  jxlh_f32x4* q = reinterpret_cast<jxlh_f32x4*>(p);
  if (NT) __builtin_nontemporal_store(v, q);
  else *q = v;
}


constexpr int kBlockDim = 8;
constexpr int kGroupDim = 256;
constexpr int kGroupBlocks = 32;           // blocks per group side
constexpr int kGroupArea = 256 * 256;      // coefficients per channel per group
constexpr int kColorTileBlocks = 8;        // COLOR_TILE_DIM_IN_BLOCKS, color_correlation_map.rs:16
constexpr float kMinSigma = -3.90524291751269967465540850526868f;  // jxl/src/lib.rs:28
constexpr float kInvSigmaNum = -1.1715728752538099024f;            // features/epf.rs:26

// transform_map.rs:97-116
__host__ __device__ constexpr int covered_x(int t) {
  constexpr int lut[27] = {1, 1, 1, 1, 2, 4, 1, 2, 1, 4, 2, 4, 1, 1, 1, 1, 1, 1, 8, 4, 8, 16, 8, 16, 32, 16, 32};
  return lut[t];
}
__host__ __device__ constexpr int covered_y(int t) {
  constexpr int lut[27] = {1, 1, 1, 1, 2, 4, 2, 1, 4, 1, 4, 2, 1, 1, 1, 1, 1, 1, 8, 8, 4, 16, 16, 8, 32, 32, 16};
  return lut[t];
}
// quant_weights.rs:321-343
__host__ __device__ constexpr int quant_table_for_type(int t) {
  constexpr int lut[27] = {0, 1, 2, 3, 4, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 10, 10, 11, 12, 12, 13, 14, 14, 15, 16, 16};
  return lut[t];
}
// quant_weights.rs:1128-1132, floats per channel
__host__ __device__ constexpr int quant_table_size(int q) {
  constexpr int rx[17] = {1, 1, 1, 1, 2, 4, 1, 1, 2, 1, 1, 8, 4, 16, 8, 32, 16};
  constexpr int ry[17] = {1, 1, 1, 1, 2, 4, 2, 4, 4, 1, 1, 8, 8, 16, 16, 32, 32};
  return rx[q] * ry[q] * 64;
}

// util/mirror.rs:8-19
__host__ __device__ inline int mirror(int v, int s) {
  while (true) {
    if (v < 0) {
      v = -v - 1;
    } else if (v >= s) {
      v = s * 2 - v - 1;
    } else {
      return v;
    }
  }
}

// Device view of one VarDCT frame (passed by value to kernels).
struct FrameDev {
  int xsize, ysize;               // unpadded pixels
  int xblocks, yblocks;           // size in 8x8 blocks
  int xgroups, ygroups;
  int cmap_stride;                // colour tiles per row
  size_t plane_stride;            // floats; planes are yblocks*8 rows
  float* planes[3];               // X, Y, B
  float* tmp[3];                  // second set for out-of-place stages
  const int32_t* coeffs;          // ngroups * 3 * 65536
  const uint8_t* transform_map;   // stride xblocks
  const int32_t* raw_quant;       // stride xblocks
  const uint8_t* epf_map;         // stride xblocks
  const int8_t* ytox;             // stride cmap_stride
  const int8_t* ytob;
  const float* lf[3];             // stride xblocks (smoothed LF)
  float* inv_sigma;               // stride xblocks
  const float* tables;            // 17 tables back to back
  int table_offset[17];           // float offset of table q
  // scalars
  float inv_global_scale;         // 65536 / global_scale
  float x_dm, b_dm;               // 0.8^(qm_scale - 2)
  float quant_biases[4];
  float color_factor;             // as f32
  float base_x, base_b;
  float gab_k[3][3];              // per channel normalised w0, w1, w2 (gaborish.rs:20-27)
  float epf_channel_scale[3];
  float epf_sm[3], epf_bsm[3];    // per pass: sigma_scale*1.65 and *border_sad_mul
  int epf_iters;
  int gab;
  // Layout of planes[] as written by K1: 0 = raster (row pitch plane_stride); 1 = 8x8-tiled: block (bx, by) is the
  // 256-byte chunk (by*xblocks + bx)*64, and inside it pixel (x, y) lives at (y & 4)*8 + (x & 7)*4 + (y & 3) -- the
  // rows 0-3 of the eight columns (128 bytes), then the rows 4-7.  Used between K1 and the fused filter kernel: a
  // varblock completes whole 256-byte chunks instead of sharing 128-byte lines with its neighbours; an IDCT lane
  // (which owns a pixel column) stores two 16-byte pieces and the eight lanes of a block fill a 128-byte line per
  // store instruction; the filter's staging lane fetches 4 rows of a column (16 bytes), and both of its halos are
  // whole sectors: the 4 rows above / below a tile are one 128-byte half of each block, the 4 columns beside it 64
  // bytes of each half.  (Round 2's x*8 + y order made the row halo half of every 32-byte sector it touched and
  // K1's lanes write 32-byte pieces: K1 -5 %, filters -3 %, pipelined step -3.6 % with this order.)
  int tiled;
  // Chroma-subsampled frames (JPEG recompressions): channel c has (size >> shift) samples.  K1 writes such
  // a channel at the down-sampled block positions of its plane (same stride / tiling as a full plane);
  // blocks the channel does not hold go to the scrap tile at scrap_off (behind the plane proper).
  int subsampled;
  int hshift[3], vshift[3];
  int scrap_off;
  // Sparse coefficient input (null = K1 reads the dense slabs in `coeffs`).  sp_sorted: the frame's
  // {u16 pos; i16 val} pairs, bucketed by 64-coefficient slot inside every (group, channel) run;
  // sp_slot_start[(group*3 + c) * kSlotTable + s] = index of the first pair of slot s, entry
  // [.. + 1024] = end of the run.  group_dense[g] != 0 (set by k1_scan): the group holds varblocks of
  // the special / large classes, whose kernels read a dense slab (expanded for just those groups).
  const uint32_t* sp_sorted;
  const uint32_t* sp_slot_start;
  uint8_t* group_dense;
  // The slot-bucketed 2-byte form read IN PLACE (round 5; null = not this form): what jxlh_submit_groups_slots
  // uploaded, untouched -- se_entries: u16 = (position & 63) | (value & 1023) << 6 in (group, channel, slot) order;
  // se_counts[(group*3 + c) * 1024 + s]: entries of slot s; se_runs[group*3 + c] = {frame-wide index of the run's
  // first entry, entries in the run}.  k1_scan turns the counts into per-varblock entry ranges (work-list side items),
  // the class kernels read the entries themselves: no unpack pass, no slot tables.
  const uint16_t* se_entries;
  const uint8_t* se_counts;
  const uint2* se_runs;
  // group_route[g] != 0 (nullable; host-written): group g is NOT read from its entries -- its coefficients live in its
  // dense slab (it was submitted as a slab or as plain pairs, carried a value outside 10 bits, or added a pass to
  // earlier content) while the rest of the frame is read in place.  k1_scan sends such a group's DCT-class varblocks to
  // WorkLists::ditems.
  const uint8_t* group_route;
  // host hint from the entries per coefficient of the frame's in-place groups (d1 content: ~0.086).  2 (> 0.25): the
  // 8x8 class runs its over-depth batches inline instead of leaving them to the fallback launch; >= 1 (> 0.125): the
  // 16..32-point classes take the dense dequantisation pass outright (no direct launch for them)
  int se_dense_hint;
  int fb_epoch;   // launch number + 1 (set by launch_vardct_groups): the value that flags a batch for the fallback launch
  int k1_stats;   // the transforms count their dense-pass batches (jxlh_frame_k1_counters): on with kernel timing
  // The entries form may dequantise ONLY the coefficients that have an entry (everything else is +0.0f) when a zero
  // coefficient provably reconstructs to +0.0f and a non-zero one never to a zero: quant biases 0..2 in [1e-6, 1e6],
  // finite chroma-from-luma bases (host: se_direct_ok), every dequant weight in [1e-20, 1e20] (device: *tables_ok,
  // written by k_check_tables when the tables are set).  Otherwise -- and for varblocks with raw_quant == 0 or more entries than a lane holds -- the
  // class kernels take the dense dequantisation pass.
  int se_direct_ok;
  const int* tables_ok;
  // Strip path (k_strip.hip; null = the frame runs K1 -> planes -> filters): written by k1_scan<STRIP> -- per block a
  // descriptor {type | dx << 5 | dy << 7 | off64 << 9 | 1 << 31, raw_quant of its varblock}, per 64x64 tile who
  // reconstructs it (0 = the strip kernel, 1 = K1's class kernels); the scan also clears the strip kernel's progress
  // flags.  strip_all_closed: the host inspected the whole transform map and every tile is the strip kernel's.
  uint2* strip_desc;
  uint8_t* strip_mode;
  int* strip_flags;
  int strip_nflags;
  int strips, tile_rows;
  int strip_all_closed;
};
constexpr int kLfGroupBlocks = 256;  // an LF group is 2048 x 2048 pixels
constexpr int kSlotTable = 1025;  // 1024 slots of 64 coefficients per (group, channel) + end marker

// pixel (x, y) relative to a varblock's top-left pixel, for either layout:
//   addr = base + at(x, y)
struct PixLayout {
  int ystep8;     // raster: plane_stride      tiled: unused (see at())
  int ystep_blk;  // raster: 8 * plane_stride  tiled: xblocks * 64
  int tiled;
  // tiled: inside a block the rows 0-3 of all eight columns, then the rows 4-7: (y & 4) * 8 + (x & 7) * 4 + (y & 3)
  __host__ __device__ int xoff(int x) const { return tiled ? ((x >> 3) * 64 + (x & 7) * 4) : x; }
  __host__ __device__ int at(int x, int y) const {
    return xoff(x) + (y >> 3) * ystep_blk + (tiled ? (y & 4) * 8 + (y & 3) : (y & 7) * ystep8);
  }
};
__host__ __device__ inline PixLayout pix_layout(const FrameDev& f) {
  PixLayout l;
  l.tiled = f.tiled;
  l.ystep8 = f.tiled ? 1 : (int)f.plane_stride;
  l.ystep_blk = f.tiled ? f.xblocks * 64 : 8 * (int)f.plane_stride;
  return l;
}
// offset of the top-left pixel of block (gbx, gby)
__host__ __device__ inline int block_px_offset(const FrameDev& f, int gbx, int gby) {
  return f.tiled ? (gby * f.xblocks + gbx) * 64 : (int)((size_t)(gby * 8) * f.plane_stride + (size_t)gbx * 8);
}

// kernels (k_*.hip); all take an explicit stream
void launch_dequant_lf(hipStream_t s, const int32_t* qy, const int32_t* qx, const int32_t* qb, size_t qstride,
                       float* ox, float* oy, float* ob, size_t ostride, int w, int h, float fac_x, float fac_y,
                       float fac_b, float cfl_x, float cfl_b);
// dequant_lf of a chroma-subsampled frame: no chroma-from-luma, out = q * fac per channel (modular/mod.rs:877-893)
void launch_dequant_lf_plain(hipStream_t s, const int32_t* qy, const int32_t* qx, const int32_t* qb, size_t qstride,
                             float* ox, float* oy, float* ob, size_t ostride, int w, int h, float fac_x, float fac_y,
                             float fac_b);
// chroma_upsample.rs: sub-sampled channel (cw x ch samples, rows [sy0, sy1)) -> full channel, horizontal pass
// then vertical pass fused; writes are clipped to out_w x out_h
void launch_chroma_upsample(hipStream_t s, const float* src, float* dst, const PixLayout& slay,
                            const PixLayout& dlay, int hshift, int vshift, int cw, int ch, int sy0, int sy1, int out_w,
                            int out_h);
// upsample.rs: n = 2, 4, 8; kernels = n*n*25 expanded taps on the device; writes are clipped to out_w x out_h
void launch_upsample(hipStream_t s, int n, const float* in, size_t in_stride, int w, int h, const float* kernels,
                     float* out, size_t out_stride, int out_w, int out_h);
// noise synthesis (k_noise.hip)
void xorshift_jump_table(uint64_t out[16][128][2]);
void launch_noise_generate(hipStream_t s, float* const out[3], size_t stride, int w, int h, int tile_y0, int tile_y1,
                           uint32_t visible, uint32_t nonvisible, const void* jump_table_dev);
void launch_noise_apply(hipStream_t s, const float* const noise[3], size_t nstride, float* const planes[3], size_t pstride,
                        int w, int h, int y0, int y1, const float lut[8], float ytox, float ytob);
void launch_noise_convolve(hipStream_t s, const float* in, float* out, int w, int h);
void launch_noise_add(hipStream_t s, float* const planes[3], const float* const rnd[3], size_t n, const float lut[8],
                      float ytox, float ytob);
void launch_lf_smooth(hipStream_t s, const float* const in[3], float* const out[3], int w, int h,
                      const float lf_factors[3]);
void launch_sigma_map(hipStream_t s, const FrameDev& f, float epf_quant_mul, const float* sharp_lut);
// *ok = 1 iff all n dequant weights lie in [1e-20, 1e20] (see FrameDev::tables_ok)
void launch_check_tables(hipStream_t s, const float* tables, size_t n, int* ok);
// K1: work-list scan + per-class kernels over group rows [group_row0, group_row1).
// worklist_mem: device scratch of vardct_worklist_bytes(f) bytes.  It opens with TWO sets of class counters, both
// zeroed by vardct_worklist_reset() when a frame begins: launch n counts in set n & 1 and its scan kernel clears the
// other set for launch n + 1 (whose last readers, the kernels of launch n - 1, are behind it in stream order) -- no
// memset launch inside K1's serial sequence.  *launch_parity is the caller's per-context launch counter.
size_t vardct_worklist_bytes(const FrameDev& f);
void vardct_worklist_reset(hipStream_t s, void* worklist_mem, uint32_t* launch_parity);
// the counter set launch number `launch` counted in: *bytes = its size, *lines = counters in it (one per `*bytes / *lines`)
const void* vardct_worklist_counters(const void* worklist_mem, uint32_t launch, size_t* bytes, int* lines);
// dense_coeffs: writable alias of f.coeffs, used in sparse mode to expand the groups k1_scan flags
// group_list (device, n_list entries) replaces the row range by an explicit list of group ids when non-null
void launch_vardct_groups(hipStream_t s, const FrameDev& f, int group_row0, int group_row1,
                          void* worklist_mem, uint32_t* launch_parity, int* error_flag, int32_t* dense_coeffs,
                          const int* group_list = nullptr, int n_list = 0, bool has_special = true,
                          bool has_large = true, int n_dense_route = 0);
// k_strip.hip: dequantisation + IDCT + the frame's filter stages in one persistent kernel, result in f.tmp (raster).
// Runs behind launch_vardct_groups on a FrameDev with the strip_* members set.  false = stage list not covered.
int strip_resident_workgroups(int cu_count);
int strip_tile_rows(const FrameDev& f);
int strip_strips(const FrameDev& f);
size_t strip_xchg_floats(const FrameDev& f);
size_t strip_flag_ints(const FrameDev& f, int bands);
bool launch_strip(hipStream_t s, const FrameDev& f, const uint2* desc, const uint8_t* tile_mode, float* xchg, int* flags,
                  int bands, int* error_flag, float deadline_s);
void launch_gaborish(hipStream_t s, const float* in, float* out, int w, int h, size_t stride, float k0, float k1,
                     float k2, int y0, int y1);
struct EpfArgs {
  const float* in[3];
  float* out[3];
  const float* inv_sigma;
  size_t stride, sigma_stride;
  int w, h;
  float scale[3];
  float sm, bsm;
};
void launch_epf(hipStream_t s, int stage, const EpfArgs& a, int y0, int y1);
// Gaborish/EPF1/EPF2 of the frame's stage list in one LDS-tiled pass, planes -> tmp, output rows [y0, y1).
// Returns false when the stage list is not covered (EPF0, i.e. epf_iters == 3, or no stage at all).
int launch_fused_filters(hipStream_t s, const FrameDev& f, int y0, int y1);
// output stages (k_output.hip): XybParams of the reference (xyb.rs:147-163), same field order as
// jxlh_xyb_params
struct XybParamsDev {
  float mat[9], bias_cbrt[3], scaled_bias[3], intensity_scale;
};
// output colour handling of the read kernels: XybStage followed by one of the reference's transfer functions
// (render/stages/from_linear.rs:133-145), or the YCbCr / identity alternatives
enum { kTfLinear = 0, kTfSrgb = 1, kTfBt709 = 2, kTfPq = 3, kTfHlg = 4, kTfGamma = 5, kModeYcbcr = 6, kModeNone = 7 };
struct TfParamsDev {
  float param;   // PQ: intensity_target; HLG: OOTF exponent; gamma: exponent
  float lum[3];  // HLG: luminance_rgb
};
// K1's output of a chroma-subsampled frame: p[c] = the full plane (no shift) or the sub-sampled one (cw x ch samples)
struct SubPlanesDev {
  const float* p[3];
  int hs[3], vs[3], cw[3], ch[3];
};
void launch_ycbcr_sub_to_rgb(hipStream_t s, const SubPlanesDev& sp, size_t stride, int w, int y0, int rows, int channels,
                             int bits, void* out, size_t out_stride);
void launch_xyb_to_rgb8(hipStream_t s, const float* const planes[3], size_t stride, int w, int y0, int rows, int mode,
                        const XybParamsDev& q, const TfParamsDev& t, int channels, uint8_t* out, size_t out_stride);
void launch_xyb_to_rgb16(hipStream_t s, const float* const planes[3], size_t stride, int w, int y0, int rows, int mode,
                         const XybParamsDev& q, const TfParamsDev& t, int channels, uint16_t* out, size_t out_stride_elems);
// sparse coefficient transport (k_coeffs.hip): one descriptor per submitted group
struct SparseGroup {
  uint32_t group;   // group id
  uint32_t offset;  // index of the group's first pair in the pair buffer (X pairs, then Y, then B)
  uint32_t n[3];    // pairs per channel
  uint32_t flags;   // bit 0: add the pairs to the group's current slab instead of starting from zero
};
// only_flagged (nullable): expand a group only if only_flagged[group] != 0
void launch_pack_pairs8(hipStream_t s, const uint16_t* pos, const int8_t* val, size_t n, uint32_t* pairs);
// the 2-byte form (jxlh_submit_groups_sparse4): n_runs = groups of the batch x 3, desc = 4 words per run, see k_coeffs.hip
void launch_pack_pairs4(hipStream_t s, const uint16_t* entries, const uint16_t* seg_counts, const uint16_t* pos8,
                        const int8_t* val8, const uint32_t* desc, int n_runs, uint32_t* pairs);
void launch_expand_sparse(hipStream_t s, int32_t* coeffs, const uint32_t* pairs, const SparseGroup* groups,
                          int n_groups, const uint2* wide, uint32_t n_wide, const uint8_t* only_flagged);
// dense slabs of the groups with flags[g] != 0 from the bucketed pairs (all groups of the frame)
void launch_expand_sorted(hipStream_t s, int32_t* coeffs, const uint32_t* sorted, const uint32_t* slot_start,
                          const uint8_t* flags, int n_groups);
// ---- the slot-bucketed form kept as uploaded (se_* members of FrameDev)
constexpr int kSlotsPerRun = 1024;  // 64-coefficient slots per (group, channel)
// 12-bit entries (JXLH_GROUP_ENTRIES12), two per three bytes -> u16 entries; n_pairs = entries / 2 of the whole batch
void launch_unpack_entries12(hipStream_t s, const uint8_t* bytes, size_t n_pairs, uint16_t* entries);
// groups with flags[g] != 0: their entries -> {u16 pos; i16 val} pair words at the same indices of `pairs` (the form
// the sort / the zero-fill + scatter read), for epochs in which not every group arrived slot-bucketed
void launch_entries_to_pairs(hipStream_t s, const uint16_t* entries, const uint8_t* counts, const uint2* runs,
                             const uint8_t* flags, int n_groups, uint32_t* pairs);
// dense slabs of the groups with flags[g] != 0 from their slot-bucketed entries
void launch_expand_entries(hipStream_t s, int32_t* coeffs, const uint16_t* entries, const uint8_t* counts,
                           const uint2* runs, const uint8_t* flags, int n_groups);
// counting sort of every (group, channel) run by slot (pos >> 6) + the slot start table
void launch_sort_sparse(hipStream_t s, const uint32_t* pairs, const SparseGroup* groups, int n_groups,
                        uint32_t* sorted, uint32_t* slot_start);
// counts floats with bit patterns in [lo_bits, hi_bits) whose fast reciprocal differs from 1.0f / w
void launch_selftest_recip(hipStream_t s, uint32_t lo_bits, uint32_t hi_bits, unsigned long long* mismatches);
void launch_transform_to_pixels(hipStream_t s, int type, uint32_t n, const float* coeffs, const float* lf,
                                float* pixels);
void launch_rct(hipStream_t s, int32_t* p0, int32_t* p1, int32_t* p2, size_t n, int op, int perm);
// rows of w samples at a pitch of `stride` samples, in one launch (padding untouched)
void launch_rct_rows(hipStream_t s, int32_t* p0, int32_t* p1, int32_t* p2, uint32_t w, uint32_t h, size_t stride, int op,
                     int perm);
// out_channel_stride: samples between the channel planes of `out` (0 = n, contiguous planes)
void launch_palette(hipStream_t s, const int32_t* index, size_t n, const int32_t* palette, int num_colors,
                    size_t palette_stride, int nb_channels, int bit_depth, int32_t* out, size_t out_channel_stride = 0);
void launch_palette_delta(hipStream_t s, const int32_t* index, int w, int h, const int32_t* palette, int num_colors,
                          int num_deltas, size_t palette_stride, int nb_channels, int bit_depth, int predictor,
                          int32_t* out, int* progress);
int palette_delta_bands(int h);
void launch_palette_wp(hipStream_t s, const int32_t* index, int w, int h, const int32_t* palette, int num_colors,
                       int num_deltas, size_t palette_stride, int nb_channels, int bit_depth, const uint32_t header[11],
                       int32_t* out, int* progress, int32_t* wp_rows);
void launch_i32_to_rgb8(hipStream_t s, const int32_t* const planes[3], size_t stride, int w, int h, int32_t mult,
                        int32_t maxv, int channels, uint8_t* out, size_t out_stride);
void launch_modular_to_f32(hipStream_t s, const int32_t* in, size_t n, float scale, float* out);
// floating-point samples: a `bits`-bit float with exp_bits exponent bits in an integer -> binary32
void launch_float_samples_to_f32(hipStream_t s, const int32_t* in, size_t n, uint32_t bits, uint32_t exp_bits, float* out);
// BitDepth as the ABI carries it: bits_per_sample | exponent_bits_per_sample << 8 (0 = integer samples).  A float format
// needs 2 <= exponent bits <= 8 and at least one, at most 23 mantissa bits (headers/bit_depth.rs).
inline bool bit_depth_ok(uint32_t packed, uint32_t max_int_bits) {
  const uint32_t bits = packed & 0xffu, eb = packed >> 8;
  if (eb == 0) return packed == bits && bits >= 1 && bits <= max_int_bits;
  return (packed >> 16) == 0 && bits <= 32 && eb >= 2 && eb <= 8 && bits >= eb + 2 && bits - eb - 1 <= 23;
}
void launch_modular_xyb_to_f32(hipStream_t s, const int32_t* y, const int32_t* x, const int32_t* b, size_t n,
                               const float scale[3], float* ox, float* oy, float* ob);
// n_planes (<= 3) planes of identical geometry in one launch (the channels of one squeeze step)
void launch_unsqueeze(hipStream_t s, int horizontal, int n_planes, const int32_t* const avg[], size_t avg_stride,
                      const int32_t* const res[], size_t res_stride, uint32_t out_w, uint32_t out_h,
                      int32_t* const out[], size_t out_stride);
// fused unsqueeze of three planes + inverse RCT; false = not applicable (plane too large), nothing launched
// consecutive streamed unsqueeze steps as one dataflow launch (k6_unsqueeze_flow): step i + 1's averages are step i's
// outputs (distinct planes per step); `scratch` holds unsqueeze_flow_words() ints, `error` one int that stays 0 unless a
// wait inside the launch outlasted deadline_s
struct FlowStep {
  int horizontal;
  const int32_t* avg[3];
  size_t avg_stride;
  const int32_t* res[3];
  size_t res_stride;
  uint32_t out_w, out_h;
  int32_t* out[3];
  size_t out_stride;
};
bool unsqueeze_tiled_eligible(int horizontal, uint32_t out_w, uint32_t out_h, size_t avg_stride, size_t res_stride,
                              size_t out_stride);
int unsqueeze_flow_max_steps();
size_t unsqueeze_flow_words(int n_planes, int n_steps, const FlowStep* steps);
void launch_unsqueeze_flow(hipStream_t s, int n_planes, int n_steps, const FlowStep* steps, int* scratch, int* error,
                           float deadline_s, unsigned long long* prof);
bool launch_unsqueeze_rct(hipStream_t s, int horizontal, const int32_t* const avg[3], size_t avg_stride,
                          const int32_t* const res[3], size_t res_stride, uint32_t out_w, uint32_t out_h,
                          int32_t* const out[3], size_t out_stride, int op, int perm);
// several consecutive squeeze steps of small planes (all sides <= 128) in one launch; res[i * 3 + p] = residual plane
// of level i, plane p.  false = the chain does not qualify, nothing launched
#define JXLH_SQL_MAX 128     // k6_unsqueeze_levels: largest plane side handled in LDS
#define JXLH_SQL_LEVELS 16  // ... and the most levels one launch takes
bool launch_unsqueeze_levels(hipStream_t s, int n_planes, int n_levels, const int* horizontal, const uint32_t* out_w,
                             const uint32_t* out_h, const int32_t* const* res, const size_t* res_stride,
                             const int32_t* const base[], size_t base_stride, uint32_t base_w, uint32_t base_h,
                             int32_t* const out[], size_t out_stride);
// smooth_{h,v,2d}_unsqueeze on a rectangle of the output channel; `in` is the whole average channel
void launch_smooth_unsqueeze(hipStream_t s, int kind, const int32_t* in, size_t in_stride, int in_w, int in_h, int x0,
                             int y0, int32_t* out, size_t out_stride, int out_w, int out_h, bool cvt_rne);

}  // namespace jxlh
