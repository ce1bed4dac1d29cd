/*
 * jxl_hip.h -- C ABI of the MI355X-native JPEG XL reconstruction hot path.
 *
 * This is the drop-in boundary for libjxl/jxl-rs (v0.6.0).  jxl-rs has no FFI of
 * its own (`#![deny(unsafe_code)]`, jxl/src/lib.rs:6): the path sits behind Rust
 * traits, so each entry point below names the reference call site it replaces;
 * INTEGRATION.md shows the Rust `extern "C"` block and the
 * `HipRenderPipeline: RenderPipeline` shim that binds them.
 *
 * Conventions
 *   - every function returns jxlh_status (0 = OK, <0 = error class); nothing
 *     throws or aborts across the ABI (reference: Result<T, Error>, error.rs:15-276)
 *   - all pointers are caller-owned; a pointer argument may be a host pointer or a
 *     device (HIP) pointer -- uploads/downloads use hipMemcpyDefault
 *   - plane descriptor == RawImageBuffer / JxlOutputBuffer::new_from_ptr
 *     (jxl/src/image/internal.rs:15-31, image/output_buffer.rs:32-52)
 *   - frame-level calls are single-threaded (any ONE thread at a time, not necessarily the
 *     same one); the jxlh_submit_group* calls are re-entrant per `slot` (one HIP stream + one
 *     pinned staging slab per slot, mirroring PerThreadStorage,
 *     jxl/src/util/per_thread_storage.rs:13-60).  Every entry point makes the context's device
 *     the calling thread's current HIP device, so pool threads need no device set-up
 *   - channel order is X, Y, B everywhere (pipeline channels 0, 1, 2)
 */
#ifndef JXL_HIP_H_
#define JXL_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JXLH_ABI_VERSION 6  /* additions only since 6: round 6 added the jxlh_host_*, jxlh_slot_writer_* and
                               jxlh_ctx_wait_* / _record_event entry points */
#define JXLH_NUM_TRANSFORMS 27   /* HfTransformType::CARDINALITY, transform_map.rs:59-61 */
#define JXLH_NUM_QUANT_TABLES 17 /* NUM_QUANT_TABLES, quantizer.rs:11 */
#define JXLH_GROUP_DIM 256       /* GROUP_DIM, jxl/src/lib.rs:24-26 */

typedef int32_t jxlh_status;
enum {
  JXLH_OK = 0,
  JXLH_ERR_INVALID_ARGUMENT = -1, /* -> Error::Gpu(InvalidArgument) */
  JXLH_ERR_OUT_OF_MEMORY = -2,    /* hipMalloc failed (reference: try_reserve, group.rs:47-55) */
  JXLH_ERR_DEVICE = -3,           /* any other HIP runtime error; see jxlh_last_error */
  JXLH_ERR_BAD_STATE = -4,        /* call order violated (e.g. submit before frame_begin) */
  JXLH_ERR_INVALID_TRANSFORM = -5,/* transform_map holds an id >= 27 (Error::InvalidVarDCTTransform) */
  JXLH_ERR_UNSUPPORTED = -6,      /* valid stream feature outside the device path (caller falls back): band or sharded
                                     runs and group re-renders of upsampled frames, frames beyond 2^31 coefficients,
                                     predictor 6 in jxlh_palette_delta (use jxlh_palette_delta_wp) */
  JXLH_ERR_INVALID_BLOCK_SIZE = -7, /* a varblock larger than 8x8 in a chroma-subsampled frame
                                      (Error::InvalidBlockSizeForChromaSubsampling, frame/modular/mod.rs:1058-1060) */
  JXLH_ERR_BLOCK_OUT_OF_BOUNDS = -8 /* a varblock crosses its group's or the frame's edge, or the first-block flags of
                                       a group cover more than its 1024 blocks (Error::HFBlockOutOfBounds,
                                       frame/modular/mod.rs:1061-1064); the offending varblocks are not reconstructed */
};

typedef struct jxlh_ctx jxlh_ctx;

/* RawImageBuffer (image/internal.rs:15-31). */
typedef struct {
  void* ptr;
  size_t bytes_per_row;
  size_t num_rows;
  size_t bytes_between_rows;
} jxlh_plane;

/* Everything the device needs that the reference keeps in FrameHeader /
 * LfGlobalState / HfGlobalState.  Field by field:
 *   xsize, ysize .......... frame_header.size_upsampled() == pipeline image size
 *                           (frame/render.rs:541-545)
 *   global_scale, quant_lf . QuantizerParams (frame/quantizer.rs:56-85)
 *   lf_quant_factors ....... LfQuantFactors::quant_factors (quantizer.rs:14-51)
 *   quant_biases ........... OpsinInverseMatrix::quant_biases (headers/transform_data.rs:30-31)
 *   x_qm_scale, b_qm_scale . FrameHeader (headers/frame_header.rs:308-315); the device computes
 *                           0.8^(scale-2) exactly like group.rs:395-396
 *   color_factor .. ytob_lf  ColorCorrelationParams (frame/color_correlation_map.rs:21-94)
 *   gab .. epf_border_sad_mul RestorationFilter (headers/frame_header.rs:146-233)
 *   do_lf_smoothing ........ FrameHeader::should_do_adaptive_lf_smoothing (:496-500)
 *   hshift, vshift ......... FrameHeader::hshift(c) / vshift(c) (headers/frame_header.rs:501-512), channel order
 *                           X/Cb, Y, B/Cr, each 0 or 1: channel c holds (size >> shift) samples (4:2:0, 4:2:2,
 *                           4:4:0 JPEG recompressions).  Such frames are limited to 8x8 transforms
 *                           (frame/modular/mod.rs:1058-1060; JXLH_ERR_INVALID_BLOCK_SIZE otherwise), their size
 *                           in blocks is rounded up to whole blocks of the coarsest channel
 *                           (FrameHeader::size_blocks, :564-569), the LF samples of a sub-sampled channel
 *                           sit in the top-left corner of each LF group's rectangle (frame/group.rs:485-504,
 *                           modular/mod.rs:877-893) and the LF dequantisation skips chroma-from-luma.
 *                           The transforms reconstruct every channel at its own resolution
 *                           (frame/group.rs:223-250) and the chroma upsampling stages
 *                           (render/stages/chroma_upsample.rs, frame/render.rs:569-576) run before Gaborish /
 *                           EPF, so everything downstream sees full-resolution planes.
 *   upsampling ............. FrameHeader::upsampling (1, 2, 4 or 8; 0 means 1).  xsize / ysize are then the CODED
 *                           size FrameHeader::size() = ceil(size_upsampled / upsampling) (:555-561) and
 *                           xsize_upsampled / ysize_upsampled the image size after the Upsample2x/4x/8x stages
 *                           (0 = xsize * upsampling); the stages run on the three colour channels after the
 *                           filters (frame/render.rs:655-671) and every jxlh_frame_read_* call then returns the
 *                           upsampled image.  Such a frame is run whole (JXLH_ERR_UNSUPPORTED for a band).
 *   noise, noise_lut ....... FrameHeader::has_noise() (:486-488) and Noise::lut (features/noise.rs:9-20).  The frame then
 *                           ends with the reference's noise synthesis at the result's resolution: random planes
 *                           from Xorshift128Plus seeded per 256x256 tile with (visible_frame_index,
 *                           nonvisible_frame_index, x0, y0) (frame/decode.rs:578-668), ConvolveNoiseStage on
 *                           each, AddNoiseStage with ColorCorrelationParams::y_to_x_lf / y_to_b_lf
 *                           (render/stages/noise.rs, frame/render.rs:673-683)
 *   epf_sigma_for_modular .. RestorationFilter field used when EPF runs on a Modular frame
 *                           (features/epf.rs:81-84): jxlh_modular_frame_filters; VarDCT frames ignore it
 */
typedef struct {
  uint32_t abi_version; /* JXLH_ABI_VERSION */
  uint32_t xsize, ysize;
  uint32_t global_scale, quant_lf;
  float lf_quant_factors[3];
  float quant_biases[4];
  uint32_t x_qm_scale, b_qm_scale;
  uint32_t color_factor;
  float base_correlation_x, base_correlation_b;
  int32_t ytox_lf, ytob_lf;
  uint32_t gab;
  float gab_w1[3], gab_w2[3];
  uint32_t epf_iters;
  float epf_sharp_lut[8];
  float epf_channel_scale[3];
  float epf_quant_mul, epf_pass0_sigma_scale, epf_pass2_sigma_scale, epf_border_sad_mul;
  uint32_t do_lf_smoothing;
  uint32_t flags; /* JXLH_FRAME_* */
  uint32_t hshift[3], vshift[3];
  float epf_sigma_for_modular;
  uint32_t upsampling;
  uint32_t xsize_upsampled, ysize_upsampled;
  uint32_t noise;
  float noise_lut[8];
  uint32_t visible_frame_index, nonvisible_frame_index;
} jxlh_frame_params;

enum {
  JXLH_FRAME_UNFUSED_FILTERS = 1u << 0, /* run Gaborish/EPF as one kernel per stage (debug/parity) */
  JXLH_FRAME_EXPAND_SPARSE = 1u << 1,   /* always expand sparse submissions into dense slabs before the
                                           transforms instead of letting them read the pairs (debug/parity) */
  JXLH_FRAME_STRIP = 1u << 2,           /* whole-frame runs go through the single strip kernel (dequantisation, IDCT and
                                           the filter stages in one persistent launch, no intermediate planes in HBM:
                                           2.35 GB of traffic per 8K frame instead of 3.58) instead of transforms ->
                                           planes -> fused filters.  Bit-identical output; slower on MI355X today
                                           (both forms are bound by instruction issue), hence
                                           opt-in */
  JXLH_FRAME_DENSE_DEQUANT = 1u << 3,   /* a frame resident in the slot-bucketed form (jxlh_submit_groups_slots): the
                                           transforms always dequantise every coefficient position instead of only the
                                           positions that have an entry (debug/parity: both give the same bits) */
};

/* Header defaults of the reference (RestorationFilter / ColorCorrelationParams /
 * quant_biases defaults), for callers that only override a few fields. */
jxlh_status jxlh_default_frame_params(jxlh_frame_params* p, uint32_t xsize, uint32_t ysize);

/* ---------------------------------------------------------------- context */
/* n_slots = number of host threads that will call jxlh_submit_group concurrently
 * (the JxlParallelRunner's thread count, jxl/src/api/mod.rs:77-81). */
jxlh_status jxlh_ctx_create(int32_t device_ordinal, int32_t n_slots, jxlh_ctx** out);
void jxlh_ctx_destroy(jxlh_ctx* ctx);
const char* jxlh_status_string(jxlh_status s);
/* human-readable detail of the last JXLH_ERR_DEVICE on this context (thread-unsafe, debugging) */
const char* jxlh_last_error(const jxlh_ctx* ctx);
/* pinned host memory for coefficient slabs (replaces VarDctBuffers::coeffs_storage,
 * frame/group.rs:27-67) */
jxlh_status jxlh_alloc_pinned(jxlh_ctx* ctx, size_t bytes, void** out);
jxlh_status jxlh_free_pinned(jxlh_ctx* ctx, void* p);

/* ---------------------------------------------------------------- VarDCT frame */
/* Replaces Frame::from_header_and_toc's LF / HfMetadata allocation
 * (frame/decode.rs:172-204) + prepare_render_pipeline (frame/render.rs:907). */
jxlh_status jxlh_frame_begin(jxlh_ctx* ctx, const jxlh_frame_params* p);

/* CustomTransformData::weights2 / weights4 / weights8 (headers/transform_data.rs:337-344): 15 / 55 / 210 weights
 * of the 2x / 4x / 8x upsampling kernels; NULL selects the codestream defaults (DEFAULT_KERN_*, :34-333).
 * Per decoder like the reference's file header: persists across frames; callable outside a frame. */
jxlh_status jxlh_set_upsampling_weights(jxlh_ctx* ctx, const float* weights2, const float* weights4,
                                        const float* weights8);

/* HfGlobalState::dequant_matrices (frame/quant_weights.rs:347-351): 17 tables, table t holds
 * 3 * n[t] inverse weights, channel-major (matrix(type, c), :1081-1086). */
jxlh_status jxlh_frame_set_dequant_tables(jxlh_ctx* ctx, const float* const tables[JXLH_NUM_QUANT_TABLES],
                                          const size_t n[JXLH_NUM_QUANT_TABLES]);

/* decode_vardct_lf -> dequant_lf (frame/modular/mod.rs:837-929), one LF-group rect at a time.
 * Rect in blocks.  qy/qx/qb are the three modular channels in coded order (Y, X, B), row stride
 * `stride` samples.  Runs K0a on the device. */
jxlh_status jxlh_frame_set_lf_quantized(jxlh_ctx* ctx, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h,
                                        const int32_t* qy, const int32_t* qx, const int32_t* qb,
                                        size_t stride, uint32_t extra_precision);
/* Alternative: LF already dequantised by the host (lf_image, frame/mod.rs). */
jxlh_status jxlh_frame_set_lf(jxlh_ctx* ctx, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h,
                              const float* x, const float* y, const float* b, size_t stride);

/* decode_hf_metadata (frame/modular/mod.rs:984-1081): HfMetadata maps for a rect in blocks
 * (frame/mod.rs:169-176).  ytox/ytob cover ceil(w/8) x ceil(h/8) colour tiles starting at
 * (x0/8, y0/8); x0, y0 must be multiples of 8. */
jxlh_status jxlh_frame_set_hf_meta(jxlh_ctx* ctx, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h,
                                   const uint8_t* transform_map, const int32_t* raw_quant,
                                   const uint8_t* epf_map, size_t map_stride, const int8_t* ytox,
                                   const int8_t* ytob, size_t cmap_stride);

/* Replaces the `if let Some(pixels)` branch of decode_vardct_group (frame/group.rs:579-611)
 * + pipeline.set_buffer_for_group for channels 0..2 (frame/decode.rs:776-784).
 * coeffs: 3 * 65536 i32 (X, Y, B planes of the group, varblocks back to back in raster order
 * of their top-left block, group.rs:437-440, :612).  Asynchronous: returns after enqueueing the
 * H2D copy on slot's stream; the slab may be reused after jxlh_slot_wait(slot). */
jxlh_status jxlh_submit_group(jxlh_ctx* ctx, int32_t slot, uint32_t group_id, const int32_t* coeffs,
                              uint32_t flags);
/* Flags of the submit calls.
 *   JXLH_GROUP_COMPLETE    set_buffer_for_group's `complete` (render/mod.rs:128-137): this is the last time the group
 *                          is submitted.  Without it the submission is a progressive pass: the frame can be rendered
 *                          with what has arrived (jxlh_frame_run), and groups that receive further passes later are
 *                          brought up to date with jxlh_frame_rerender_groups.
 *   JXLH_GROUP_ACCUMULATE  sparse forms only: the pairs are ADDED to the group's coefficients of the earlier passes
 *                          on the device (frame/group.rs:572 `+=` on Frame::hf_coefficients, frame/decode.rs:547-558)
 *                          instead of replacing them.  A dense slab always replaces the group's coefficients, so a
 *                          progressive caller that keeps dense slabs submits its accumulated slab.
 *   JXLH_GROUP_ENTRIES12   jxlh_submit_groups_slots only: the entries are 12 bits -- (position & 63) | (value & 63) << 6,
 *                          value in [-32, 31] -- packed two per three bytes (byte 0 = e0 & 255, byte 1 = e0 >> 8 |
 *                          (e1 & 15) << 4, byte 2 = e1 >> 4); every (group, channel) run holds an even number of them
 *                          (an odd run is closed with a zero update, counted in its last slot), so each run starts on
 *                          a byte.  1.5 bytes per update on the bus. */
enum { JXLH_GROUP_COMPLETE = 1u << 0, JXLH_GROUP_ACCUMULATE = 1u << 1, JXLH_GROUP_ENTRIES12 = 1u << 2 };

/* Sparse form of jxlh_submit_group (the dense i32 slab is ~90 % zeros at d1 and
 * its PCIe transfer bounds end-to-end decode).  The entropy loop of decode_vardct_group
 * (frame/group.rs:560-575: `coeffs[c][offset + order[k]] += v`) emits one pair per non-zero
 * coefficient instead of writing a dense slab:
 *   pairs   n[0] pairs of channel X, then n[1] of Y, then n[2] of B, contiguous; pos = index inside
 *           the channel's 65536-entry slab (the same index space as the dense form)
 *   wide    values that do not fit i16 (the reference stores i32), pos = channel * 65536 + index
 * Duplicate positions accumulate with wrapping i32 adds (multi-pass accumulation: put all passes'
 * updates of a group in ONE list -- a group may be submitted sparse once between two jxlh_frame_run
 * calls, JXLH_ERR_BAD_STATE otherwise).  When every group of the frame arrives this way the transforms
 * read the pairs directly (bucketed per varblock on the device); otherwise the device zero-fills the
 * groups' slabs and scatters the pairs.  Results are identical to the dense path bit for bit.  Asynchronous like jxlh_submit_group (H2D on the slot's stream;
 * pinned host memory from jxlh_alloc_pinned overlaps with compute). */
typedef struct jxlh_coeff16 {
  uint16_t pos;
  int16_t val;
} jxlh_coeff16;
typedef struct jxlh_coeff32 {
  uint32_t pos;
  int32_t val;
} jxlh_coeff32;
jxlh_status jxlh_submit_group_sparse(jxlh_ctx* ctx, int32_t slot, uint32_t group_id, const jxlh_coeff16* pairs,
                                     const uint32_t n[3], const jxlh_coeff32* wide, uint32_t n_wide,
                                     uint32_t flags);
/* The same for `count` groups decoded by one host thread: one H2D copy for all of them.
 * pairs holds the groups' pair runs back to back in the order of group_ids; n is count x 3;
 * wide positions carry the group: pos = (group_id * 3 + channel) * 65536 + index. */
jxlh_status jxlh_submit_groups_sparse(jxlh_ctx* ctx, int32_t slot, uint32_t count, const uint32_t* group_ids,
                                      const jxlh_coeff16* pairs, const uint32_t* n, const jxlh_coeff32* wide,
                                      uint32_t n_wide, uint32_t flags);
/* The same with 3 bytes per update on the bus: positions (u16) and values (i8) as two arrays in the order of the
 * pairs above; updates whose value does not fit 8 bits go to `wide` like those that do not fit 16 above.  At d1 nearly
 * every coefficient fits, and the host-to-device copy is what bounds the decode-to-device rate. */
jxlh_status jxlh_submit_groups_sparse8(jxlh_ctx* ctx, int32_t slot, uint32_t count, const uint32_t* group_ids,
                                       const uint16_t* pos, const int8_t* val, const uint32_t* n,
                                       const jxlh_coeff32* wide, uint32_t n_wide, uint32_t flags);
/* The same with 2 bytes per update on the bus (round 4): a channel's 65536 coefficient positions are cut into 16
 * SEGMENTS of 4096; an update is one u16 = (position inside its segment) | (value & 15) << 12 with the value in
 * [-8, 7] (two's complement nibble), and seg_counts[(i * 3 + c) * 16 + s] says how many entries group i, channel c,
 * segment s has; `entries` holds them in that order (group, channel, segment), any order inside a segment.  What a
 * decoder does at `coeffs[idx] += coeff` (frame/group.rs:572): append (idx & 4095) | coeff << 12 to the bucket of
 * idx >> 12.  Updates whose value does not fit the nibble go to the 3-byte overflow arrays pos8 / val8 (positions
 * inside the channel slab as in jxlh_submit_groups_sparse8; n8 is count x 3, may be NULL when there is none), values
 * beyond 8 bits to `wide`.  Every update is an addition into the group's slab, so the split changes nothing. */
jxlh_status jxlh_submit_groups_sparse4(jxlh_ctx* ctx, int32_t slot, uint32_t count, const uint32_t* group_ids,
                                       const uint16_t* entries, const uint16_t* seg_counts, const uint16_t* pos8,
                                       const int8_t* val8, const uint32_t* n8, const jxlh_coeff32* wide,
                                       uint32_t n_wide, uint32_t flags);
/* Slot-bucketed form: 2 bytes per update, read by the transforms IN PLACE (round 5: no device-side sort, no unpacking
 * pass, no slot tables).  A channel's 65536 positions are 1024 SLOTS of 64 coefficients (the unit varblock coefficient
 * offsets are counted in, frame/group.rs:612: an 8x8 block is one slot, a 16x16 varblock four); an update is one u16 =
 * (position & 63) | (value & 1023) << 6 with the value in [-512, 511]; slot_counts[(i * 3 + c) * 1024 + s] (u8) says how
 * many updates group i, channel c, slot s has and `entries` holds them in that order, any order inside a slot;
 * n[3 * i + c] = the channel's total.  A decoder appends to the slot buckets of the varblock it is decoding and flushes
 * them when the varblock ends (frame/group.rs:557-575).  Positions may repeat (several passes' updates in one list): they
 * add up as integers before the dequantisation, like `coeffs[i] += v`.  A slot-count table that disagrees with n stays
 * inside its run: counts beyond n are cut at the run's end, entries the table does not cover are ignored.
 *   When EVERY group of the frame arrives in this form (and nothing is added to earlier passes), jxlh_frame_run reads the
 * upload as it is: the transforms dequantise only the positions that have an entry (everything else is the +0.0f the
 * reference computes for a zero coefficient; varblocks with far more entries than d1 content has, or raw_quant == 0, take
 * a dense dequantisation pass -- same bits, see JXLH_FRAME_DENSE_DEQUANT).  The uploads land in a second set of buffers:
 * the call may be issued for the NEXT frame while the previous jxlh_frame_run is still executing and waits for nothing
 * of it (jxlh_ctx_mark / jxlh_ctx_wait_mark is the host loop for that).  Groups of the epoch that are NOT self-contained
 * slot-bucketed submissions -- another form (dense slab, plain pairs), a value in `wide`, JXLH_GROUP_ACCUMULATE -- are
 * brought into their dense slabs on the device and read from there, group by group, while the rest of the frame is still
 * read in place (round 6; as long as every group arrived in the epoch and at least half of them in place -- otherwise
 * the whole frame takes the general route: sort, or zero-fill + scatter into dense slabs).  Values beyond 10 bits
 * need not go to `wide`: see "WIDE VALUES" below.
 *   JXLH_ERR_INVALID_ARGUMENT is returned before anything is reserved (a rejected call leaves the epoch as it was). */
jxlh_status jxlh_submit_groups_slots(jxlh_ctx* ctx, int32_t slot, uint32_t count, const uint32_t* group_ids,
                                     const uint16_t* entries, const uint8_t* slot_counts, const uint32_t* n,
                                     const jxlh_coeff32* wide, uint32_t n_wide, uint32_t flags);
/* ---- host side of the slot-bucketed form (round 6): plain CPU code, no context, no device, any thread ----
 * WIDE VALUES.  An entry holds 10 bits (6 with JXLH_GROUP_ENTRIES12); the reference accumulates arbitrary i32
 * (`coeff = read_signed_inline(..) << shift; current_coeffs[idx] += coeff`, frame/group.rs:568-572).  A producer SPLITS a
 * value outside the range into repeated in-range entries at the same position -- 2000 = 511 + 511 + 511 + 467 -- which
 * add up as integers on the device before the dequantisation, exactly like several passes' updates of one coefficient:
 * the frame stays in the form the transforms read in place (varblocks with more entries than their lanes hold take a
 * dense dequantisation pass, same bits).  Only what no slot can hold (a slot's count is a u8: 255 entries) goes to
 * `wide`, and a group with a `wide` value -- like a group that arrives in another form, or adds a pass to earlier
 * content (JXLH_GROUP_ACCUMULATE) -- is read from its dense slab while every other group of the frame is still read in
 * place (per-group routing; round 5 took the whole frame out of the in-place form).  Both helpers below do the split.
 *
 * jxlh_host_pack_slots: one group's dense slab (3 x 65536 i32: VarDctBuffers::coeffs_storage, frame/group.rs:27-67,
 * :437-440) -> the arguments of jxlh_submit_groups_slots for that group.  entries: room for entries_capacity entries
 * (u16 each; with JXLH_GROUP_ENTRIES12 the three runs are written as bytes, 1.5 per entry, each closed to an even
 * count); slot_counts: 3 x 1024; n[3]: entries per channel; wide / n_wide: values that could not be split (positions
 * already carry group_id as the batched call wants them).  JXLH_ERR_INVALID_ARGUMENT when a capacity is too small. */
jxlh_status jxlh_host_pack_slots(const int32_t* coeffs, uint32_t group_id, uint32_t flags, void* entries,
                                 size_t entries_capacity, uint8_t* slot_counts, uint32_t n[3], jxlh_coeff32* wide,
                                 uint32_t wide_capacity, uint32_t* n_wide);
/* A batch of groups in one call: exactly the arrays of ONE jxlh_submit_groups_slots call for them -- group_coeffs[g] is
 * group group_ids[g]'s dense slab (the reference keeps one Vec per group: Frame::hf_coefficients, frame/mod.rs), entries
 * / slot_counts (n_groups x 3 x 1024) / n (n_groups x 3) are written group after group, *entries_used = entries written
 * (nullable).  What a decoder thread calls for its share of a frame (bench.py: host_pack_ms_per_frame times this). */
jxlh_status jxlh_host_pack_slots_many(const int32_t* const* group_coeffs, const uint32_t* group_ids, uint32_t n_groups,
                                      uint32_t flags, void* entries, size_t entries_capacity, uint8_t* slot_counts,
                                      uint32_t* n, jxlh_coeff32* wide, uint32_t wide_capacity, uint32_t* n_wide,
                                      size_t* entries_used);
/* The same form written by the entropy loop itself (frame/group.rs:557-575), one writer per decoding thread:
 *   begin_group(buffers)                                 once per group (zeroes slot_counts)
 *   begin_varblock(first_slot, num_slots)                coeffs_offset / 64 and cx * cy of the varblock (group.rs:612);
 *                                                        varblocks in the order they are decoded (ascending offsets)
 *   add(channel, pos, value)                             `coeffs[channel][coeffs_offset + pos] += value`, pos inside the
 *                                                        varblock (order[k]); channels and positions in any order
 *   end_group(n, n_wide)                                 closes the last varblock, writes the three runs to `entries`
 * Entries of a one-slot varblock (8x8: the common case) are appended as they arrive; a larger varblock's are
 * bucketed by slot when it ends. */
typedef struct jxlh_slot_writer jxlh_slot_writer;
jxlh_status jxlh_slot_writer_create(jxlh_slot_writer** out);
void jxlh_slot_writer_destroy(jxlh_slot_writer* w);
jxlh_status jxlh_slot_writer_begin_group(jxlh_slot_writer* w, uint32_t group_id, uint32_t flags, void* entries,
                                         size_t entries_capacity, uint8_t* slot_counts, jxlh_coeff32* wide,
                                         uint32_t wide_capacity);
jxlh_status jxlh_slot_writer_begin_varblock(jxlh_slot_writer* w, uint32_t first_slot, uint32_t num_slots);
jxlh_status jxlh_slot_writer_add(jxlh_slot_writer* w, uint32_t channel, uint32_t pos, int32_t value);
jxlh_status jxlh_slot_writer_add_many(jxlh_slot_writer* w, uint32_t channel, const uint32_t* pos, const int32_t* value,
                                      size_t count);
jxlh_status jxlh_slot_writer_end_group(jxlh_slot_writer* w, uint32_t n[3], uint32_t* n_wide);

/* Blocks until the host buffers of the submissions made on `slot` so far may be reused (their host-to-device copies have
 * landed).  Device-side work a submission queues behind its copies on the slot's stream (the unpack pass of
 * JXLH_GROUP_ENTRIES12) is NOT waited for here: jxlh_frame_run orders itself behind it. */
jxlh_status jxlh_slot_wait(jxlh_ctx* ctx, int32_t slot);
/* Orders uploads ACROSS contexts on the device: whatever is submitted on (ctx, slot) after this call starts when the
 * uploads enqueued so far on (after_ctx, after_slot) have landed -- the device-side form of "jxlh_slot_wait(after_ctx,
 * after_slot), then submit", without blocking the host.  Two contexts that stream frames keep the bus busy back to back
 * and stay in anti-phase (one uploads while the other computes) whatever the host's latency is (round 5,
 * profiles/r05_f_e2e_host_loops.txt).  Both contexts must live on the same device. */
jxlh_status jxlh_slot_after(jxlh_ctx* ctx, int32_t slot, jxlh_ctx* after_ctx, int32_t after_slot);

/* Device-resident coefficient store of the current frame (ngroups * 3 * 65536 i32), for callers
 * that already hold coefficients in HBM (bench harness, multi-GPU shards). */
jxlh_status jxlh_frame_coeff_buffer(jxlh_ctx* ctx, int32_t** device_ptr, size_t* n_int32);

/* Runs everything that has not run yet for the frame on the context's main stream:
 * K0b adaptive LF smoothing (Frame::finalize_lf, frame/mod.rs:360-378), K3sigma
 * (SigmaSource::new, features/epf.rs:35-87), K1 for every submitted group, then the stage list
 * of frame/render.rs:569-622 (Gaborish x3, EPF0/1/2).  group range [g0, g1) restricts K1 and the
 * filters to a band of group rows (multi-GPU sharding); pass 0, UINT32_MAX for the whole frame. */
jxlh_status jxlh_frame_run(jxlh_ctx* ctx, uint32_t group_row0, uint32_t group_row1);
/* mark_group_to_rerender + the re-render it triggers (render/mod.rs:143-146, callers frame/decode.rs:703-711): after
 * a jxlh_frame_run, groups whose coefficients changed (a later progressive pass) are reconstructed again -- the
 * transforms of exactly the listed groups, then the filters on every pixel row those groups influence (their rows
 * widened by the stage list's border: the part of the reference's 3x3 group neighbourhood that can change).  Stage
 * lists that overwrite the transforms' output (epf_iters == 3, per-stage kernels with an even stage count) and
 * chroma-subsampled frames re-render the whole frame instead; upsampled frames return JXLH_ERR_UNSUPPORTED. */
jxlh_status jxlh_frame_rerender_groups(jxlh_ctx* ctx, const uint32_t* group_ids, uint32_t count);
/* blocks until the main stream is idle */
jxlh_status jxlh_ctx_sync(jxlh_ctx* ctx);
/* PLACEMENT OF A CONTEXT'S BUFFERS (round 6; profiles/r06_q_context_placement.txt).  Where the driver places a context's
 * large buffers decides how fast the transforms and filters run on them: the same kernels on the same data are up to 10 %
 * apart between two contexts of one process -- persistently, reproducibly for a box and an allocation order, and not
 * visible in plain copies between the buffers.  jxlh_ctx_tune_placement(ctx, trials, ...) makes the context's NEXT first
 * allocation of {three planes, three filter planes, coefficient buffer} (the first jxlh_frame_begin, or the first one
 * after the buffers were released) a pick among `trials` candidate sets (up to twice as many while none of them stands
 * out), rated on the device by two byte movers with the streams of the 8x8 transform class and of the filters; the
 * candidates are held until the pick, then all but the best are freed.  Costs setup time (a few ms per candidate) and trials x the buffers' size in transient device memory (2.6 GB per
 * candidate at 8192^2); a candidate that cannot be allocated ends the trials early.  trials = 0: query only; 1: plain
 * allocation (the default).  report (nullable): the last pick's ratings, two floats per candidate (ms of the two
 * movers), *n_report floats; *picked: the candidate taken (-1: no pick yet).  No effect on results. */
jxlh_status jxlh_ctx_tune_placement(jxlh_ctx* ctx, int32_t trials, float* report, int32_t report_capacity, int32_t* n_report,
                                    int32_t* picked);

/* STREAM ORDERING OF DEVICE POINTERS.  Every stream of a context is a hipStreamNonBlocking stream: it does NOT
 * synchronise with the NULL (legacy default) stream or with any stream of the caller.  Host pointers are safe by
 * construction (the library's own copies are ordered on its streams), but a caller that fills a DEVICE buffer -- a
 * hipMemset / hipMemcpy on the NULL stream, a kernel on its own stream -- and then hands the pointer to an entry
 * point (jxlh_unsqueeze_chain, jxlh_rct, jxlh_palette*, jxlh_submit_group* with device sources, the stage hooks, the
 * *_device destinations of the read calls) must order that work in FRONT of the call, or the library's kernels may
 * run before the fill has (hipMemset returns before the fill has run; the same holds for the outputs: an output plane
 * the caller clears on the NULL stream can be cleared AFTER the library wrote it).  The reference hands buffers over by
 * ownership (RenderPipeline::set_buffer_for_group takes `buf: Image<T>`, render/mod.rs:124-137), so a binding has to
 * make the hand-over explicit with one of:
 *   jxlh_ctx_wait_stream(ctx, stream)   everything enqueued so far on `stream` (a hipStream_t; NULL = the legacy default
 *                                       stream) happens before whatever the context enqueues from now on -- on the
 *                                       device, the host does not block
 *   jxlh_ctx_wait_event(ctx, event)     the same for a hipEvent_t the caller has recorded
 *   jxlh_ctx_record_event(ctx, event)   the other direction: records the caller's event behind everything enqueued so
 *                                       far on the context's main stream, so a caller stream can hipStreamWaitEvent on
 *                                       the library's results without a host-side jxlh_ctx_sync
 * (a host-side hipDeviceSynchronize / hipStreamSynchronize before the call is the blunt alternative).  hipStream_t /
 * hipEvent_t travel as void* so that this header needs no HIP header. */
jxlh_status jxlh_ctx_wait_stream(jxlh_ctx* ctx, void* hip_stream);
jxlh_status jxlh_ctx_wait_event(jxlh_ctx* ctx, void* hip_event);
jxlh_status jxlh_ctx_record_event(jxlh_ctx* ctx, void* hip_event);
/* A point in the context's main stream: everything enqueued so far (frame runs, asynchronous reads).
 * jxlh_ctx_wait_mark blocks until that point has been reached and -- unlike jxlh_ctx_sync -- not for work enqueued after
 * the mark.  That is what lets ONE context stream consecutive frames (round 5: the slot-bucketed submission of frame
 * i + 1 no longer waits for frame i's transforms): submit frame i + 1 (its upload runs under frame i's kernels),
 * jxlh_frame_run, jxlh_frame_read_*_async, mark; then wait for the mark of frame i, whose output is complete.  Up to
 * JXLH_MAX_MARKS marks are alive at a time; waiting for an older one waits for the mark that replaced it.  Device-side
 * errors (JXLH_ERR_INVALID_TRANSFORM ...) are reported by jxlh_ctx_sync only. */
#define JXLH_MAX_MARKS 8
jxlh_status jxlh_ctx_mark(jxlh_ctx* ctx, uint32_t* mark);
jxlh_status jxlh_ctx_wait_mark(jxlh_ctx* ctx, uint32_t mark);
/* Copies the finished planes out (host or device destination).  Replaces the save stage for
 * f32 XYB output; xsize x ysize samples per plane. */
jxlh_status jxlh_frame_read_planes(jxlh_ctx* ctx, const jxlh_plane out[3]);
/* The same for one rectangle of the result: pixels [x0, x0 + w) x [y0, y0 + h), cut at the result's right and bottom
 * edge, into out[c] (row 0 of out[c] = row y0; at least min(w, xsize - x0) samples per row and min(h, ysize - y0)
 * rows; what lies beyond the cut is left untouched).  This is the unit the reference's pipeline moves:
 * RenderPipeline::get_buffer / set_buffer_for_group (render/mod.rs:124-137) hand over one 256 x 256 group per
 * channel, in buffers rounded up to 16 pixels (group_size_for_channel, render/internal.rs:144-167) -- a caller that
 * feeds the finished XYB planes to the remaining CPU stages group by group reads group g with
 * x0 = (g % xgroups) * 256, y0 = (g / xgroups) * 256, w = h = 256.  JXLH_ERR_INVALID_ARGUMENT
 * if the rect starts outside the result.  _async: without the final wait (host destinations should be pinned);
 * `out` is valid after the next jxlh_ctx_sync. */
jxlh_status jxlh_frame_read_planes_rect(jxlh_ctx* ctx, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h,
                                        const jxlh_plane out[3]);
jxlh_status jxlh_frame_read_planes_rect_async(jxlh_ctx* ctx, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h,
                                              const jxlh_plane out[3]);
/* device-resident result (row stride in floats), valid until the next frame_begin */
jxlh_status jxlh_frame_device_planes(jxlh_ctx* ctx, float* planes[3], size_t* stride);
/* Extra channels of the frame (alpha, depth, ...: channels 3 + ec of the reference's pipeline, frame/render.rs:564-567,
 * :624-637, :655-671).  jxlh_frame_set_extra_channel hands over Modular channel `ec` as decoded -- w x h i32 samples at
 * row stride `stride` (host or device memory) --; the next jxlh_frame_run (whole frame) applies
 * ConvertModularToF32Stage with the channel's bit depth (render/stages/convert.rs:488-533) and, for
 * ec_upsampling = 2 / 4 / 8, Upsample2x / 4x / 8x with the weights of jxlh_set_upsampling_weights
 * (render/stages/upsample.rs) -- before the colour channels' own upsampling or together with it, the result is the
 * same per channel.  ec_upsampling counts from the channel's own resolution: w = ceil(full width / ec_upsampling).
 * jxlh_frame_read_extra_channel copies the finished channel out: out_w x out_h f32 samples with
 * out_w = min(w * ec_upsampling, width of the frame's result), likewise the height.  Up to JXLH_MAX_EXTRA_CHANNELS. */
/* bits_per_sample carries the channel's BitDepth (headers/bit_depth.rs): bits_per_sample | exponent_bits_per_sample
 * << 8.  Exponent bits 0 = integer samples (val / (2^bits - 1)); otherwise the samples are `bits`-bit floats stored in
 * integers (binary16 = 16 | 5 << 8, binary32 = 32 | 8 << 8, any custom format) and are widened to binary32 exactly
 * (int_to_float, convert.rs:416-486).  The same convention holds for jxlh_modular_to_f32.  A channel handed over after
 * the frame's first render is converted by the next jxlh_frame_run or jxlh_frame_rerender_groups. */
#define JXLH_MAX_EXTRA_CHANNELS 8
jxlh_status jxlh_frame_set_extra_channel(jxlh_ctx* ctx, uint32_t ec, const int32_t* samples, size_t stride, uint32_t w,
                                         uint32_t h, uint32_t bits_per_sample, uint32_t ec_upsampling);
jxlh_status jxlh_frame_read_extra_channel(jxlh_ctx* ctx, uint32_t ec, const jxlh_plane* out);
/* smoothed LF image as used by K1 (tests) */
jxlh_status jxlh_frame_read_lf(jxlh_ctx* ctx, float* x, float* y, float* b, size_t stride);

/* ---------------------------------------------------------------- 8-bit sRGB output
 * The stages the reference runs after EPF for an XYB-encoded frame saved as 8-bit sRGB -- XybStage
 * (render/stages/xyb.rs:208-240), FromLinearStage with the sRGB curve (color/tf.rs:13-44),
 * ConvertF32ToU8Stage incl. its dither (render/stages/convert.rs:570-606), chained as in
 * frame/render.rs:757-762, :118 -- in one pass over the finished planes, written interleaved
 * (channels = 3: RGB, 4: RGBA with A = 255) for frame rows [y0, y1).  `out` points at row y0 and may be
 * host or device memory; for host memory the call returns after the copy has completed.
 * The parameters are the reference's XybParams::new(opsin, intensity_target) (xyb.rs:147-163):
 * inverse matrix, cbrt(biases), biases * intensity_scale, intensity_scale = 255 / intensity_target.
 * Frames whose output colour space is not sRGB/D65 with the sRGB transfer function, or that need
 * upsampling / blending / extra channels, keep the reference's CPU stages (JXLH_ERR_UNSUPPORTED is the
 * caller's decision: this entry point does what it says). */
typedef struct jxlh_xyb_params {
  float opsin_inverse_matrix[9];
  float bias_cbrt[3];
  float scaled_bias[3];
  float intensity_scale;
} jxlh_xyb_params;
jxlh_status jxlh_frame_read_rgb8(jxlh_ctx* ctx, const jxlh_xyb_params* p, uint32_t channels, uint32_t y0,
                                 uint32_t y1, void* out, size_t bytes_per_row);
/* jxlh_frame_read_rgb8 without the final wait: the conversion and the copy to `out` are queued on the context's
 * stream and the call returns; `out` (pinned host memory, or device memory) holds the image after the next
 * jxlh_ctx_sync.  Lets the download of frame i run while the caller submits frame i + 1 to another context. */
jxlh_status jxlh_frame_read_rgb8_async(jxlh_ctx* ctx, const jxlh_xyb_params* p, uint32_t channels, uint32_t y0,
                                       uint32_t y1, void* out, size_t bytes_per_row);
/* The same with ConvertF32ToU16Stage at 16 bits (render/stages/convert.rs:743-761: clamp to [0,1], x65535,
 * round to nearest even, no dither): native-endian u16 samples, interleaved. */
jxlh_status jxlh_frame_read_rgb16(jxlh_ctx* ctx, const jxlh_xyb_params* p, uint32_t channels, uint32_t y0,
                                  uint32_t y1, void* out, size_t bytes_per_row);

/* The same for a YCbCr frame (do_ycbcr, not XYB-encoded -- JPEG recompressions): YcbcrToRgbStage
 * (render/stages/ycbcr.rs:35-78) on the planes taken as Cb, Y, Cr, then the integer conversion; no
 * transfer-function stage (frame/render.rs:755-763). */
jxlh_status jxlh_frame_read_ycbcr_rgb8(jxlh_ctx* ctx, uint32_t channels, uint32_t y0, uint32_t y1, void* out,
                                       size_t bytes_per_row);
jxlh_status jxlh_frame_read_ycbcr_rgb16(jxlh_ctx* ctx, uint32_t channels, uint32_t y0, uint32_t y1, void* out,
                                        size_t bytes_per_row);

/* General form of the output calls: the colour stage of the frame (frame/render.rs:755-763) -- XybStage, for an
 * XYB-encoded frame, followed by FromLinearStage with one of the reference's transfer functions
 * (render/stages/from_linear.rs:133-145; curves of color/tf.rs and util/fast_math.rs, evaluated operation for
 * operation), YcbcrToRgbStage, or nothing -- then ConvertF32ToU8Stage (bits = 8) or ConvertF32ToU16Stage (16).
 *   JXLH_TF_LINEAR  no FromLinearStage (linear output, TransferFunction::is_linear)
 *   JXLH_TF_SRGB    linear_to_srgb_simd            JXLH_TF_BT709  linear_to_bt709_simd
 *   JXLH_TF_PQ      linear_to_pq_simd, tf_param = intensity_target
 *   JXLH_TF_HLG     hlg_display_to_scene + scene_to_hlg; tf_param = (1 - g) / g with the system gamma
 *                   g = 1.2 * 1.111^log2(intensity_target / 1000) evaluated by the caller exactly as
 *                   color/tf.rs:442-446 does (host libm), hlg_luminance_rgb = luminance_rgb
 *   JXLH_TF_GAMMA   fast_powf_simd(|v|, tf_param), tf_param = the encoding exponent in (0, 1] */
enum { JXLH_COLOR_XYB = 0, JXLH_COLOR_YCBCR = 1, JXLH_COLOR_NONE = 2 };
enum { JXLH_TF_LINEAR = 0, JXLH_TF_SRGB = 1, JXLH_TF_BT709 = 2, JXLH_TF_PQ = 3, JXLH_TF_HLG = 4, JXLH_TF_GAMMA = 5 };
typedef struct jxlh_output_desc {
  uint32_t color;     /* JXLH_COLOR_* */
  uint32_t transfer;  /* JXLH_TF_*, used with JXLH_COLOR_XYB */
  jxlh_xyb_params xyb;
  float tf_param;
  float hlg_luminance_rgb[3];
  uint32_t bits;      /* 8 or 16 */
  uint32_t channels;  /* 3 or 4 (alpha = opaque) */
} jxlh_output_desc;
jxlh_status jxlh_frame_read_output(jxlh_ctx* ctx, const jxlh_output_desc* d, uint32_t y0, uint32_t y1, void* out,
                                   size_t bytes_per_row);
/* without the final wait, like jxlh_frame_read_rgb8_async: `out` is valid after the next jxlh_ctx_sync */
jxlh_status jxlh_frame_read_output_async(jxlh_ctx* ctx, const jxlh_output_desc* d, uint32_t y0, uint32_t y1, void* out,
                                         size_t bytes_per_row);

/* ---------------------------------------------------------------- stage-level hooks */
/* Whole-image single stages with the pipeline's mirror edge semantics; the analogue of
 * make_and_run_simple_pipeline (jxl/src/render/test.rs:83-179).  Planes: w x h f32, row stride
 * `stride` floats, host or device pointers. */
jxlh_status jxlh_stage_gaborish(jxlh_ctx* ctx, const float* in, float* out, uint32_t w, uint32_t h,
                                size_t stride, float w1, float w2);
/* stage = 0, 1, 2 (Epf0Stage / Epf1Stage / Epf2Stage); inv_sigma: ceil(w/8) x ceil(h/8) floats,
 * row stride sigma_stride; uses the epf_* fields of p */
jxlh_status jxlh_stage_epf(jxlh_ctx* ctx, int32_t stage, const jxlh_frame_params* p,
                           const float* const in[3], float* const out[3], uint32_t w, uint32_t h,
                           size_t stride, const float* inv_sigma, size_t sigma_stride);
/* HorizontalChromaUpsample (horizontal != 0: out is 2w x h) or VerticalChromaUpsample (out is w x 2h) on a
 * tight w x h plane, edges mirrored (render/stages/chroma_upsample.rs:31-63, :108-147) */
jxlh_status jxlh_stage_chroma_upsample(jxlh_ctx* ctx, const float* in, float* out, uint32_t w, uint32_t h,
                                       int32_t horizontal);
/* Upsample2x / 4x / 8x (render/stages/upsample.rs) on a tight w x h plane -> (n*w) x (n*h), n = 2, 4, 8, with the
 * weights of jxlh_set_upsampling_weights; input mirrored 2 pixels at its edges */
jxlh_status jxlh_stage_upsample(jxlh_ctx* ctx, int32_t n, const float* in, float* out, uint32_t w, uint32_t h);
/* Noise synthesis, stage by stage (tight arrays): the three random planes of a w x h image
 * (render_noise_for_group, frame/decode.rs:578-668); ConvolveNoiseStage (render/stages/noise.rs:32-86);
 * AddNoiseStage on n samples with p's noise_lut and colour-correlation fields (noise.rs:140-189), in place. */
jxlh_status jxlh_stage_noise_generate(jxlh_ctx* ctx, uint32_t visible_frame_index, uint32_t nonvisible_frame_index,
                                      uint32_t w, uint32_t h, float* const out[3]);
jxlh_status jxlh_stage_noise_convolve(jxlh_ctx* ctx, const float* in, float* out, uint32_t w, uint32_t h);
jxlh_status jxlh_stage_noise_add(jxlh_ctx* ctx, const jxlh_frame_params* p, float* const planes[3],
                                 const float* const rnd[3], size_t n);
/* adaptive_lf_smoothing on w x h tight planes (frame/adaptive_lf_smoothing.rs:44-125) */
jxlh_status jxlh_stage_lf_smooth(jxlh_ctx* ctx, const jxlh_frame_params* p, const float* const in[3],
                                 float* const out[3], uint32_t w, uint32_t h);
/* transform_to_pixels on a batch of independent varblocks of one type (transform.rs:666-677):
 * coeffs n * cx*cy*64 dequantised coefficients, lf n * cx*cy samples, pixels n * (cy*8)*(cx*8). */
jxlh_status jxlh_stage_transform_to_pixels(jxlh_ctx* ctx, int32_t type, uint32_t n, const float* coeffs,
                                           const float* lf, float* pixels);

/* ---------------------------------------------------------------- Modular (wrapping i32) */
/* do_rct_step (modular/transforms/rct.rs:118-157) on whole planes of n samples, in place. */
jxlh_status jxlh_rct(jxlh_ctx* ctx, int32_t* p0, int32_t* p1, int32_t* p2, size_t n, int32_t op,
                     int32_t perm);
/* do_palette_step_general, num_deltas == 0 && predictor == Zero branch (palette.rs:182-199).
 * palette: nb_channels rows x palette_stride; out: nb_channels planes of n samples, contiguous. */
jxlh_status jxlh_palette(jxlh_ctx* ctx, const int32_t* index, size_t n, const int32_t* palette,
                         int32_t num_colors, size_t palette_stride, int32_t nb_channels,
                         int32_t bit_depth, int32_t* out);
/* The same on a run of n samples of a larger image (device pointers only): channel c of the run is written at
 * out + c * out_channel_stride, so a rank can expand its share of the index plane straight into full-size planes. */
jxlh_status jxlh_palette_strided(jxlh_ctx* ctx, const int32_t* index, size_t n, const int32_t* palette,
                                 int32_t num_colors, size_t palette_stride, int32_t nb_channels, int32_t bit_depth,
                                 int32_t* out, size_t out_channel_stride);
/* do_palette_step_general with delta entries and / or a neighbour predictor (palette.rs:228-251): index is w x h,
 * entries below num_deltas are added to Predictor::predict_one (modular/predict.rs:152-198; predictor = Predictor
 * as u32; 6 = Weighted needs its header: jxlh_palette_delta_wp, JXLH_ERR_UNSUPPORTED here) of the already
 * reconstructed neighbours, palette_size =
 * num_colors + num_deltas.  Sequential by nature: runs as a skewed wavefront, far from bandwidth bound. */
jxlh_status jxlh_palette_delta(jxlh_ctx* ctx, const int32_t* index, uint32_t w, uint32_t h, const int32_t* palette,
                               int32_t num_colors, int32_t num_deltas, size_t palette_stride, int32_t nb_channels,
                               int32_t bit_depth, int32_t predictor, int32_t* out);
/* The same step with Predictor::Weighted (palette.rs:200-227): every pixel runs the self-correcting predictor
 * (WeightedPredictorState, modular/predict.rs:221-519) and updates its error state, delta entries are added to its
 * prediction.  wp = the group's WeightedHeader (headers/modular.rs:16-66; 5-bit p*, 4-bit w*). */
typedef struct jxlh_wp_header {
  uint32_t p1c, p2c, p3ca, p3cb, p3cc, p3cd, p3ce, w0, w1, w2, w3;
} jxlh_wp_header;
jxlh_status jxlh_palette_delta_wp(jxlh_ctx* ctx, const int32_t* index, uint32_t w, uint32_t h, const int32_t* palette,
                                  int32_t num_colors, int32_t num_deltas, size_t palette_stride, int32_t nb_channels,
                                  int32_t bit_depth, const jxlh_wp_header* wp, int32_t* out);
/* The stages between the Modular channels and the rest of the pipeline (render/stages/convert.rs), whole planes:
 *  - jxlh_modular_to_rgb8: ConvertI32ToU8Stage (:642-715) on three channels, interleaved -- what the pipeline builder
 *    substitutes for ConvertModularToF32 + ConvertF32ToU8 when the output depth is a multiple of the channel depth
 *    (render/builder.rs:152-170): multiplier = (2^8 - 1) / (2^bits - 1), max = 255; a lossless 8-bit image after
 *    RCT / Palette / Squeeze goes straight to displayable bytes
 *  - jxlh_modular_to_f32: ConvertModularToF32Stage for integer samples (:488-533), val * 1 / (2^bits - 1)
 *  - jxlh_modular_xyb_to_f32: ConvertModularXYBToF32Stage (:306-343), channels in coded order Y, X, B,
 *    quant_factors = LfQuantFactors::quant_factors (X, Y, B) */
jxlh_status jxlh_modular_to_rgb8(jxlh_ctx* ctx, const int32_t* const planes[3], size_t stride, uint32_t w, uint32_t h,
                                 int32_t multiplier, int32_t max, uint32_t channels, void* out, size_t bytes_per_row);
jxlh_status jxlh_modular_to_f32(jxlh_ctx* ctx, const int32_t* in, size_t n, uint32_t bits_per_sample, float* out);
jxlh_status jxlh_modular_xyb_to_f32(jxlh_ctx* ctx, const int32_t* y, const int32_t* x, const int32_t* b, size_t n,
                                    const float quant_factors[3], float* ox, float* oy, float* ob);
/* SAMPLE RANGE of every squeeze entry point below (jxlh_unsqueeze*, jxlh_unsqueeze_chain): averages, residuals and
 * reconstructed samples in [-2^28, 2^28).  Inside it the result equals BOTH forms the reference holds -- the i64
 * `unsqueeze_scalar` (squeeze.rs:187-194) and the wrapping-i32 `unsqueeze_impl` / `smooth_tendency_impl` of its SIMD
 * back-ends (squeeze.rs:107-185) -- bit for bit.  From about 2^29 on those two forms give DIFFERENT results for the
 * same input (which one a sample gets depends on its place in the back-end's tiling), so the reference defines no
 * value there; the device then computes a third wrapping-i32 form, deterministic but equal to neither.  8..24-bit
 * images, also after an RCT, are far inside the range (tests/test_oracle_pin.py pins the bound on all three forms). */
/* do_hsqueeze_step / do_vsqueeze_step (squeeze.rs:456-481, :661-682), whole plane.
 * horizontal: avg is ceil(out_w/2) x h, res floor(out_w/2) x h; vertical likewise in y.  `out` may not overlap `avg` or
 * `res` (lines are streamed: inputs are read ahead of the outputs being written). */
jxlh_status jxlh_unsqueeze(jxlh_ctx* ctx, int32_t horizontal, const int32_t* avg, size_t avg_stride,
                           const int32_t* res, size_t res_stride, uint32_t out_w, uint32_t out_h,
                           int32_t* out, size_t out_stride);

/* Batched form: the n_planes (1..3) channels one squeeze step covers (SqueezeParams::num_channels,
 * modular/transforms/squeeze.rs:20-37) in a single launch -- the recurrence is latency bound, so
 * extra planes are almost free.  All planes share geometry; device pointers only. */
jxlh_status jxlh_unsqueeze_planes(jxlh_ctx* ctx, int32_t horizontal, int32_t n_planes,
                                  const int32_t* const avg[], size_t avg_stride, const int32_t* const res[],
                                  size_t res_stride, uint32_t out_w, uint32_t out_h, int32_t* const out[],
                                  size_t out_stride);

/* Several consecutive squeeze steps in one call: level i takes the previous level's output (the base planes for
 * level 0) as its averages and levels[i].res as its residuals; only the last level's planes are written.  The first
 * levels of a chain are tiny (default_squeeze, squeeze.rs:71-105, starts from <= 8 x 8): as separate launches they cost
 * ~10 us each whatever their size, so up to 16 levels whose planes stay within 128 x 128 run as ONE launch with the
 * planes in LDS; anything else takes jxlh_unsqueeze_chain's route without an RCT (same result).  Device pointers only. */
typedef struct jxlh_squeeze_level {
  int32_t horizontal;
  uint32_t out_w, out_h;
  const int32_t* res[3]; /* floor(out_w/2) x out_h (horizontal) or out_w x floor(out_h/2); unused planes NULL */
  size_t res_stride;
} jxlh_squeeze_level;
jxlh_status jxlh_unsqueeze_levels(jxlh_ctx* ctx, int32_t n_planes, int32_t n_levels, const jxlh_squeeze_level* levels,
                                  const int32_t* const base[], size_t base_stride, uint32_t base_w, uint32_t base_h,
                                  int32_t* const out[], size_t out_stride);

/* The inverse of a WHOLE squeeze transform in one call: jxlh_unsqueeze_levels' level list (smallest level first, the
 * order the decoder applies the steps of default_squeeze / an explicit SqueezeParams list in, squeeze.rs:39-105,
 * transforms/apply.rs), optionally followed by the RCT that comes next in the transform list (rct_op 0..6 and
 * rct_perm 0..5 as in jxlh_rct, n_planes == 3; rct_op < 0: none) -- the last, full-resolution step and the RCT are
 * then one pass over the planes.  The library picks the launches (LDS-resident first levels; the streamed middle
 * levels as ONE dataflow launch in which a level starts on the rows / columns the level before it has finished;
 * fused last level); intermediate planes live in context scratch.  Device pointers only; `out` may not alias
 * the base or residual planes.  Asynchronous like every call on the context's stream; should a wait between two
 * levels of the dataflow launch ever outlast its deadline (4 s: a fault, not a load condition), the launch ends with
 * an undefined result and the next jxlh_ctx_sync returns JXLH_ERR_DEVICE.  JXLH_CHAIN_FLOW=0 in the environment
 * selects one launch per streamed level (same result; tests, A/B).
 * Layout advice (speed only, any layout is accepted): residual and output planes that start on 16-byte boundaries with
 * strides that are multiples of 4 samples are moved with 16-byte accesses (the library lays its own intermediate
 * planes out that way); other layouts take the 4-byte movers, about 1.3x the time of a streamed level. */
jxlh_status jxlh_unsqueeze_chain(jxlh_ctx* ctx, int32_t n_planes, int32_t n_levels, const jxlh_squeeze_level* levels,
                                 const int32_t* const base[], size_t base_stride, uint32_t base_w, uint32_t base_h,
                                 int32_t* const out[], size_t out_stride, int32_t rct_op, int32_t rct_perm);

/* A squeeze step of three channels followed by do_rct_step (rct.rs:118-157) on the same three channels, in one pass:
 * the shape the end of a colour image's inverse transform chain has (default_squeeze, squeeze.rs:71-105, finishes with
 * the full-size step -- vertical for square and tall images, horizontal for wide ones; the encoder applied the RCT
 * before the squeeze, so it is undone after it).  Equivalent to jxlh_unsqueeze_planes(horizontal, 3 planes) +
 * jxlh_rct(out[0..2], op, perm) without the second read and write of the image.  out planes may not alias avg / res.
 * Device pointers only. */
jxlh_status jxlh_unsqueeze_rct(jxlh_ctx* ctx, int32_t horizontal, const int32_t* const avg[3], size_t avg_stride,
                               const int32_t* const res[3], size_t res_stride, uint32_t out_w, uint32_t out_h,
                               int32_t* const out[3], size_t out_stride, int32_t op, int32_t perm);

/* smooth_h_unsqueeze / smooth_v_unsqueeze / smooth_2d_unsqueeze (modular/transforms/squeeze.rs:1010-1105, :1120-1225,
 * :908-1003): the step a squeeze runs while its residual channel has not arrived (DataStatus::Zero,
 * transforms/step.rs:138-150, dispatched at :841-851) -- the progressive previews of a squeezed image.  `avg` is the
 * WHOLE average channel (avg_w x avg_h; for JXLH_SMOOTH_2D the average of two steps back, half size in both axes);
 * the out_w x out_h output rectangle sits at (x0, y0) of the output channel -- (0, 0) and the full size for a whole
 * channel, a grid tile's Rect otherwise (the window reads across tile edges, columns clamp and rows mirror at the
 * CHANNEL's borders: TiledChannelView::load_row_to_scratch, step.rs:372-420).  As in the reference, a rectangle
 * without one complete sample pair on a doubled axis is left untouched.
 * Float-to-int conversion: the reference's `as_i32` differs between its SIMD back-ends -- truncation on the scalar,
 * NEON and wasm ones (jxl_simd/src/scalar.rs:178, aarch64/neon.rs:400, wasm32/simd128.rs:371), round-to-nearest-even
 * (cvtps2dq) on the x86 ones (x86_64/avx.rs:580, sse42.rs:472, avx512.rs:638), both applied after the +-0.5
 * (squeeze.rs:807-810, :882-883); the two differ by one on about half the samples.  The caller says which reference
 * build it stands in for: plain kind = truncation (an aarch64 / scalar reference), kind | JXLH_SMOOTH_CVT_NEAREST_EVEN
 * = an x86 reference.  Host or device pointers. */
enum { JXLH_SMOOTH_H = 0, JXLH_SMOOTH_V = 1, JXLH_SMOOTH_2D = 2, JXLH_SMOOTH_CVT_NEAREST_EVEN = 0x100 };
jxlh_status jxlh_smooth_unsqueeze(jxlh_ctx* ctx, int32_t kind, const int32_t* avg, size_t avg_stride, uint32_t avg_w,
                                  uint32_t avg_h, uint32_t x0, uint32_t y0, int32_t* out, size_t out_stride,
                                  uint32_t out_w, uint32_t out_h);

/* ---------------------------------------------------------------- multi-GPU
 * The reference renders a frame's groups on a pool of host threads and joins them in the pipeline's output buffers
 * (frame/render.rs:395-479).  Here a frame is cut into contiguous bands of group rows, one per GPU ("rank"):
 * rank r owns group rows [r * per, min((r + 1) * per, ygroups)), per = ceil(ygroups / nranks).  Each rank gets the
 * replicated small inputs (LF, HfMetadata maps, dequant tables) and the coefficient groups of ITS band, runs the
 * transforms on exactly that band, exchanges the one block row (8 pixel rows x 3 channels, 0.8 MB at 8K) its filters
 * read across each band edge with the neighbour rank, filters its band, and the finished bands are all-gathered so
 * that every rank holds the whole frame.  Frames with upsampling run whole (JXLH_ERR_UNSUPPORTED here).
 * Chroma-subsampled frames recompute the halo group row instead of exchanging it (their chroma upsampling reads
 * across the band edge in the sub-sampled domain): for such a frame a rank must ALSO be given the coefficient
 * groups of the one group row above and the one below its band, [r * per - 1, (r + 1) * per + 1) clipped to the
 * frame; and its chroma is brought to full resolution before the gather even when no filter stage follows.
 *
 * One process per GPU (torch.distributed.run, mpirun, ...): the library owns an RCCL communicator.
 *   rank 0: jxlh_comm_unique_id(id); the launcher broadcasts the 128 bytes; every rank: jxlh_comm_init(ctx, id, rank, n)
 *   BEFORE its first jxlh_frame_begin (which sizes the planes for the in-place gather).  Per frame, every rank calls
 *   jxlh_frame_run_sharded then jxlh_frame_allgather (collectives: all ranks, same order); both only enqueue work on
 *   the context's stream (ncclSend/ncclRecv of the edge rows, ncclAllGather per plane).
 * One process driving several GPUs (or several contexts on one GPU): jxlh_comm_init_local(peers, n) makes peers[i]
 *   rank i, jxlh_frames_run_sharded_local / jxlh_frames_allgather_local drive all of them with direct device copies.
 *
 * A collective whose peer never arrives (a rank that died, calls in a different order) would park the stream for
 * ever: jxlh_ctx_sync of a context with an RCCL communicator of more than one rank therefore waits with a deadline
 * (JXLH_COMM_TIMEOUT_S, default 120; <= 0: none) and polls ncclCommGetAsyncError; on expiry it returns JXLH_ERR_DEVICE
 * and jxlh_last_error names the rank, the world size and the last collective enqueued (e.g. "halo exchange ... with
 * rank 3"). */
#define JXLH_COMM_ID_BYTES 128
jxlh_status jxlh_comm_unique_id(uint8_t id[JXLH_COMM_ID_BYTES]);
jxlh_status jxlh_comm_init(jxlh_ctx* ctx, const uint8_t id[JXLH_COMM_ID_BYTES], int32_t rank, int32_t nranks);
jxlh_status jxlh_comm_init_local(jxlh_ctx* const peers[], int32_t nranks);
jxlh_status jxlh_comm_destroy(jxlh_ctx* ctx);
/* rank / nranks of the context (0 / 1 without a communicator) and, inside a frame, the band of group rows it owns;
 * any output pointer may be NULL */
jxlh_status jxlh_comm_band(jxlh_ctx* ctx, int32_t* rank, int32_t* nranks, uint32_t* group_row0, uint32_t* group_row1);
jxlh_status jxlh_frame_run_sharded(jxlh_ctx* ctx);
jxlh_status jxlh_frame_allgather(jxlh_ctx* ctx);
jxlh_status jxlh_frames_run_sharded_local(jxlh_ctx* const peers[], int32_t nranks);
jxlh_status jxlh_frames_allgather_local(jxlh_ctx* const peers[], int32_t nranks);
/* The gather of the CONVERTED image instead of the f32 planes (round 6): every rank runs the frame's colour stage +
 * integer conversion (jxlh_frame_read_output's stages: XybStage / FromLinearStage / ConvertF32ToU8 / U16) on the pixel
 * rows of ITS band, writes them interleaved into `out` -- a DEVICE buffer of at least nranks * per * 256 *
 * bytes_per_row bytes (per = the band height in group rows, jxlh_comm_band) that holds the whole image, row y at
 * y * bytes_per_row -- and the bands are all-gathered in place: 201 MB for an 8K RGB8 image instead of the 805 MB of
 * jxlh_frame_allgather.  Replaces jxlh_frame_allgather (call one of the two after jxlh_frame_run_sharded; same rules:
 * all ranks, same order, work only enqueued on the context's stream).  Frames without upsampling / noise, like every
 * sharded run. */
jxlh_status jxlh_frame_allgather_output(jxlh_ctx* ctx, const jxlh_output_desc* d, void* out, size_t bytes_per_row);
/* the same for an in-process group: outs[i] = rank i's image buffer */
jxlh_status jxlh_frames_allgather_output_local(jxlh_ctx* const peers[], int32_t nranks, const jxlh_output_desc* d,
                                               void* const outs[], size_t bytes_per_row);
/* In-place all-gather of a device buffer of nranks * bytes_per_rank bytes (rank r's part at r * bytes_per_rank), on the
 * context's stream: the join of band-sharded Modular work (RCT / Palette on row bands of whole planes). */
jxlh_status jxlh_comm_allgather(jxlh_ctx* ctx, void* buf, size_t bytes_per_rank);
/* the same for an in-process group: bufs[i] = rank i's copy of the buffer */
jxlh_status jxlh_comm_allgather_local(jxlh_ctx* const peers[], int32_t nranks, void* const bufs[], size_t bytes_per_rank);
/* Modular across GPUs.  RCT and the non-delta Palette are per-sample: a rank runs jxlh_rct /
 * jxlh_palette on its contiguous share of the samples (any split the caller likes; jxl_rs_amd/shard.py uses
 * ceil(n / nranks) rounded up to 4 samples so that every share starts 16-byte aligned) and the planes are joined with
 * jxlh_comm_allgather.  Squeeze is NOT sharded: the recurrence along a line is a serial dependency chain, and the
 * kernel's time is that chain's latency (one full-resolution vertical step takes 0.46 ms for one plane and for three,
 * profiles/r02_h_modular.txt) -- giving each GPU fewer lines leaves the chain as long as it was, so N GPUs run N
 * images (or the channels of one) as replicas instead. */

/* Gaborish / EPF on a Modular frame (a lossless or lossy-modular frame whose restoration filter is on): the stage list
 * of frame/render.rs:569-622 on three f32 planes the caller holds on the device (the output of jxlh_modular_to_f32 /
 * jxlh_modular_xyb_to_f32), with the CONSTANT sigma of features/epf.rs:81-84 -- INV_SIGMA_NUM / epf_sigma_for_modular
 * for every pixel -- instead of the per-block map a VarDCT frame derives from its quantisation field.  Uses gab, gab_w*,
 * epf_iters and the epf_* fields of p; w x h samples per plane, row stride `stride` floats (a multiple of 4; planes
 * 16-byte aligned); the result is written to out[], in[] may be overwritten (epf_iters == 3). */
jxlh_status jxlh_modular_frame_filters(jxlh_ctx* ctx, const jxlh_frame_params* p, float* const in[3],
                                       float* const out[3], uint32_t w, uint32_t h, size_t stride);

/* library info */
uint32_t jxlh_abi_version(void);
/* covered_blocks_x / _y and the type -> dequant table map (transform_map.rs:97-107,
 * quant_weights.rs:321-343), for hosts that build coefficient slabs */
int32_t jxlh_covered_blocks_x(int32_t type);
int32_t jxlh_covered_blocks_y(int32_t type);
int32_t jxlh_quant_table_for_type(int32_t type);
int32_t jxlh_quant_table_size(int32_t table);

#ifdef __cplusplus
}
#endif
#endif /* JXL_HIP_H_ */
