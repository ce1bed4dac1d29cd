/*
 * jxl_hip_dev.h -- developer / bench instruments of libjxl_hip.so.
 *
 * NOT part of the drop-in boundary: a decoder binds include/jxl_hip.h only (tools/gen_rust_binding.py generates the
 * Rust -sys crate from that header alone).  These entry points exist for bench.py, the profiling tools and the GPU
 * tests: HIP-event timers on the stream the kernels run on, per-kernel timing, a measured copy ceiling and a device
 * self-test.  They are exported by the same library and follow the same conventions (status codes, no exceptions).
 */
#ifndef JXL_HIP_DEV_H_
#define JXL_HIP_DEV_H_

#include "jxl_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* HIP-event timing on the stream the kernels are launched on. */
jxlh_status jxlh_timer_start(jxlh_ctx* ctx);
jxlh_status jxlh_timer_stop(jxlh_ctx* ctx, float* elapsed_ms);
/* per-kernel event pairs; accumulates while enabled */
jxlh_status jxlh_kernel_timing_enable(jxlh_ctx* ctx, int32_t enable);
/* i-th kernel that ran while timing was enabled: name, total ms, launches. Returns
 * JXLH_ERR_INVALID_ARGUMENT past the end. */
jxlh_status jxlh_kernel_timing_get(jxlh_ctx* ctx, int32_t i, const char** name, float* total_ms,
                                   int32_t* launches);
jxlh_status jxlh_kernel_timing_reset(jxlh_ctx* ctx);
/* Device-to-device copy ceiling, measured: a float4 copy of `bytes` (src and dst buffers allocated for the call)
 * repeated `reps` times on the context's stream, once with plain and once with non-temporal accesses;
 * *gb_per_s = (bytes read + bytes written) / time of the faster policy.  The yardstick SURVEY.md 8(d) asks for next
 * to the 8 TB/s spec peak. */
jxlh_status jxlh_probe_copy_bandwidth(jxlh_ctx* ctx, size_t bytes, int32_t reps, float* gb_per_s);
/* How well the context's large buffers are PLACED (round 6, profiles/r06_q_context_placement.txt: the same kernels on the
 * same data run up to 10 % apart between two contexts of one process; copies between the buffers do not show it).  Two byte
 * movers with the streams of the 8x8 transform class (coefficient buffer -> three planes) and of the filters (three planes
 * -> three planes), average ms per launch.  Overwrites the pixel planes: call between jxlh_frame_begin and the next
 * jxlh_frame_run (JXLH_ERR_BAD_STATE outside a frame). */
jxlh_status jxlh_probe_placement(jxlh_ctx* ctx, float* k1_like_ms, float* filter_like_ms);


/* Which path the last jxlh_frame_run of the current frame took: *strip = 1 if the single strip kernel (k123_strip)
 * produced the result; *tiles = 64x64 tiles of the frame, *tiles_by_class_kernels = how many of them were left to the
 * transform class kernels (varblocks that leave their tile, special / large transforms) and only loaded by the strip
 * kernel.  All 0 for the two-kernel path.  Synchronises the context's stream. */
jxlh_status jxlh_frame_path(jxlh_ctx* ctx, int32_t* strip, int32_t* tiles, int32_t* tiles_by_class_kernels);

/* Work-list counters of the last transform launch (k1_scan's class lists), for the bench legs that sweep the content
 * of the slot-bucketed form: out[0..10] = varblocks per transform class (DCT8, 16x8, 8x16, 16x16, 32x8, 8x32, 32x16,
 * 16x32, 32x32, special, large -- of the groups read in place, or of all groups for a frame of dense slabs),
 * out[11..19] = batches of the classes DCT8 .. 32x32 that left the direct path of the entries form for the dense
 * dequantisation pass (more entries than their lanes hold, raw_quant == 0: inline for DCT8, the fallback launch for the
 * others), out[20..28] = varblocks of those classes in the groups routed to their dense slabs.  n = ints `out` holds
 * (29 for everything).  Synchronises the context's stream. */
jxlh_status jxlh_frame_k1_counters(jxlh_ctx* ctx, int32_t* out, int32_t n);

/* Device self-test of the EPF weight normalisation: the filters compute 1/(1 + sum of weights)
 * (epf0.rs:208, epf1.rs:140, epf2.rs:130 divide) with rcp + two FMA refinement steps.  Counts the
 * floats whose bit pattern lies in [lo_bits, hi_bits) for which that differs from the IEEE
 * quotient 1.0f / w; the filters rely on 0 mismatches over [1.0f, 16.0f). */
jxlh_status jxlh_selftest_recip(jxlh_ctx* ctx, uint32_t lo_bits, uint32_t hi_bits, uint64_t* mismatches);

/* Timeline of the last dataflow launch of jxlh_unsqueeze_chain (k6_unsqueeze_flow: the streamed squeeze levels of a
 * chain in one launch, levels overlapping).  Call with enable = 1 before the chain; a later call returns, per level of
 * that launch, rows[11 i + 0..4] = first workgroup start, last workgroup end, time the first mover wave of every
 * workgroup spent polling progress words (sum), its polls (sum), workgroup lifetimes (sum) -- times in s_memrealtime
 * ticks (100 MHz); rows[11 i + 5..10] are zero unless the library was built with the mover-phase experiment.  The profile costs two small memsets and five atomics per workgroup; off by default. */
jxlh_status jxlh_flow_profile(jxlh_ctx* ctx, int32_t enable, int32_t* n_levels, uint64_t* rows, int32_t max_levels);

#ifdef __cplusplus
}
#endif
#endif /* JXL_HIP_DEV_H_ */
