// Developer / bench instruments declared in include/jxl_hip_dev.h (NOT part of the product ABI in jxl_hip.h, not bound
// by the generated Rust -sys crate): HIP-event timers on the kernels' own stream, per-kernel timing, the exhaustive
// reciprocal self-test.  (jxlh_probe_copy_bandwidth lives next to its kernel in k_probe.hip.)
#include <algorithm>

#include "jxlh_ctx.h"
#include "../../include/jxl_hip_dev.h"

extern "C" {

// ---------------------------------------------------------------- timing
jxlh_status jxlh_timer_start(jxlh_ctx* ctx) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx) return JXLH_ERR_INVALID_ARGUMENT;
  HIPCHK(ctx, hipEventRecord(ctx->t0, ctx->stream));
  return JXLH_OK;
}

jxlh_status jxlh_timer_stop(jxlh_ctx* ctx, float* elapsed_ms) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !elapsed_ms) return JXLH_ERR_INVALID_ARGUMENT;
  HIPCHK(ctx, hipEventRecord(ctx->t1, ctx->stream));
  JXLH_SYNC(ctx);  // the event is the last thing on the stream; on a sharded context this wait has a deadline
  HIPCHK(ctx, hipEventSynchronize(ctx->t1));
  HIPCHK(ctx, hipEventElapsedTime(elapsed_ms, ctx->t0, ctx->t1));
  return JXLH_OK;
}

jxlh_status jxlh_kernel_timing_enable(jxlh_ctx* ctx, int32_t enable) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx) return JXLH_ERR_INVALID_ARGUMENT;
  ctx->timing = enable != 0;
  return JXLH_OK;
}

jxlh_status jxlh_kernel_timing_get(jxlh_ctx* ctx, int32_t i, const char** name, float* total_ms, int32_t* launches) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || i < 0) return JXLH_ERR_INVALID_ARGUMENT;
  drain_timers(ctx);
  if ((size_t)i >= ctx->ktimes.size()) return JXLH_ERR_INVALID_ARGUMENT;
  if (name) *name = ctx->ktimes[i].name.c_str();
  if (total_ms) *total_ms = ctx->ktimes[i].total_ms;
  if (launches) *launches = ctx->ktimes[i].launches;
  return JXLH_OK;
}

jxlh_status jxlh_kernel_timing_reset(jxlh_ctx* ctx) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx) return JXLH_ERR_INVALID_ARGUMENT;
  drain_timers(ctx);
  ctx->ktimes.clear();
  return JXLH_OK;
}

jxlh_status jxlh_frame_path(jxlh_ctx* ctx, int32_t* strip, int32_t* tiles, int32_t* tiles_by_class_kernels) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx) return JXLH_ERR_INVALID_ARGUMENT;
  if (strip) *strip = ctx->strip_ran ? 1 : 0;
  int32_t n = 0, k = 0;
  if (ctx->strip_ran && ctx->strip_mode.p) {
    n = (int32_t)(strip_strips(ctx->fd) * strip_tile_rows(ctx->fd));
    std::vector<uint8_t> m((size_t)n);
    HIPCHK(ctx, hipMemcpyAsync(m.data(), ctx->strip_mode.p, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    JXLH_SYNC(ctx);
    for (uint8_t v : m) k += v != 0;
  }
  if (tiles) *tiles = n;
  if (tiles_by_class_kernels) *tiles_by_class_kernels = k;
  return JXLH_OK;
}

jxlh_status jxlh_frame_k1_counters(jxlh_ctx* ctx, int32_t* out, int32_t n) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !out || n < 0) return JXLH_ERR_INVALID_ARGUMENT;
  if (!ctx->in_frame || !ctx->worklist.p || ctx->k1_launches == 0) return JXLH_ERR_BAD_STATE;
  int lines = 0;
  const int* h = nullptr;
  std::vector<int> host;
  {
    // the counter set of the last launch (launch n counts in set n & 1; the NEXT launch's scan clears the other one)
    size_t bytes = 0;
    const void* src = vardct_worklist_counters(ctx->worklist.p, ctx->k1_launches - 1, &bytes, &lines);
    host.resize(bytes / sizeof(int));
    HIPCHK(ctx, hipMemcpyAsync(host.data(), src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    JXLH_SYNC(ctx);
    h = host.data();
  }
  const int pitch = (int)(host.size() / (size_t)lines);
  // layout of the lines: classes, large-transform unit lists (4), fallback batches per class (9), dense-route lists (9)
  const int map_n = 29;
  for (int i = 0; i < n && i < map_n; i++) {
    const int line = i < 11 ? i : i < 20 ? 11 + 4 + (i - 11) : 11 + 4 + 9 + (i - 20);
    out[i] = h[(size_t)line * pitch];
  }
  return JXLH_OK;
}

jxlh_status jxlh_flow_profile(jxlh_ctx* ctx, int32_t enable, int32_t* n_levels, uint64_t* rows, int32_t max_levels) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx) return JXLH_ERR_INVALID_ARGUMENT;
  if (rows && n_levels && ctx->flow_prof_on && ctx->flow_prof.p && ctx->flow_prof_levels > 0) {
    const int cap = unsqueeze_flow_max_steps();
    std::vector<unsigned long long> h(11 * (size_t)cap);
    HIPCHK(ctx, hipMemcpyAsync(h.data(), ctx->flow_prof.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    JXLH_SYNC(ctx);
    const int n = std::min(ctx->flow_prof_levels, (int)max_levels);
    for (int i = 0; i < n; i++)
      for (int r = 0; r < 11; r++) rows[11 * i + r] = h[(size_t)r * cap + i];
    *n_levels = n;
  } else if (n_levels) {
    *n_levels = 0;
  }
  ctx->flow_prof_on = enable != 0;
  return JXLH_OK;
}

jxlh_status jxlh_selftest_recip(jxlh_ctx* ctx, uint32_t lo_bits, uint32_t hi_bits, uint64_t* mismatches) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !mismatches || hi_bits < lo_bits) return JXLH_ERR_INVALID_ARGUMENT;
  jxlh_status st;
  if ((st = ensure(ctx, ctx->hook_i[0], 2))) return st;
  unsigned long long* d = reinterpret_cast<unsigned long long*>(ctx->hook_i[0].p);
  HIPCHK(ctx, hipMemsetAsync(d, 0, sizeof(unsigned long long), ctx->stream));
  launch_selftest_recip(ctx->stream, lo_bits, hi_bits, d);
  HIPCHK(ctx, hipGetLastError());
  unsigned long long host = 0;
  HIPCHK(ctx, hipMemcpyAsync(&host, d, sizeof(host), hipMemcpyDeviceToHost, ctx->stream));
  JXLH_SYNC(ctx);
  *mismatches = host;
  return JXLH_OK;
}

}  // extern "C"
