// Device-to-device copy ceiling ("measure an on-device copy ceiling ... in the same run").  A float4
// grid-stride copy of `bytes` from one buffer to another, timed with HIP events on the context's stream: what a
// kernel that reads N bytes and writes N bytes can reach at all on this device for a working set of 2 * bytes
// (working sets below the 256 MiB Infinity Cache copy faster than frame-sized ones, profiles/r02_a_mall_probe.txt).
// Measured with plain and with `nt` accesses; the better rate is reported.
#include "jxlh_ctx.h"

namespace {
typedef float probe_f4 __attribute__((ext_vector_type(4)));
template <int MODE>  // 0 plain, 3 nt loads + nt stores (1 / 2: one stream only, tools/probe_policy.py history)
__global__ __launch_bounds__(256) void k_probe_copy(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const probe_f4* s = reinterpret_cast<const probe_f4*>(src);
  probe_f4* d = reinterpret_cast<probe_f4*>(dst);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const probe_f4 v = (MODE & 1) ? __builtin_nontemporal_load(s + i) : s[i];
    if (MODE & 2) __builtin_nontemporal_store(v, d + i);
    else d[i] = v;
  }
}
}  // namespace

extern "C" jxlh_status jxlh_probe_copy_bandwidth(jxlh_ctx* ctx, size_t bytes, int32_t reps, float* gb_per_s) {
  if (!ctx || !gb_per_s || bytes < 16 || reps < 1) return JXLH_ERR_INVALID_ARGUMENT;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const size_t n = bytes / 16;
  float4 *a = nullptr, *b = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&a), n * 16) != hipSuccess) return JXLH_ERR_OUT_OF_MEMORY;
  if (hipMalloc(reinterpret_cast<void**>(&b), n * 16) != hipSuccess) {
    (void)hipFree(a);
    return JXLH_ERR_OUT_OF_MEMORY;
  }
  jxlh_status st = JXLH_OK;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  auto body = [&]() -> jxlh_status {
    HIPCHK(ctx, hipMemsetAsync(a, 1, n * 16, ctx->stream));
    HIPCHK(ctx, hipEventCreate(&e0));
    HIPCHK(ctx, hipEventCreate(&e1));
    size_t g = (n + 256 * 8 - 1) / (256 * 8);
    if (g > 8192) g = 8192;
    if (g < 1) g = 1;
    // two cache policies, the better one counts: plain accesses, and `nt` on both streams (read once / not read again
    // by this kernel), which copies a frame-sized buffer ~10 % faster on this device (4.8 vs 5.3 TB/s; a buffer that
    // fits the Infinity Cache prefers plain: 6.5 vs 5.9)
    double best = 0.0;
    for (int mode = 0; mode < 2; mode++) {
      auto launch = [&]() {
        if (mode) hipLaunchKernelGGL(k_probe_copy<3>, dim3((unsigned)g), dim3(256), 0, ctx->stream, a, b, n);
        else hipLaunchKernelGGL(k_probe_copy<0>, dim3((unsigned)g), dim3(256), 0, ctx->stream, a, b, n);
      };
      for (int i = 0; i < 2; i++) launch();
      HIPCHK(ctx, hipEventRecord(e0, ctx->stream));
      for (int i = 0; i < reps; i++) launch();
      HIPCHK(ctx, hipEventRecord(e1, ctx->stream));
      HIPCHK(ctx, hipEventSynchronize(e1));
      float ms = 0.f;
      HIPCHK(ctx, hipEventElapsedTime(&ms, e0, e1));
      const double rate = 2.0 * (double)(n * 16) * reps / ((double)ms * 1e-3) / 1e9;
      if (rate > best) best = rate;
    }
    *gb_per_s = (float)best;
    return JXLH_OK;
  };
  st = body();
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(a);
  (void)hipFree(b);
  return st;
}
