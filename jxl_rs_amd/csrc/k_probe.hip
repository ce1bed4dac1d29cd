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

// ---- placement probe (round 6, profiles/r06_q_context_placement.txt): the same kernels on the same data run up to 10 %
// apart between two contexts of one process -- where the driver placed a context's large buffers.  Two byte movers with
// the streams of K1's 8x8 class (2 KB of each channel of a group slab, 256 KB apart -> 2 KB of each plane) and of the
// filters (three planes in, three planes out) show the same spread on bare buffers, so a context can rate a placement
// before it commits to it (jxlh_ctx_tune_placement).
namespace {
__global__ __launch_bounds__(256) void k_probe_k1_like(const int32_t* __restrict__ coeffs, size_t ngroups, float* p0, float* p1,
                                                       float* p2, size_t plane_elems) {
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (size_t)gridDim.x * 4;
  const size_t nbatches = ngroups * 128;  // 512 coefficients (8 blocks) per batch and channel
  float* planes[3] = {p0, p1, p2};
  for (size_t b = wave; b < nbatches; b += nwaves) {
    const size_t g = b >> 7, i = b & 127;
    if ((b + 1) * 512 > plane_elems) break;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const int4* src = reinterpret_cast<const int4*>(coeffs + (g * 3 + c) * kGroupArea + i * 512);
      const int4 a = src[lane], q = src[lane + 64];
      float4* dst = reinterpret_cast<float4*>(planes[c] + b * 512);
      dst[lane] = make_float4((float)a.x, (float)a.y, (float)a.z, (float)a.w);
      dst[lane + 64] = make_float4((float)q.x, (float)q.y, (float)q.z, (float)q.w);
    }
  }
}
__global__ __launch_bounds__(256) void k_probe_filter_like(const float* __restrict__ p0, const float* __restrict__ p1,
                                                           const float* __restrict__ p2, float* t0, float* t1, float* t2,
                                                           size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 a = reinterpret_cast<const float4*>(p0)[i], b = reinterpret_cast<const float4*>(p1)[i],
                 c = reinterpret_cast<const float4*>(p2)[i];
    reinterpret_cast<float4*>(t0)[i] = make_float4(a.x + b.x, a.y, a.z, a.w);
    reinterpret_cast<float4*>(t1)[i] = make_float4(b.x + c.x, b.y, b.z, b.w);
    reinterpret_cast<float4*>(t2)[i] = make_float4(c.x + a.x, c.y, c.z, c.w);
  }
}
}  // namespace

namespace jxlh_host {
// average ms of the two movers on a candidate set of buffers (contents are overwritten / garbage is read: timing only)
jxlh_status probe_placement(jxlh_ctx* ctx, const int32_t* coeffs, size_t ngroups, float* const planes[3], float* const tmp[3],
                            size_t plane_elems, float* k1_like_ms, float* filter_like_ms) {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  jxlh_status st = JXLH_OK;
  auto body = [&]() -> jxlh_status {
    HIPCHK(ctx, hipEventCreate(&e0));
    HIPCHK(ctx, hipEventCreate(&e1));
    const int warm = 2, reps = 6;
    float ms[2] = {0.f, 0.f};
    for (int which = 0; which < 2; which++) {
      auto launch = [&]() {
        if (which == 0)
          hipLaunchKernelGGL(k_probe_k1_like, dim3(2048), dim3(256), 0, ctx->stream, coeffs, ngroups, planes[0], planes[1],
                             planes[2], plane_elems);
        else
          hipLaunchKernelGGL(k_probe_filter_like, dim3(4096), dim3(256), 0, ctx->stream, planes[0], planes[1], planes[2], tmp[0],
                             tmp[1], tmp[2], plane_elems / 4);
      };
      for (int i = 0; i < warm; i++) launch();
      HIPCHK(ctx, hipEventRecord(e0, ctx->stream));
      for (int i = 0; i < reps; i++) launch();
      HIPCHK(ctx, hipEventRecord(e1, ctx->stream));
      HIPCHK(ctx, hipEventSynchronize(e1));
      HIPCHK(ctx, hipEventElapsedTime(&ms[which], e0, e1));
      ms[which] /= reps;
    }
    *k1_like_ms = ms[0];
    *filter_like_ms = ms[1];
    return JXLH_OK;
  };
  st = body();
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  return st;
}
}  // namespace jxlh_host

// The probe on the context's own buffers (overwrites the pixel planes: between jxlh_frame_begin and the next jxlh_frame_run)
extern "C" jxlh_status jxlh_probe_placement(jxlh_ctx* ctx, float* k1_like_ms, float* filter_like_ms) {
  if (!ctx || !k1_like_ms || !filter_like_ms) return JXLH_ERR_INVALID_ARGUMENT;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (!ctx->in_frame || !ctx->coeffs.p || !ctx->planes[0].p) return JXLH_ERR_BAD_STATE;
  float* p[3] = {ctx->planes[0].p, ctx->planes[1].p, ctx->planes[2].p};
  float* t[3] = {ctx->tmp[0].p, ctx->tmp[1].p, ctx->tmp[2].p};
  const size_t plane_elems = std::min(std::min(ctx->planes[0].n, ctx->planes[1].n), ctx->planes[2].n) & ~(size_t)511;
  return jxlh_host::probe_placement(ctx, ctx->coeffs.p, ctx->ngroups, p, t, plane_elems, k1_like_ms, filter_like_ms);
}
