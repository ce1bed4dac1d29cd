// C ABI, stage-level hooks (SURVEY.md 8(b) "stage-level test hooks"): one reference stage on caller planes, the
// analogue of make_and_run_simple_pipeline.  Each hook stages its arguments into context-owned device scratch (so
// host pointers work), runs the kernel(s) on the main stream and copies the result back.
#include <algorithm>

#include "jxlh_ctx.h"

extern "C" {

jxlh_status jxlh_stage_gaborish(jxlh_ctx* ctx, const float* in, float* out, uint32_t w, uint32_t h, size_t stride,
                                float w1, float w2) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !in || !out || stride < w) return JXLH_ERR_INVALID_ARGUMENT;
  if (w == 0 || h == 0) return JXLH_OK;
  const size_t n = stride * h;
  jxlh_status st;
  if ((st = stage_in(ctx, ctx->hook_f[0], in, n))) return st;
  if ((st = ensure(ctx, ctx->hook_f[1], n))) return st;
  const float total = 1.0f + w1 * 4.0f + w2 * 4.0f;
  launch_gaborish(ctx->stream, ctx->hook_f[0].p, ctx->hook_f[1].p, (int)w, (int)h, stride, 1.0f / total, w1 / total,
                  w2 / total, 0, (int)h);
  HIPCHK(ctx, hipGetLastError());
  return stage_out(ctx, out, ctx->hook_f[1].p, n);
}

jxlh_status jxlh_stage_epf(jxlh_ctx* ctx, int32_t stage, const jxlh_frame_params* p, const float* const in[3],
                           float* const out[3], uint32_t w, uint32_t h, size_t stride, const float* inv_sigma,
                           size_t sigma_stride) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !p || !in || !out || !inv_sigma || stage < 0 || stage > 2 || stride < w || sigma_stride < (w + 7) / 8)
    return JXLH_ERR_INVALID_ARGUMENT;
  if (w == 0 || h == 0) return JXLH_OK;
  const size_t n = stride * h;
  const size_t ns = sigma_stride * ((h + 7) / 8);
  jxlh_status st;
  EpfArgs a;
  for (int c = 0; c < 3; c++) {
    if (!in[c] || !out[c]) return JXLH_ERR_INVALID_ARGUMENT;
    if ((st = stage_in(ctx, ctx->hook_f[c], in[c], n))) return st;
    if ((st = ensure(ctx, ctx->hook_f[3 + c], n))) return st;
    a.in[c] = ctx->hook_f[c].p;
    a.out[c] = ctx->hook_f[3 + c].p;
    a.scale[c] = p->epf_channel_scale[c];
  }
  if ((st = stage_in(ctx, ctx->hook_f[6], inv_sigma, ns))) return st;
  a.inv_sigma = ctx->hook_f[6].p;
  a.stride = stride;
  a.sigma_stride = sigma_stride;
  a.w = (int)w;
  a.h = (int)h;
  const float sigma_scale = stage == 0 ? p->epf_pass0_sigma_scale : (stage == 1 ? 1.0f : p->epf_pass2_sigma_scale);
  a.sm = sigma_scale * 1.65f;
  a.bsm = a.sm * p->epf_border_sad_mul;
  launch_epf(ctx->stream, stage, a, 0, (int)h);
  HIPCHK(ctx, hipGetLastError());
  for (int c = 0; c < 3; c++)
    if ((st = stage_out(ctx, out[c], ctx->hook_f[3 + c].p, n))) return st;
  return JXLH_OK;
}

jxlh_status jxlh_modular_frame_filters(jxlh_ctx* ctx, const jxlh_frame_params* p, float* const in[3],
                                       float* const out[3], uint32_t w, uint32_t h, size_t stride) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !p || !in || !out || stride < w || (stride & 3) || p->epf_iters > 3) return JXLH_ERR_INVALID_ARGUMENT;
  for (int c = 0; c < 3; c++) {
    if (!in[c] || !out[c] || in[c] == out[c] || !is_device_ptr(in[c]) || !is_device_ptr(out[c]) ||
        (reinterpret_cast<uintptr_t>(in[c]) & 15) || (reinterpret_cast<uintptr_t>(out[c]) & 15))
      return JXLH_ERR_INVALID_ARGUMENT;
  }
  if (!(p->epf_sigma_for_modular > 0.0f)) return JXLH_ERR_INVALID_ARGUMENT;
  if (w == 0 || h == 0) return JXLH_OK;
  FrameDev f{};
  f.xsize = (int)w;
  f.ysize = (int)h;
  f.xblocks = (int)((w + 7) / 8);
  f.yblocks = (int)((h + 7) / 8);
  f.plane_stride = stride;
  f.tiled = 0;
  set_filter_params(f, *p);
  for (int c = 0; c < 3; c++) {
    f.planes[c] = in[c];
    f.tmp[c] = out[c];
  }
  // SigmaSource::Constant (features/epf.rs:81-84): one value for every block
  const size_t nb = (size_t)f.xblocks * f.yblocks;
  if (jxlh_status st = ensure(ctx, ctx->hook_f[7], nb)) return st;
  if (f.epf_iters > 0) {
    const float sigma = kInvSigmaNum / p->epf_sigma_for_modular;
    std::vector<float> host(nb, sigma);
    HIPCHK(ctx, hipMemcpyAsync(ctx->hook_f[7].p, host.data(), nb * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    JXLH_SYNC(ctx);  // `host` goes out of scope
  }
  f.inv_sigma = ctx->hook_f[7].p;
  int where;
  {
    ScopedKernelTimer t(ctx, "k23_fused_filters");
    where = launch_fused_filters(ctx->stream, f, 0, (int)h);
  }
  HIPCHK(ctx, hipGetLastError());
  if (where != 1) {  // no stage at all, or a stage list that ends in its input planes (epf_iters == 3)
    for (int c = 0; c < 3; c++)
      HIPCHK(ctx, hipMemcpy2DAsync(out[c], stride * sizeof(float), in[c], stride * sizeof(float), (size_t)w * sizeof(float), h,
                                   hipMemcpyDeviceToDevice, ctx->stream));
  }
  return JXLH_OK;
}

jxlh_status jxlh_stage_lf_smooth(jxlh_ctx* ctx, const jxlh_frame_params* p, const float* const in[3],
                                 float* const out[3], uint32_t w, uint32_t h) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !p || !in || !out || p->global_scale == 0 || p->quant_lf == 0) return JXLH_ERR_INVALID_ARGUMENT;
  if (w == 0 || h == 0) return JXLH_OK;
  const size_t n = (size_t)w * h;
  jxlh_status st;
  const float* din[3];
  float* dout[3];
  for (int c = 0; c < 3; c++) {
    if (!in[c] || !out[c]) return JXLH_ERR_INVALID_ARGUMENT;
    if ((st = stage_in(ctx, ctx->hook_f[c], in[c], n))) return st;
    if ((st = ensure(ctx, ctx->hook_f[3 + c], n))) return st;
    din[c] = ctx->hook_f[c].p;
    dout[c] = ctx->hook_f[3 + c].p;
  }
  if (w <= 2 || h <= 2) {  // adaptive_lf_smoothing.rs:51-53: untouched
    for (int c = 0; c < 3; c++)
      if ((st = stage_out(ctx, out[c], din[c], n))) return st;
    return JXLH_OK;
  }
  const float inv_quant_lf = ((float)(1 << 16) / (float)p->global_scale) / (float)p->quant_lf;
  const float lf_factors[3] = {inv_quant_lf * p->lf_quant_factors[0], inv_quant_lf * p->lf_quant_factors[1],
                               inv_quant_lf * p->lf_quant_factors[2]};
  launch_lf_smooth(ctx->stream, din, dout, (int)w, (int)h, lf_factors);
  HIPCHK(ctx, hipGetLastError());
  for (int c = 0; c < 3; c++)
    if ((st = stage_out(ctx, out[c], dout[c], n))) return st;
  return JXLH_OK;
}

jxlh_status jxlh_stage_chroma_upsample(jxlh_ctx* ctx, const float* in, float* out, uint32_t w, uint32_t h,
                                       int32_t horizontal) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !in || !out || w > (1u << 20) || h > (1u << 20)) return JXLH_ERR_INVALID_ARGUMENT;
  if (w == 0 || h == 0) return JXLH_OK;
  const size_t n = (size_t)w * h;
  if (2 * n >= (1ull << 31)) return JXLH_ERR_UNSUPPORTED;
  jxlh_status st;
  if ((st = stage_in(ctx, ctx->hook_f[0], in, n))) return st;
  if ((st = ensure(ctx, ctx->hook_f[1], 2 * n))) return st;
  const int ow = horizontal ? 2 * (int)w : (int)w, oh = horizontal ? (int)h : 2 * (int)h;
  PixLayout sl, dl;
  sl.tiled = dl.tiled = 0;
  sl.ystep8 = (int)w;
  sl.ystep_blk = 8 * (int)w;
  dl.ystep8 = ow;
  dl.ystep_blk = 8 * ow;
  launch_chroma_upsample(ctx->stream, ctx->hook_f[0].p, ctx->hook_f[1].p, sl, dl, horizontal ? 1 : 0, horizontal ? 0 : 1,
                         (int)w, (int)h, 0, (int)h, ow, oh);
  HIPCHK(ctx, hipGetLastError());
  return stage_out(ctx, out, ctx->hook_f[1].p, 2 * n);
}

jxlh_status jxlh_stage_upsample(jxlh_ctx* ctx, int32_t n, const float* in, float* out, uint32_t w, uint32_t h) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !in || !out || (n != 2 && n != 4 && n != 8) || w > (1u << 20) || h > (1u << 20))
    return JXLH_ERR_INVALID_ARGUMENT;
  if (w == 0 || h == 0) return JXLH_OK;
  const size_t ni = (size_t)w * h, no = ni * (size_t)n * n;
  if (no >= (1ull << 32)) return JXLH_ERR_UNSUPPORTED;
  jxlh_status st;
  if ((st = stage_in(ctx, ctx->hook_f[0], in, ni))) return st;
  if ((st = ensure(ctx, ctx->hook_f[1], no))) return st;
  if ((st = upload_upsampling_kernels(ctx, n))) return st;
  launch_upsample(ctx->stream, n, ctx->hook_f[0].p, w, (int)w, (int)h, ctx->ups_kernels.p, ctx->hook_f[1].p,
                  (size_t)w * n, (int)w * n, (int)h * n);
  HIPCHK(ctx, hipGetLastError());
  return stage_out(ctx, out, ctx->hook_f[1].p, no);
}

jxlh_status jxlh_stage_noise_generate(jxlh_ctx* ctx, uint32_t visible_frame_index, uint32_t nonvisible_frame_index,
                                      uint32_t w, uint32_t h, float* const out[3]) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !out || !out[0] || !out[1] || !out[2] || w == 0 || h == 0 || w > (1u << 20) || h > (1u << 20))
    return JXLH_ERR_INVALID_ARGUMENT;
  const size_t n = (size_t)w * h;
  jxlh_status st;
  if ((st = ensure_jump_table(ctx))) return st;
  float* d[3];
  for (int c = 0; c < 3; c++) {
    if ((st = ensure(ctx, ctx->hook_f[c], n))) return st;
    d[c] = ctx->hook_f[c].p;
  }
  launch_noise_generate(ctx->stream, d, w, (int)w, (int)h, 0, ((int)h + 255) / 256, visible_frame_index,
                        nonvisible_frame_index, ctx->xs_jump.p);
  HIPCHK(ctx, hipGetLastError());
  for (int c = 0; c < 3; c++)
    if ((st = stage_out(ctx, out[c], d[c], n))) return st;
  return JXLH_OK;
}

jxlh_status jxlh_stage_noise_convolve(jxlh_ctx* ctx, const float* in, float* out, uint32_t w, uint32_t h) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !in || !out || w > (1u << 20) || h > (1u << 20)) return JXLH_ERR_INVALID_ARGUMENT;
  if (w == 0 || h == 0) return JXLH_OK;
  const size_t n = (size_t)w * h;
  jxlh_status st;
  if ((st = stage_in(ctx, ctx->hook_f[0], in, n))) return st;
  if ((st = ensure(ctx, ctx->hook_f[1], n))) return st;
  launch_noise_convolve(ctx->stream, ctx->hook_f[0].p, ctx->hook_f[1].p, (int)w, (int)h);
  HIPCHK(ctx, hipGetLastError());
  return stage_out(ctx, out, ctx->hook_f[1].p, n);
}

jxlh_status jxlh_stage_noise_add(jxlh_ctx* ctx, const jxlh_frame_params* p, float* const planes[3],
                                 const float* const rnd[3], size_t n) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !p || !planes || !rnd || p->color_factor == 0) return JXLH_ERR_INVALID_ARGUMENT;
  if (n == 0 || noise_lut_is_zero(p->noise_lut)) return JXLH_OK;
  jxlh_status st;
  float* dp[3];
  const float* dr[3];
  for (int c = 0; c < 3; c++) {
    if (!planes[c] || !rnd[c]) return JXLH_ERR_INVALID_ARGUMENT;
    if ((st = stage_in(ctx, ctx->hook_f[c], planes[c], n))) return st;
    if ((st = stage_in(ctx, ctx->hook_f[3 + c], rnd[c], n))) return st;
    dp[c] = ctx->hook_f[c].p;
    dr[c] = ctx->hook_f[3 + c].p;
  }
  const float ytox = p->base_correlation_x + (float)p->ytox_lf / (float)p->color_factor;
  const float ytob = p->base_correlation_b + (float)p->ytob_lf / (float)p->color_factor;
  launch_noise_add(ctx->stream, dp, dr, n, p->noise_lut, ytox, ytob);
  HIPCHK(ctx, hipGetLastError());
  for (int c = 0; c < 3; c++)
    if ((st = stage_out(ctx, planes[c], dp[c], n))) return st;
  return JXLH_OK;
}

jxlh_status jxlh_stage_transform_to_pixels(jxlh_ctx* ctx, int32_t type, uint32_t n, const float* coeffs,
                                           const float* lf, float* pixels) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || type < 0 || type >= JXLH_NUM_TRANSFORMS || !coeffs || !lf || !pixels) return JXLH_ERR_INVALID_ARGUMENT;
  if (n == 0) return JXLH_OK;
  const size_t nb = (size_t)covered_x(type) * covered_y(type);
  jxlh_status st;
  if ((st = stage_in(ctx, ctx->hook_f[0], coeffs, n * nb * 64))) return st;
  if ((st = stage_in(ctx, ctx->hook_f[1], lf, n * nb))) return st;
  if ((st = ensure(ctx, ctx->hook_f[2], n * nb * 64))) return st;
  launch_transform_to_pixels(ctx->stream, type, n, ctx->hook_f[0].p, ctx->hook_f[1].p, ctx->hook_f[2].p);
  HIPCHK(ctx, hipGetLastError());
  return stage_out(ctx, pixels, ctx->hook_f[2].p, n * nb * 64);
}

}  // extern "C"
