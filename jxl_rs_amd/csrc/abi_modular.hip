// C ABI, Modular transforms: RCT, Palette (plain, delta / predicted, Weighted), Squeeze (steps, fused chains,
// progressive smooth steps) and the bridges from Modular channels to pixels.
#include <algorithm>

#include "jxlh_ctx.h"

extern "C" {

// ---------------------------------------------------------------- Modular
// In-place / out-of-place on the caller's buffers when they are device pointers; host
// pointers are staged through context scratch.
jxlh_status jxlh_rct(jxlh_ctx* ctx, int32_t* p0, int32_t* p1, int32_t* p2, size_t n, int32_t op, int32_t perm) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !p0 || !p1 || !p2 || op < 0 || op > 6 || perm < 0 || perm > 5) return JXLH_ERR_INVALID_ARGUMENT;
  if (n == 0) return JXLH_OK;
  if (is_device_ptr(p0) && is_device_ptr(p1) && is_device_ptr(p2)) {
    ScopedKernelTimer t(ctx, "k4_rct");
    launch_rct(ctx->stream, p0, p1, p2, n, op, perm);
    HIPCHK(ctx, hipGetLastError());
    return JXLH_OK;
  }
  jxlh_status st;
  int32_t* h[3] = {p0, p1, p2};
  for (int c = 0; c < 3; c++)
    if ((st = stage_in(ctx, ctx->hook_i[c], (const int32_t*)h[c], n))) return st;
  launch_rct(ctx->stream, ctx->hook_i[0].p, ctx->hook_i[1].p, ctx->hook_i[2].p, n, op, perm);
  HIPCHK(ctx, hipGetLastError());
  for (int c = 0; c < 3; c++)
    if ((st = stage_out(ctx, h[c], (const int32_t*)ctx->hook_i[c].p, n))) return st;
  return JXLH_OK;
}

jxlh_status jxlh_palette(jxlh_ctx* ctx, const int32_t* index, size_t n, const int32_t* palette, int32_t num_colors,
                         size_t palette_stride, int32_t nb_channels, int32_t bit_depth, int32_t* out) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !index || !palette || !out || num_colors < 0 || nb_channels < 1 || nb_channels > 64 || bit_depth < 1 ||
      bit_depth > 24 || palette_stride < (size_t)num_colors)
    return JXLH_ERR_INVALID_ARGUMENT;
  if (n == 0) return JXLH_OK;
  const size_t pal_n = palette_stride * (size_t)nb_channels;
  if (is_device_ptr(index) && is_device_ptr(palette) && is_device_ptr(out)) {
    ScopedKernelTimer t(ctx, "k5_palette");
    launch_palette(ctx->stream, index, n, palette, num_colors, palette_stride, nb_channels, bit_depth, out);
    HIPCHK(ctx, hipGetLastError());
    return JXLH_OK;
  }
  jxlh_status st;
  if ((st = stage_in(ctx, ctx->hook_i[0], index, n))) return st;
  if ((st = stage_in(ctx, ctx->hook_i[1], palette, pal_n ? pal_n : 1))) return st;
  if ((st = ensure(ctx, ctx->hook_i[2], n * nb_channels))) return st;
  launch_palette(ctx->stream, ctx->hook_i[0].p, n, ctx->hook_i[1].p, num_colors, palette_stride, nb_channels,
                 bit_depth, ctx->hook_i[2].p);
  HIPCHK(ctx, hipGetLastError());
  return stage_out(ctx, out, (const int32_t*)ctx->hook_i[2].p, n * nb_channels);
}

jxlh_status jxlh_palette_strided(jxlh_ctx* ctx, const int32_t* index, size_t n, const int32_t* palette,
                                 int32_t num_colors, size_t palette_stride, int32_t nb_channels, int32_t bit_depth,
                                 int32_t* out, size_t out_channel_stride) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !index || !palette || !out || num_colors < 0 || nb_channels < 1 || nb_channels > 64 || bit_depth < 1 ||
      bit_depth > 24 || palette_stride < (size_t)num_colors || out_channel_stride < n)
    return JXLH_ERR_INVALID_ARGUMENT;
  if (!is_device_ptr(index) || !is_device_ptr(palette) || !is_device_ptr(out)) return JXLH_ERR_INVALID_ARGUMENT;
  if (n == 0) return JXLH_OK;
  ScopedKernelTimer t(ctx, "k5_palette");
  launch_palette(ctx->stream, index, n, palette, num_colors, palette_stride, nb_channels, bit_depth, out,
                 out_channel_stride);
  HIPCHK(ctx, hipGetLastError());
  return JXLH_OK;
}

jxlh_status jxlh_palette_delta(jxlh_ctx* ctx, const int32_t* index, uint32_t w, uint32_t h, const int32_t* palette,
                               int32_t num_colors, int32_t num_deltas, size_t palette_stride, int32_t nb_channels,
                               int32_t bit_depth, int32_t predictor, int32_t* out) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !index || !palette || !out || num_colors < 0 || num_deltas < 0 || nb_channels < 1 || nb_channels > 64 ||
      bit_depth < 1 || bit_depth > 24 || palette_stride < (size_t)num_colors + (size_t)num_deltas || predictor < 0 ||
      predictor > 13 || w > (1u << 20) || h > (1u << 20))
    return JXLH_ERR_INVALID_ARGUMENT;
  if (predictor == 6) return JXLH_ERR_UNSUPPORTED;  // Weighted: its own stateful branch (palette.rs:200-227), host
  if (w == 0 || h == 0) return JXLH_OK;
  const size_t n = (size_t)w * h, pal_n = palette_stride * (size_t)nb_channels;
  if (n * (size_t)nb_channels >= (1ull << 31)) return JXLH_ERR_UNSUPPORTED;
  if (jxlh_status st0 = ensure(ctx, ctx->hook_i[3], (size_t)nb_channels * palette_delta_bands((int)h))) return st0;
  int* progress = reinterpret_cast<int*>(ctx->hook_i[3].p);
  if (is_device_ptr(index) && is_device_ptr(palette) && is_device_ptr(out)) {
    ScopedKernelTimer t(ctx, "k5_palette_delta");
    launch_palette_delta(ctx->stream, index, (int)w, (int)h, palette, num_colors, num_deltas, palette_stride,
                         nb_channels, bit_depth, predictor, out, progress);
    HIPCHK(ctx, hipGetLastError());
    return JXLH_OK;
  }
  jxlh_status st;
  if ((st = stage_in(ctx, ctx->hook_i[0], index, n))) return st;
  if ((st = stage_in(ctx, ctx->hook_i[1], palette, pal_n ? pal_n : 1))) return st;
  if ((st = ensure(ctx, ctx->hook_i[2], n * nb_channels))) return st;
  launch_palette_delta(ctx->stream, ctx->hook_i[0].p, (int)w, (int)h, ctx->hook_i[1].p, num_colors, num_deltas,
                       palette_stride, nb_channels, bit_depth, predictor, ctx->hook_i[2].p, progress);
  HIPCHK(ctx, hipGetLastError());
  return stage_out(ctx, out, (const int32_t*)ctx->hook_i[2].p, n * nb_channels);
}

// ---- Modular channels -> pipeline samples (render/stages/convert.rs)
jxlh_status jxlh_modular_to_rgb8(jxlh_ctx* ctx, const int32_t* const planes[3], size_t stride, uint32_t w, uint32_t h,
                                 int32_t multiplier, int32_t max, uint32_t channels, void* out, size_t bytes_per_row) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !planes || !planes[0] || !planes[1] || !planes[2] || !out || stride < w || (channels != 3 && channels != 4) ||
      bytes_per_row < (size_t)w * channels || max < 0 || max > 255 || w > (1u << 20) || h > (1u << 20))
    return JXLH_ERR_INVALID_ARGUMENT;
  if (w == 0 || h == 0) return JXLH_OK;
  const bool dev_in = is_device_ptr(planes[0]) && is_device_ptr(planes[1]) && is_device_ptr(planes[2]);
  const int32_t* src[3] = {planes[0], planes[1], planes[2]};
  size_t sstride = stride;
  jxlh_status st;
  if (!dev_in) {
    const size_t n = (size_t)stride * h;
    for (int c = 0; c < 3; c++) {
      if ((st = stage_in(ctx, ctx->hook_i[c], planes[c], n))) return st;
      src[c] = ctx->hook_i[c].p;
    }
  }
  ScopedKernelTimer t(ctx, "k_i32_to_rgb8");
  if (is_device_ptr(out)) {
    launch_i32_to_rgb8(ctx->stream, src, sstride, (int)w, (int)h, multiplier, max, (int)channels,
                       static_cast<uint8_t*>(out), bytes_per_row);
    HIPCHK(ctx, hipGetLastError());
    return JXLH_OK;
  }
  const size_t tight = (size_t)w * channels;
  if ((st = ensure(ctx, ctx->rgb8, tight * (size_t)h))) return st;
  launch_i32_to_rgb8(ctx->stream, src, sstride, (int)w, (int)h, multiplier, max, (int)channels, ctx->rgb8.p, tight);
  HIPCHK(ctx, hipGetLastError());
  if ((st = copy2d(ctx, out, bytes_per_row, ctx->rgb8.p, tight, tight, (size_t)h, ctx->stream))) return st;
  JXLH_SYNC(ctx);
  return JXLH_OK;
}

jxlh_status jxlh_modular_to_f32(jxlh_ctx* ctx, const int32_t* in, size_t n, uint32_t bits_per_sample, float* out) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !in || !out || !bit_depth_ok(bits_per_sample, 32)) return JXLH_ERR_INVALID_ARGUMENT;
  if (n == 0) return JXLH_OK;
  const uint32_t bits = bits_per_sample & 0xffu, exp_bits = bits_per_sample >> 8;  // BitDepth::floating_point_sample()
  const float scale = 1.0f / (float)((1ull << bits) - 1);  // convert.rs:528
  auto convert = [&](const int32_t* src, float* dst) {
    if (exp_bits) launch_float_samples_to_f32(ctx->stream, src, n, bits, exp_bits, dst);  // convert.rs:525-526
    else launch_modular_to_f32(ctx->stream, src, n, scale, dst);
  };
  if (is_device_ptr(in) && is_device_ptr(out)) {
    convert(in, out);
    HIPCHK(ctx, hipGetLastError());
    return JXLH_OK;
  }
  jxlh_status st;
  if ((st = stage_in(ctx, ctx->hook_i[0], in, n))) return st;
  if ((st = ensure(ctx, ctx->hook_f[0], n))) return st;
  convert(ctx->hook_i[0].p, ctx->hook_f[0].p);
  HIPCHK(ctx, hipGetLastError());
  return stage_out(ctx, out, (const float*)ctx->hook_f[0].p, n);
}

jxlh_status jxlh_modular_xyb_to_f32(jxlh_ctx* ctx, const int32_t* y, const int32_t* x, const int32_t* b, size_t n,
                                    const float quant_factors[3], float* ox, float* oy, float* ob) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !y || !x || !b || !quant_factors || !ox || !oy || !ob) return JXLH_ERR_INVALID_ARGUMENT;
  if (n == 0) return JXLH_OK;
  const int32_t* in[3] = {y, x, b};
  float* outp[3] = {ox, oy, ob};
  if (is_device_ptr(y) && is_device_ptr(x) && is_device_ptr(b) && is_device_ptr(ox) && is_device_ptr(oy) && is_device_ptr(ob)) {
    launch_modular_xyb_to_f32(ctx->stream, y, x, b, n, quant_factors, ox, oy, ob);
    HIPCHK(ctx, hipGetLastError());
    return JXLH_OK;
  }
  jxlh_status st;
  for (int c = 0; c < 3; c++) {
    if ((st = stage_in(ctx, ctx->hook_i[c], in[c], n))) return st;
    if ((st = ensure(ctx, ctx->hook_f[c], n))) return st;
  }
  launch_modular_xyb_to_f32(ctx->stream, ctx->hook_i[0].p, ctx->hook_i[1].p, ctx->hook_i[2].p, n, quant_factors,
                            ctx->hook_f[0].p, ctx->hook_f[1].p, ctx->hook_f[2].p);
  HIPCHK(ctx, hipGetLastError());
  for (int c = 0; c < 3; c++)
    if ((st = stage_out(ctx, outp[c], (const float*)ctx->hook_f[c].p, n))) return st;
  return JXLH_OK;
}

jxlh_status jxlh_palette_delta_wp(jxlh_ctx* ctx, const int32_t* index, uint32_t w, uint32_t h, const int32_t* palette,
                                  int32_t num_colors, int32_t num_deltas, size_t palette_stride, int32_t nb_channels,
                                  int32_t bit_depth, const jxlh_wp_header* wp, int32_t* out) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !index || !palette || !out || !wp || num_colors < 0 || num_deltas < 0 || nb_channels < 1 ||
      nb_channels > 64 || bit_depth < 1 || bit_depth > 24 ||
      palette_stride < (size_t)num_colors + (size_t)num_deltas || w > (1u << 20) || h > (1u << 20))
    return JXLH_ERR_INVALID_ARGUMENT;
  const uint32_t header[11] = {wp->p1c, wp->p2c, wp->p3ca, wp->p3cb, wp->p3cc, wp->p3cd, wp->p3ce,
                               wp->w0,  wp->w1,  wp->w2,   wp->w3};
  for (int i = 0; i < 11; i++)
    if (header[i] >= (i < 7 ? 32u : 16u)) return JXLH_ERR_INVALID_ARGUMENT;  // Bits(5) / Bits(4) fields
  if (w == 0 || h == 0) return JXLH_OK;
  const size_t n = (size_t)w * h, pal_n = palette_stride * (size_t)nb_channels;
  if (n * (size_t)nb_channels >= (1ull << 31)) return JXLH_ERR_UNSUPPORTED;
  const size_t nbands = (size_t)palette_delta_bands((int)h);
  // progress counters, then the band-edge rows of predictor state (5 rows of w per channel and band)
  const size_t n_prog = (size_t)nb_channels * nbands, n_rows = n_prog * 5 * (size_t)w;
  if (jxlh_status st0 = ensure(ctx, ctx->hook_i[3], n_prog + n_rows)) return st0;
  int* progress = reinterpret_cast<int*>(ctx->hook_i[3].p);
  int32_t* wp_rows = ctx->hook_i[3].p + n_prog;
  if (is_device_ptr(index) && is_device_ptr(palette) && is_device_ptr(out)) {
    ScopedKernelTimer t(ctx, "k5_palette_wp");
    launch_palette_wp(ctx->stream, index, (int)w, (int)h, palette, num_colors, num_deltas, palette_stride, nb_channels,
                      bit_depth, header, out, progress, wp_rows);
    HIPCHK(ctx, hipGetLastError());
    return JXLH_OK;
  }
  jxlh_status st;
  if ((st = stage_in(ctx, ctx->hook_i[0], index, n))) return st;
  if ((st = stage_in(ctx, ctx->hook_i[1], palette, pal_n ? pal_n : 1))) return st;
  if ((st = ensure(ctx, ctx->hook_i[2], n * nb_channels))) return st;
  launch_palette_wp(ctx->stream, ctx->hook_i[0].p, (int)w, (int)h, ctx->hook_i[1].p, num_colors, num_deltas,
                    palette_stride, nb_channels, bit_depth, header, ctx->hook_i[2].p, progress, wp_rows);
  HIPCHK(ctx, hipGetLastError());
  return stage_out(ctx, out, (const int32_t*)ctx->hook_i[2].p, n * nb_channels);
}

// dimension bound of the squeeze entry points (the kernels take `int` line counts / lengths)
static constexpr uint32_t kMaxModularDim = 1u << 20;

jxlh_status jxlh_unsqueeze(jxlh_ctx* ctx, int32_t horizontal, const int32_t* avg, size_t avg_stride,
                           const int32_t* res, size_t res_stride, uint32_t out_w, uint32_t out_h, int32_t* out,
                           size_t out_stride) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !avg || !out || out_stride < out_w) return JXLH_ERR_INVALID_ARGUMENT;
  if (out_w == 0 || out_h == 0) return JXLH_OK;
  if (out_w > kMaxModularDim || out_h > kMaxModularDim) return JXLH_ERR_UNSUPPORTED;
  const uint32_t avg_w = horizontal ? (out_w + 1) / 2 : out_w, avg_h = horizontal ? out_h : (out_h + 1) / 2;
  const uint32_t res_w = horizontal ? out_w / 2 : out_w, res_h = horizontal ? out_h : out_h / 2;
  const bool has_res = (size_t)res_w * res_h > 0;
  if (avg_stride < avg_w || (has_res && (!res || res_stride < res_w))) return JXLH_ERR_INVALID_ARGUMENT;
  if (is_device_ptr(avg) && is_device_ptr(out) && (!has_res || is_device_ptr(res))) {
    ScopedKernelTimer t(ctx, horizontal ? "k6_unsqueeze_h" : "k6_unsqueeze_v");
    const int32_t* av[1] = {avg};
    const int32_t* rv[1] = {res ? res : avg};
    int32_t* ov[1] = {out};
    launch_unsqueeze(ctx->stream, horizontal, 1, av, avg_stride, rv, res_stride, out_w, out_h, ov, out_stride);
    HIPCHK(ctx, hipGetLastError());
    return JXLH_OK;
  }
  jxlh_status st;
  if ((st = stage_in(ctx, ctx->hook_i[0], avg, avg_stride * avg_h))) return st;
  const size_t res_n = has_res ? res_stride * res_h : 0;
  if (res_n) {
    if ((st = stage_in(ctx, ctx->hook_i[1], res, res_n))) return st;
  } else if ((st = ensure(ctx, ctx->hook_i[1], 1))) {
    return st;
  }
  if ((st = ensure(ctx, ctx->hook_i[2], out_stride * out_h))) return st;
  {
    const int32_t* av[1] = {ctx->hook_i[0].p};
    const int32_t* rv[1] = {ctx->hook_i[1].p};
    int32_t* ov[1] = {ctx->hook_i[2].p};
    launch_unsqueeze(ctx->stream, horizontal, 1, av, avg_stride, rv, res_stride, out_w, out_h, ov, out_stride);
  }
  HIPCHK(ctx, hipGetLastError());
  return stage_out(ctx, out, (const int32_t*)ctx->hook_i[2].p, out_stride * out_h);
}

jxlh_status jxlh_unsqueeze_levels(jxlh_ctx* ctx, int32_t n_planes, int32_t n_levels, const jxlh_squeeze_level* levels,
                                  const int32_t* const base[], size_t base_stride, uint32_t base_w, uint32_t base_h,
                                  int32_t* const out[], size_t out_stride) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !levels || !base || !out || n_planes < 1 || n_planes > 3 || n_levels < 1 || n_levels > 64 || base_w == 0 ||
      base_h == 0 || base_stride < base_w)
    return JXLH_ERR_INVALID_ARGUMENT;
  // geometry: every level doubles (up to the odd sample) the axis it squeezes
  uint32_t cw = base_w, ch = base_h;
  for (int i = 0; i < n_levels; i++) {
    const jxlh_squeeze_level& lv = levels[i];
    if (lv.out_w == 0 || lv.out_h == 0 || lv.out_w > kMaxModularDim || lv.out_h > kMaxModularDim) return JXLH_ERR_INVALID_ARGUMENT;
    const uint32_t aw = lv.horizontal ? (lv.out_w + 1) / 2 : lv.out_w, ah = lv.horizontal ? lv.out_h : (lv.out_h + 1) / 2;
    if (aw != cw || ah != ch) return JXLH_ERR_INVALID_ARGUMENT;
    const uint32_t rw = lv.horizontal ? lv.out_w / 2 : lv.out_w, rh = lv.horizontal ? lv.out_h : lv.out_h / 2;
    for (int p = 0; p < n_planes; p++)
      if ((size_t)rw * rh > 0 && (!lv.res[p] || !is_device_ptr(lv.res[p]) || lv.res_stride < rw))
        return JXLH_ERR_INVALID_ARGUMENT;
    cw = lv.out_w;
    ch = lv.out_h;
  }
  if (out_stride < cw) return JXLH_ERR_INVALID_ARGUMENT;
  for (int p = 0; p < n_planes; p++)
    if (!base[p] || !out[p] || !is_device_ptr(base[p]) || !is_device_ptr(out[p])) return JXLH_ERR_INVALID_ARGUMENT;
  if (n_levels <= 16) {
    int hz[16];
    uint32_t ow[16], oh[16];
    size_t rs[16];
    const int32_t* rp[16 * 3];
    for (int i = 0; i < n_levels; i++) {
      hz[i] = levels[i].horizontal ? 1 : 0;
      ow[i] = levels[i].out_w;
      oh[i] = levels[i].out_h;
      rs[i] = levels[i].res_stride;
      for (int p = 0; p < 3; p++) rp[i * 3 + p] = p < n_planes && levels[i].res[p] ? levels[i].res[p] : base[0];
    }
    ScopedKernelTimer t(ctx, "k6_unsqueeze_levels");
    if (launch_unsqueeze_levels(ctx->stream, n_planes, n_levels, hz, ow, oh, rp, rs, base, base_stride, base_w, base_h,
                                out, out_stride)) {
      HIPCHK(ctx, hipGetLastError());
      return JXLH_OK;
    }
  }
  // anything else: the chain's route without an RCT (LDS-resident prefix, the streamed levels as one dataflow launch)
  return jxlh_unsqueeze_chain(ctx, n_planes, n_levels, levels, base, base_stride, base_w, base_h, out, out_stride, -1, 0);
}

// The inverse of a whole squeeze transform as one call (+ the RCT that follows it in the transform list).
jxlh_status jxlh_unsqueeze_chain(jxlh_ctx* ctx, int32_t n_planes, int32_t n_levels, const jxlh_squeeze_level* levels,
                                 const int32_t* const base[], size_t base_stride, uint32_t base_w, uint32_t base_h,
                                 int32_t* const out[], size_t out_stride, int32_t rct_op, int32_t rct_perm) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !levels || !base || !out || n_planes < 1 || n_planes > 3 || n_levels < 1 || n_levels > 64 || base_w == 0 ||
      base_h == 0 || base_stride < base_w)
    return JXLH_ERR_INVALID_ARGUMENT;
  const bool with_rct = rct_op >= 0;
  if (with_rct && (n_planes != 3 || rct_op > 6 || rct_perm < 0 || rct_perm > 5)) return JXLH_ERR_INVALID_ARGUMENT;
  uint32_t cw = base_w, ch = base_h;
  for (int i = 0; i < n_levels; i++) {
    const jxlh_squeeze_level& lv = levels[i];
    if (lv.out_w == 0 || lv.out_h == 0 || lv.out_w > kMaxModularDim || lv.out_h > kMaxModularDim)
      return JXLH_ERR_INVALID_ARGUMENT;
    const uint32_t aw = lv.horizontal ? (lv.out_w + 1) / 2 : lv.out_w, ah = lv.horizontal ? lv.out_h : (lv.out_h + 1) / 2;
    if (aw != cw || ah != ch) return JXLH_ERR_INVALID_ARGUMENT;
    const uint32_t rw = lv.horizontal ? lv.out_w / 2 : lv.out_w, rh = lv.horizontal ? lv.out_h : lv.out_h / 2;
    for (int p = 0; p < n_planes; p++)
      if ((size_t)rw * rh > 0 && (!lv.res[p] || !is_device_ptr(lv.res[p]) || lv.res_stride < rw))
        return JXLH_ERR_INVALID_ARGUMENT;
    cw = lv.out_w;
    ch = lv.out_h;
  }
  if (out_stride < cw) return JXLH_ERR_INVALID_ARGUMENT;
  for (int p = 0; p < n_planes; p++)
    if (!base[p] || !out[p] || !is_device_ptr(base[p]) || !is_device_ptr(out[p])) return JXLH_ERR_INVALID_ARGUMENT;
  // intermediate planes: every level but the last writes its own plane set in context scratch (levels overlap in
  // the dataflow launch below, so no ping-pong; the sizes halve per level: about twice the largest one in all)
  jxlh_status st;
  // Rows of the intermediate planes are padded to whole 16 bytes and every plane starts on a 256-byte line: with the
  // caller's residual planes laid out the same way the tiled kernels move every level with 16-byte accesses whatever
  // the image width is (tiled_vec_ok).
  size_t level_off[64], level_stride[64], level_plane[64], arena = 0;
  for (int i = 0; i < n_levels - 1; i++) {
    level_off[i] = arena;
    level_stride[i] = ((size_t)levels[i].out_w + 3) & ~(size_t)3;
    level_plane[i] = (level_stride[i] * levels[i].out_h + 63) & ~(size_t)63;
    arena += level_plane[i] * n_planes;
  }
  if ((st = ensure(ctx, ctx->hook_i[0], std::max<size_t>(arena, 1)))) return st;
  ScopedKernelTimer t(ctx, "k6_unsqueeze_chain");
  const int32_t* cur[3];
  size_t cur_stride = base_stride;
  for (int p = 0; p < n_planes; p++) cur[p] = base[p];
  auto dst_of = [&](int i, int32_t* dst[3], size_t* stride) {
    const bool last = i == n_levels - 1;
    for (int p = 0; p < n_planes; p++)
      dst[p] = last ? out[p] : ctx->hook_i[0].p + level_off[i] + (size_t)p * level_plane[i];
    *stride = last ? out_stride : level_stride[i];
  };
  int i = 0;
  // ---- the first levels, while the planes fit LDS: one launch (the chain starts from <= 8 x 8)
  {
    int n_small = 0;
    while (n_small < n_levels - (with_rct ? 1 : 0) && n_small < JXLH_SQL_LEVELS && levels[n_small].out_w <= JXLH_SQL_MAX &&
           levels[n_small].out_h <= JXLH_SQL_MAX)
      n_small++;
    while (n_small >= 2) {
      int hz[JXLH_SQL_LEVELS];
      uint32_t ow[JXLH_SQL_LEVELS], oh[JXLH_SQL_LEVELS];
      size_t rs[JXLH_SQL_LEVELS];
      const int32_t* rp[JXLH_SQL_LEVELS * 3];
      for (int k = 0; k < n_small; k++) {
        hz[k] = levels[k].horizontal ? 1 : 0;
        ow[k] = levels[k].out_w;
        oh[k] = levels[k].out_h;
        rs[k] = levels[k].res_stride;
        for (int p = 0; p < 3; p++) rp[k * 3 + p] = p < n_planes && levels[k].res[p] ? levels[k].res[p] : base[0];
      }
      int32_t* dst[3];
      size_t dst_stride;
      dst_of(n_small - 1, dst, &dst_stride);
      if (launch_unsqueeze_levels(ctx->stream, n_planes, n_small, hz, ow, oh, rp, rs, base, base_stride, base_w, base_h, dst,
                                  dst_stride)) {
        for (int p = 0; p < n_planes; p++) cur[p] = dst[p];
        cur_stride = dst_stride;
        i = n_small;
        break;
      }
      n_small--;  // a level that does not fit the kernel's half-size buffer: try a shorter prefix
    }
  }
  // JXLH_SEPARATE_RCT=1 (tests): take the two-pass route that planes of 2^31 samples and more need;
  // JXLH_CHAIN_FLOW=0 (tests, A/B): one launch per streamed level instead of the dataflow launch
  const char* sep = getenv("JXLH_SEPARATE_RCT");
  const bool fuse_rct = with_rct && !(sep && *sep == '1');
  const char* fl = getenv("JXLH_CHAIN_FLOW");
  const bool flow = !(fl && *fl == '0');
  // ---- the remaining levels: runs of streamed levels as ONE dataflow launch (levels overlap: k6_unsqueeze_flow),
  // anything else one launch per level over the three planes; the last one fused with the RCT
  while (i < n_levels) {
    if (flow) {
      FlowStep steps[16];
      const int max_run = std::min(16, unsqueeze_flow_max_steps());
      const int32_t* a[3];
      size_t a_stride = cur_stride;
      for (int p = 0; p < n_planes; p++) a[p] = cur[p];
      int n = 0;
      for (int j = i; j < n_levels && n < max_run; j++, n++) {
        const jxlh_squeeze_level& lv = levels[j];
        if (j == n_levels - 1 && with_rct) break;  // the fused kernel (or, without fusion, the level + RCT pair below) takes it
        int32_t* dst[3];
        size_t dst_stride;
        dst_of(j, dst, &dst_stride);
        if (!unsqueeze_tiled_eligible(lv.horizontal ? 1 : 0, lv.out_w, lv.out_h, a_stride, lv.res_stride, dst_stride)) break;
        FlowStep& fs = steps[n];
        fs.horizontal = lv.horizontal ? 1 : 0;
        fs.avg_stride = a_stride;
        fs.res_stride = lv.res_stride;
        fs.out_w = lv.out_w;
        fs.out_h = lv.out_h;
        fs.out_stride = dst_stride;
        for (int p = 0; p < 3; p++) {
          const int q = p < n_planes ? p : 0;
          fs.avg[p] = a[q];
          fs.res[p] = lv.res[q] ? lv.res[q] : a[q];
          fs.out[p] = dst[q];
        }
        for (int p = 0; p < n_planes; p++) a[p] = dst[p];
        a_stride = dst_stride;
      }
      if (n >= 2) {
        if ((st = ensure(ctx, ctx->flow_words, unsqueeze_flow_words(n_planes, n, steps)))) return st;
        if (!ctx->host_flow_flag) {
          // the error word lives in pinned host memory (device-visible under the same address): jxlh_ctx_sync AND
          // jxlh_ctx_wait_mark read it without queueing a copy behind later work
          HIPCHK(ctx, hipHostMalloc(reinterpret_cast<void**>(&ctx->host_flow_flag), sizeof(int), hipHostMallocDefault));
          *ctx->host_flow_flag = 0;
        }
        if (ctx->flow_prof_on && (st = ensure(ctx, ctx->flow_prof, 11 * (size_t)unsqueeze_flow_max_steps()))) return st;
        launch_unsqueeze_flow(ctx->stream, n_planes, n, steps, ctx->flow_words.p, ctx->host_flow_flag, 4.0f,
                              ctx->flow_prof_on ? ctx->flow_prof.p : nullptr);
        ctx->flow_prof_levels = n;
        ctx->flow_used = true;
        for (int p = 0; p < n_planes; p++) cur[p] = a[p];
        cur_stride = a_stride;
        i += n;
        continue;
      }
    }
    const jxlh_squeeze_level& lv = levels[i];
    const bool last = i == n_levels - 1;
    int32_t* dst[3];
    size_t dst_stride;
    dst_of(i, dst, &dst_stride);
    const int32_t* rv[3];
    for (int p = 0; p < n_planes; p++) rv[p] = lv.res[p] ? lv.res[p] : cur[p];
    bool fused = false;
    if (last && fuse_rct)
      fused = launch_unsqueeze_rct(ctx->stream, lv.horizontal ? 1 : 0, cur, cur_stride, rv, lv.res_stride, lv.out_w, lv.out_h,
                                   dst, dst_stride, rct_op, rct_perm);
    if (!fused)
      launch_unsqueeze(ctx->stream, lv.horizontal ? 1 : 0, n_planes, cur, cur_stride, rv, lv.res_stride, lv.out_w, lv.out_h,
                       dst, dst_stride);
    if (last && with_rct && !fused) {
      if (dst_stride == lv.out_w) {
        launch_rct(ctx->stream, dst[0], dst[1], dst[2], (size_t)lv.out_w * lv.out_h, rct_op, rct_perm);
      } else {
        launch_rct_rows(ctx->stream, dst[0], dst[1], dst[2], lv.out_w, lv.out_h, dst_stride, rct_op, rct_perm);
      }
    }
    for (int p = 0; p < n_planes; p++) cur[p] = dst[p];
    cur_stride = dst_stride;
    i++;
  }
  HIPCHK(ctx, hipGetLastError());
  return JXLH_OK;
}

jxlh_status jxlh_unsqueeze_rct(jxlh_ctx* ctx, int32_t horizontal, const int32_t* const avg[3], size_t avg_stride,
                               const int32_t* const res[3], size_t res_stride, uint32_t out_w, uint32_t out_h,
                               int32_t* const out[3], size_t out_stride, int32_t op, int32_t perm) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !avg || !res || !out || out_stride < out_w || op < 0 || op > 6 || perm < 0 || perm > 5)
    return JXLH_ERR_INVALID_ARGUMENT;
  if (out_w == 0 || out_h == 0) return JXLH_OK;
  const uint32_t avg_w = horizontal ? (out_w + 1) / 2 : out_w;
  const uint32_t res_w = horizontal ? out_w / 2 : out_w, res_h = horizontal ? out_h : out_h / 2;
  const bool has_res = (size_t)res_w * res_h > 0;
  if (avg_stride < avg_w || (has_res && res_stride < res_w)) return JXLH_ERR_INVALID_ARGUMENT;
  const int32_t* rv[3];
  for (int i = 0; i < 3; i++) {
    if (!avg[i] || !out[i] || !is_device_ptr(avg[i]) || !is_device_ptr(out[i])) return JXLH_ERR_INVALID_ARGUMENT;
    if (has_res && (!res[i] || !is_device_ptr(res[i]))) return JXLH_ERR_INVALID_ARGUMENT;
    rv[i] = res[i] ? res[i] : avg[i];
  }
  const char* sep = getenv("JXLH_SEPARATE_RCT");
  if (!(sep && *sep == '1')) {
    ScopedKernelTimer t(ctx, horizontal ? "k6_unsqueeze_rct_h" : "k6_unsqueeze_rct_v");
    if (launch_unsqueeze_rct(ctx->stream, horizontal, avg, avg_stride, rv, res_stride, out_w, out_h, out, out_stride, op,
                             perm)) {
      HIPCHK(ctx, hipGetLastError());
      return JXLH_OK;
    }
  }
  // planes of 2^31 samples or more: the two separate passes (the RCT with a row pitch when the rows are padded)
  launch_unsqueeze(ctx->stream, horizontal, 3, avg, avg_stride, rv, res_stride, out_w, out_h, out, out_stride);
  if (out_stride == out_w) {
    launch_rct(ctx->stream, out[0], out[1], out[2], (size_t)out_w * out_h, op, perm);
  } else {
    launch_rct_rows(ctx->stream, out[0], out[1], out[2], out_w, out_h, out_stride, op, perm);
  }
  HIPCHK(ctx, hipGetLastError());
  return JXLH_OK;
}

jxlh_status jxlh_smooth_unsqueeze(jxlh_ctx* ctx, int32_t kind, const int32_t* avg, size_t avg_stride, uint32_t avg_w,
                                  uint32_t avg_h, uint32_t x0, uint32_t y0, int32_t* out, size_t out_stride,
                                  uint32_t out_w, uint32_t out_h) {
  JXLH_ON_DEVICE(ctx);
  // the float -> int conversion of the reference's build target rides in the kind argument
  const bool cvt_rne = kind >= 0 && (kind & JXLH_SMOOTH_CVT_NEAREST_EVEN) != 0;
  if (kind >= 0) kind &= ~JXLH_SMOOTH_CVT_NEAREST_EVEN;
  if (!ctx || !avg || !out || kind < JXLH_SMOOTH_H || kind > JXLH_SMOOTH_2D || avg_w == 0 || avg_h == 0 ||
      avg_stride < avg_w || out_stride < out_w || avg_w > (1u << 30) || avg_h > (1u << 30) || x0 > (1u << 30) ||
      y0 > (1u << 30) || out_w > (1u << 30) || out_h > (1u << 30))
    return JXLH_ERR_INVALID_ARGUMENT;
  const bool fx = kind != JXLH_SMOOTH_V, fy = kind != JXLH_SMOOTH_H;
  if ((fx ? out_w / 2 : out_w) == 0 || (fy ? out_h / 2 : out_h) == 0) return JXLH_OK; /* squeeze.rs:921-923 */
  static const char* const kNames[3] = {"k6_smooth_unsqueeze_h", "k6_smooth_unsqueeze_v", "k6_smooth_unsqueeze_2d"};
  if (is_device_ptr(avg) && is_device_ptr(out)) {
    ScopedKernelTimer t(ctx, kNames[kind]);
    launch_smooth_unsqueeze(ctx->stream, kind, avg, avg_stride, (int)avg_w, (int)avg_h, (int)x0, (int)y0, out,
                            out_stride, (int)out_w, (int)out_h, cvt_rne);
    HIPCHK(ctx, hipGetLastError());
    return JXLH_OK;
  }
  jxlh_status st;
  if ((st = stage_in(ctx, ctx->hook_i[0], avg, avg_stride * avg_h))) return st;
  if ((st = ensure(ctx, ctx->hook_i[2], out_stride * out_h))) return st;
  launch_smooth_unsqueeze(ctx->stream, kind, ctx->hook_i[0].p, avg_stride, (int)avg_w, (int)avg_h, (int)x0, (int)y0,
                          ctx->hook_i[2].p, out_stride, (int)out_w, (int)out_h, cvt_rne);
  HIPCHK(ctx, hipGetLastError());
  /* every sample of the rectangle is written; the stride padding of a host `out` is overwritten with whatever the
   * staging buffer held only if out_stride > out_w -- copy row by row instead */
  if (out_stride == out_w) return stage_out(ctx, out, (const int32_t*)ctx->hook_i[2].p, out_stride * out_h);
  HIPCHK(ctx, hipMemcpy2DAsync(out, out_stride * sizeof(int32_t), ctx->hook_i[2].p, out_stride * sizeof(int32_t),
                               out_w * sizeof(int32_t), out_h, hipMemcpyDeviceToHost, ctx->stream));
  JXLH_SYNC(ctx);
  return JXLH_OK;
}

jxlh_status jxlh_unsqueeze_planes(jxlh_ctx* ctx, int32_t horizontal, int32_t n_planes, const int32_t* const avg[],
                                  size_t avg_stride, const int32_t* const res[], size_t res_stride, uint32_t out_w,
                                  uint32_t out_h, int32_t* const out[], size_t out_stride) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !avg || !res || !out || n_planes < 1 || n_planes > 3 || out_stride < out_w)
    return JXLH_ERR_INVALID_ARGUMENT;
  if (out_w == 0 || out_h == 0) return JXLH_OK;
  if (out_w > kMaxModularDim || out_h > kMaxModularDim) return JXLH_ERR_UNSUPPORTED;
  const uint32_t avg_w = horizontal ? (out_w + 1) / 2 : out_w;
  const uint32_t res_w = horizontal ? out_w / 2 : out_w, res_h = horizontal ? out_h : out_h / 2;
  const bool has_res = (size_t)res_w * res_h > 0;
  if (avg_stride < avg_w || (has_res && res_stride < res_w)) return JXLH_ERR_INVALID_ARGUMENT;
  const int32_t* rv[3];
  for (int i = 0; i < n_planes; i++) {
    if (!avg[i] || !out[i] || !is_device_ptr(avg[i]) || !is_device_ptr(out[i])) return JXLH_ERR_INVALID_ARGUMENT;
    rv[i] = res[i] ? res[i] : avg[i];
    if (has_res && (!res[i] || !is_device_ptr(res[i]))) return JXLH_ERR_INVALID_ARGUMENT;
  }
  ScopedKernelTimer t(ctx, horizontal ? "k6_unsqueeze_h" : "k6_unsqueeze_v");
  launch_unsqueeze(ctx->stream, horizontal, n_planes, avg, avg_stride, rv, res_stride, out_w, out_h, out, out_stride);
  HIPCHK(ctx, hipGetLastError());
  return JXLH_OK;
}

}  // extern "C"
