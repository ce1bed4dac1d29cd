// C ABI, reading a finished frame: f32 planes, the LF image, and the output stages after EPF (XYB / YCbCr -> RGB,
// every transfer function, 8 / 16 bit interleaved), SURVEY.md 8(f) item 2.
#include <algorithm>

#include "jxlh_ctx.h"

extern "C" {

namespace {
const XybParamsDev* xyb_params_dev(const jxlh_xyb_params* p, XybParamsDev* d) {
  if (!p) return nullptr;
  for (int i = 0; i < 9; i++) d->mat[i] = p->opsin_inverse_matrix[i];
  for (int i = 0; i < 3; i++) {
    d->bias_cbrt[i] = p->bias_cbrt[i];
    d->scaled_bias[i] = p->scaled_bias[i];
  }
  d->intensity_scale = p->intensity_scale;
  return d;
}

SubPlanesDev sub_planes_dev(const jxlh_ctx* ctx) {
  const FrameDev& f = ctx->fd;
  SubPlanesDev sp;
  for (int c = 0; c < 3; c++) {
    const int hs = f.hshift[c], vs = f.vshift[c];
    sp.p[c] = (hs | vs) ? f.tmp[c] : f.planes[c];
    sp.hs[c] = hs;
    sp.vs[c] = vs;
    sp.cw[c] = (f.xsize + (1 << hs) - 1) >> hs;
    sp.ch[c] = (f.ysize + (1 << vs) - 1) >> vs;
  }
  return sp;
}

// mode: kTfLinear..kTfGamma = XybStage (p) + that transfer function (t); kModeYcbcr; kModeNone
jxlh_status read_rgb8(jxlh_ctx* ctx, int mode, const jxlh_xyb_params* p, const TfParamsDev& tf, uint32_t channels,
                      uint32_t y0, uint32_t y1, void* out, size_t bytes_per_row, bool wait = true) {
  if (!ctx || !out || (channels != 3 && channels != 4)) return JXLH_ERR_INVALID_ARGUMENT;
  if (!ctx->in_frame || !ctx->result[0]) return JXLH_ERR_BAD_STATE;
  if (y1 > (uint32_t)ctx->res_h) y1 = (uint32_t)ctx->res_h;
  if (y0 >= y1 || bytes_per_row < (size_t)ctx->res_w * channels) return JXLH_ERR_INVALID_ARGUMENT;
  XybParamsDev d = {};
  xyb_params_dev(p, &d);
  const int rows = (int)(y1 - y0);
  const bool fused_chroma = ctx->chroma_lazy && mode == kModeYcbcr;
  if (!fused_chroma) materialise_chroma(ctx);
  const float* planes[3] = {ctx->result[0], ctx->result[1], ctx->result[2]};
  if (fused_chroma) {
    const SubPlanesDev sp = sub_planes_dev(ctx);
    const size_t tight = ((size_t)ctx->res_w * channels + 3) & ~(size_t)3;
    const bool dev = is_device_ptr(out);
    if (!dev)
      if (jxlh_status st = ensure(ctx, ctx->rgb8, tight * (size_t)rows)) return st;
    {
      ScopedKernelTimer t(ctx, "k_ycbcr_sub_to_rgb");
      launch_ycbcr_sub_to_rgb(ctx->stream, sp, ctx->res_stride, ctx->res_w, (int)y0, rows, (int)channels, 8,
                              dev ? out : (void*)ctx->rgb8.p, dev ? bytes_per_row : tight);
    }
    HIPCHK(ctx, hipGetLastError());
    if (dev) return JXLH_OK;
    if (jxlh_status st = copy2d(ctx, out, bytes_per_row, ctx->rgb8.p, tight, (size_t)ctx->res_w * channels, (size_t)rows,
                                ctx->stream))
      return st;
    if (wait) JXLH_SYNC(ctx);
    return JXLH_OK;
  }
  if (is_device_ptr(out)) {
    ScopedKernelTimer t(ctx, "k_xyb_to_rgb8");
    launch_xyb_to_rgb8(ctx->stream, planes, ctx->res_stride, ctx->res_w, (int)y0, rows, mode, d, tf, (int)channels,
                       static_cast<uint8_t*>(out), bytes_per_row);
    HIPCHK(ctx, hipGetLastError());
    return JXLH_OK;
  }
  // staging rows are dword aligned; when the caller's rows are tight and already aligned the D2H
  // is one linear copy, otherwise a 2-D copy of exactly the pixel bytes (row padding is never written)
  const size_t tight = ((size_t)ctx->res_w * channels + 3) & ~(size_t)3;
  if (jxlh_status st = ensure(ctx, ctx->rgb8, tight * (size_t)rows)) return st;
  {
    ScopedKernelTimer t(ctx, "k_xyb_to_rgb8");
    launch_xyb_to_rgb8(ctx->stream, planes, ctx->res_stride, ctx->res_w, (int)y0, rows, mode, d, tf, (int)channels, ctx->rgb8.p,
                       tight);
  }
  HIPCHK(ctx, hipGetLastError());
  const size_t row_bytes = (size_t)ctx->res_w * channels;
  if (jxlh_status st = copy2d(ctx, out, bytes_per_row, ctx->rgb8.p, tight, row_bytes, (size_t)rows, ctx->stream))
    return st;
  if (wait) JXLH_SYNC(ctx);
  return JXLH_OK;
}

jxlh_status read_rgb16(jxlh_ctx* ctx, int mode, const jxlh_xyb_params* p, const TfParamsDev& tf, uint32_t channels,
                       uint32_t y0, uint32_t y1, void* out, size_t bytes_per_row, bool wait = true) {
  if (!ctx || !out || (channels != 3 && channels != 4)) return JXLH_ERR_INVALID_ARGUMENT;
  if (!ctx->in_frame || !ctx->result[0]) return JXLH_ERR_BAD_STATE;
  if (y1 > (uint32_t)ctx->res_h) y1 = (uint32_t)ctx->res_h;
  const size_t row_bytes = (size_t)ctx->res_w * channels * sizeof(uint16_t);
  if (y0 >= y1 || bytes_per_row < row_bytes || bytes_per_row % sizeof(uint16_t) != 0 ||
      reinterpret_cast<uintptr_t>(out) % sizeof(uint16_t) != 0)
    return JXLH_ERR_INVALID_ARGUMENT;
  XybParamsDev d = {};
  xyb_params_dev(p, &d);
  const int rows = (int)(y1 - y0);
  const bool fused_chroma = ctx->chroma_lazy && mode == kModeYcbcr;
  if (!fused_chroma) materialise_chroma(ctx);
  const float* planes[3] = {ctx->result[0], ctx->result[1], ctx->result[2]};
  if (fused_chroma) {
    const SubPlanesDev sp = sub_planes_dev(ctx);
    const bool dev = is_device_ptr(out);
    if (!dev)
      if (jxlh_status st = ensure(ctx, ctx->rgb8, row_bytes * (size_t)rows)) return st;
    {
      ScopedKernelTimer t(ctx, "k_ycbcr_sub_to_rgb");
      launch_ycbcr_sub_to_rgb(ctx->stream, sp, ctx->res_stride, ctx->res_w, (int)y0, rows, (int)channels, 16,
                              dev ? out : (void*)ctx->rgb8.p,
                              dev ? bytes_per_row / sizeof(uint16_t) : (size_t)ctx->res_w * channels);
    }
    HIPCHK(ctx, hipGetLastError());
    if (dev) return JXLH_OK;
    if (jxlh_status st = copy2d(ctx, out, bytes_per_row, ctx->rgb8.p, row_bytes, row_bytes, (size_t)rows, ctx->stream))
      return st;
    if (wait) JXLH_SYNC(ctx);
    return JXLH_OK;
  }
  if (is_device_ptr(out)) {
    ScopedKernelTimer t(ctx, "k_xyb_to_rgb16");
    launch_xyb_to_rgb16(ctx->stream, planes, ctx->res_stride, ctx->res_w, (int)y0, rows, mode, d, tf, (int)channels,
                        static_cast<uint16_t*>(out), bytes_per_row / sizeof(uint16_t));
    HIPCHK(ctx, hipGetLastError());
    return JXLH_OK;
  }
  if (jxlh_status st = ensure(ctx, ctx->rgb8, row_bytes * (size_t)rows)) return st;
  {
    ScopedKernelTimer t(ctx, "k_xyb_to_rgb16");
    launch_xyb_to_rgb16(ctx->stream, planes, ctx->res_stride, ctx->res_w, (int)y0, rows, mode, d, tf, (int)channels,
                        reinterpret_cast<uint16_t*>(ctx->rgb8.p), (size_t)ctx->res_w * channels);
  }
  HIPCHK(ctx, hipGetLastError());
  if (jxlh_status st = copy2d(ctx, out, bytes_per_row, ctx->rgb8.p, row_bytes, row_bytes, (size_t)rows, ctx->stream))
    return st;
  if (wait) JXLH_SYNC(ctx);
  return JXLH_OK;
}
}  // namespace

jxlh_status jxlh_frame_read_rgb8(jxlh_ctx* ctx, const jxlh_xyb_params* p, uint32_t channels, uint32_t y0,
                                 uint32_t y1, void* out, size_t bytes_per_row) {
  JXLH_ON_DEVICE(ctx);
  if (!p) return JXLH_ERR_INVALID_ARGUMENT;
  return read_rgb8(ctx, kTfSrgb, p, TfParamsDev{}, channels, y0, y1, out, bytes_per_row);
}
jxlh_status jxlh_frame_read_rgb8_async(jxlh_ctx* ctx, const jxlh_xyb_params* p, uint32_t channels, uint32_t y0,
                                       uint32_t y1, void* out, size_t bytes_per_row) {
  JXLH_ON_DEVICE(ctx);
  if (!p) return JXLH_ERR_INVALID_ARGUMENT;
  return read_rgb8(ctx, kTfSrgb, p, TfParamsDev{}, channels, y0, y1, out, bytes_per_row, /*wait=*/false);
}
jxlh_status jxlh_frame_read_rgb16(jxlh_ctx* ctx, const jxlh_xyb_params* p, uint32_t channels, uint32_t y0,
                                  uint32_t y1, void* out, size_t bytes_per_row) {
  JXLH_ON_DEVICE(ctx);
  if (!p) return JXLH_ERR_INVALID_ARGUMENT;
  return read_rgb16(ctx, kTfSrgb, p, TfParamsDev{}, channels, y0, y1, out, bytes_per_row);
}
jxlh_status jxlh_frame_read_ycbcr_rgb8(jxlh_ctx* ctx, uint32_t channels, uint32_t y0, uint32_t y1, void* out,
                                       size_t bytes_per_row) {
  JXLH_ON_DEVICE(ctx);
  return read_rgb8(ctx, kModeYcbcr, nullptr, TfParamsDev{}, channels, y0, y1, out, bytes_per_row);
}
jxlh_status jxlh_frame_read_ycbcr_rgb16(jxlh_ctx* ctx, uint32_t channels, uint32_t y0, uint32_t y1, void* out,
                                        size_t bytes_per_row) {
  JXLH_ON_DEVICE(ctx);
  return read_rgb16(ctx, kModeYcbcr, nullptr, TfParamsDev{}, channels, y0, y1, out, bytes_per_row);
}

namespace {
jxlh_status read_output(jxlh_ctx* ctx, const jxlh_output_desc* d, uint32_t y0, uint32_t y1, void* out,
                        size_t bytes_per_row, bool wait) {
  if (!ctx || !d || (d->bits != 8 && d->bits != 16)) return JXLH_ERR_INVALID_ARGUMENT;
  int mode;
  switch (d->color) {
    case JXLH_COLOR_XYB:
      if (d->transfer > JXLH_TF_GAMMA) return JXLH_ERR_INVALID_ARGUMENT;
      mode = (int)d->transfer;  // JXLH_TF_* share the values of the internal modes
      break;
    case JXLH_COLOR_YCBCR: mode = kModeYcbcr; break;
    case JXLH_COLOR_NONE: mode = kModeNone; break;
    default: return JXLH_ERR_INVALID_ARGUMENT;
  }
  TfParamsDev t;
  t.param = d->tf_param;
  for (int i = 0; i < 3; i++) t.lum[i] = d->hlg_luminance_rgb[i];
  const jxlh_xyb_params* xp = d->color == JXLH_COLOR_XYB ? &d->xyb : nullptr;
  return d->bits == 8 ? read_rgb8(ctx, mode, xp, t, d->channels, y0, y1, out, bytes_per_row, wait)
                      : read_rgb16(ctx, mode, xp, t, d->channels, y0, y1, out, bytes_per_row, wait);
}
}  // namespace

}  // extern "C"
// comm.hip: a rank's band of the converted image into its rows of `out` (device memory), queued on the context's stream
jxlh_status jxlh_host::convert_band_to_output(jxlh_ctx* ctx, const jxlh_output_desc* d, uint32_t y0, uint32_t y1, void* out,
                                   size_t bytes_per_row) {
  if (y0 >= y1) return JXLH_OK;
  if (!is_device_ptr(out)) return JXLH_ERR_INVALID_ARGUMENT;
  return read_output(ctx, d, y0, y1, static_cast<char*>(out) + (size_t)y0 * bytes_per_row, bytes_per_row, /*wait=*/false);
}

extern "C" {
jxlh_status jxlh_frame_read_output(jxlh_ctx* ctx, const jxlh_output_desc* d, uint32_t y0, uint32_t y1, void* out,
                                   size_t bytes_per_row) {
  JXLH_ON_DEVICE(ctx);
  return read_output(ctx, d, y0, y1, out, bytes_per_row, /*wait=*/true);
}
jxlh_status jxlh_frame_read_output_async(jxlh_ctx* ctx, const jxlh_output_desc* d, uint32_t y0, uint32_t y1, void* out,
                                         size_t bytes_per_row) {
  JXLH_ON_DEVICE(ctx);
  return read_output(ctx, d, y0, y1, out, bytes_per_row, /*wait=*/false);
}

jxlh_status jxlh_frame_read_planes(jxlh_ctx* ctx, const jxlh_plane out[3]) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !out) return JXLH_ERR_INVALID_ARGUMENT;
  if (!ctx->in_frame || !ctx->result[0]) return JXLH_ERR_BAD_STATE;
  materialise_chroma(ctx);
  for (int c = 0; c < 3; c++) {
    if (!out[c].ptr || out[c].bytes_per_row < (size_t)ctx->res_w * sizeof(float) || out[c].num_rows < (size_t)ctx->res_h ||
        out[c].bytes_between_rows < out[c].bytes_per_row)
      return JXLH_ERR_INVALID_ARGUMENT;
    jxlh_status st = copy2d(ctx, out[c].ptr, out[c].bytes_between_rows, ctx->result[c],
                            ctx->res_stride * sizeof(float), (size_t)ctx->res_w * sizeof(float), ctx->res_h, ctx->stream);
    if (st != JXLH_OK) return st;
  }
  return jxlh_ctx_sync(ctx);
}

namespace {
// the rect [x0, x0 + w) x [y0, y0 + h) of the finished planes, cut at the result's right / bottom edge
jxlh_status read_planes_rect(jxlh_ctx* ctx, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, const jxlh_plane out[3],
                             bool wait) {
  if (!ctx || !out || w == 0 || h == 0) return JXLH_ERR_INVALID_ARGUMENT;
  if (!ctx->in_frame || !ctx->result[0]) return JXLH_ERR_BAD_STATE;
  if (x0 >= (uint32_t)ctx->res_w || y0 >= (uint32_t)ctx->res_h) return JXLH_ERR_INVALID_ARGUMENT;
  const size_t cw = std::min<size_t>(w, (size_t)ctx->res_w - x0), ch = std::min<size_t>(h, (size_t)ctx->res_h - y0);
  for (int c = 0; c < 3; c++)
    if (!out[c].ptr || out[c].bytes_per_row < cw * sizeof(float) || out[c].num_rows < ch ||
        out[c].bytes_between_rows < out[c].bytes_per_row)
      return JXLH_ERR_INVALID_ARGUMENT;
  materialise_chroma(ctx);
  for (int c = 0; c < 3; c++) {
    const float* src = ctx->result[c] + (size_t)y0 * ctx->res_stride + x0;
    if (jxlh_status st = copy2d(ctx, out[c].ptr, out[c].bytes_between_rows, src, ctx->res_stride * sizeof(float),
                                cw * sizeof(float), ch, ctx->stream))
      return st;
  }
  return wait ? jxlh_ctx_sync(ctx) : JXLH_OK;
}
}  // namespace

jxlh_status jxlh_frame_read_planes_rect(jxlh_ctx* ctx, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h,
                                        const jxlh_plane out[3]) {
  JXLH_ON_DEVICE(ctx);
  return read_planes_rect(ctx, x0, y0, w, h, out, /*wait=*/true);
}
jxlh_status jxlh_frame_read_planes_rect_async(jxlh_ctx* ctx, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h,
                                              const jxlh_plane out[3]) {
  JXLH_ON_DEVICE(ctx);
  return read_planes_rect(ctx, x0, y0, w, h, out, /*wait=*/false);
}

jxlh_status jxlh_frame_device_planes(jxlh_ctx* ctx, float* planes[3], size_t* stride) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !planes) return JXLH_ERR_INVALID_ARGUMENT;
  if (!ctx->in_frame || !ctx->result[0]) return JXLH_ERR_BAD_STATE;
  materialise_chroma(ctx);
  for (int c = 0; c < 3; c++) planes[c] = ctx->result[c];
  if (stride) *stride = ctx->res_stride;
  return JXLH_OK;
}

jxlh_status jxlh_frame_read_lf(jxlh_ctx* ctx, float* x, float* y, float* b, size_t stride) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !x || !y || !b) return JXLH_ERR_INVALID_ARGUMENT;
  if (!ctx->in_frame) return JXLH_ERR_BAD_STATE;
  const FrameDev& f = ctx->fd;
  if (stride < (size_t)f.xblocks) return JXLH_ERR_INVALID_ARGUMENT;
  float* dst[3] = {x, y, b};
  for (int c = 0; c < 3; c++) {
    jxlh_status st = copy2d(ctx, dst[c], stride * sizeof(float), f.lf[c], f.xblocks * sizeof(float),
                            f.xblocks * sizeof(float), f.yblocks, ctx->stream);
    if (st != JXLH_OK) return st;
  }
  JXLH_SYNC(ctx);
  return JXLH_OK;
}

}  // extern "C"
