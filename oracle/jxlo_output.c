/* CPU oracle (TEST INFRASTRUCTURE, see jxlo.h): the output stages after EPF for an XYB-encoded
 * frame shown as 8-bit sRGB --
 *   XybStage            jxl/src/render/stages/xyb.rs:146-164 (params), :208-240 (per pixel)
 *   FromLinearStage     jxl/src/render/stages/from_linear.rs:73-78 -> jxl/src/color/tf.rs:13-44,
 *                       eval_rational_poly_simd jxl/src/util/rational_poly.rs:20-35
 *   ConvertF32ToU8Stage jxl/src/render/stages/convert.rs:570-606 (scale, dither, clamp, round)
 * in the order frame/render.rs:757-762, :118 applies them.  JXLO_FUSED=1 models the AVX2 back-end
 * (mul_add = FMA, round-to-nearest-even store), JXLO_FUSED=0 the scalar one (a*b+c, f32::round). */
#include <math.h>
#include <string.h>

#include "jxlo.h"

#ifndef JXLO_FUSED
#define JXLO_FUSED 1
#endif

static inline float mul_add(float a, float b, float c) {
#if JXLO_FUSED
  return fmaf(a, b, c);
#else
  return (a * b) + c;
#endif
}

static const float kDither[32 * 32] = {
#include "dither_table.inc"
};

/* XybParams::new (xyb.rs:154-163); cbrtf stands in for Rust's f32::cbrt (the device ABI takes the
 * finished parameters, so the caller's own cbrt is what counts there) */
void jxlo_xyb_params(const float inverse_matrix[9], const float opsin_biases[3], float intensity_target,
                     JxloXybParams* out) {
  const float intensity_scale = 255.0f / intensity_target;
  for (int i = 0; i < 9; i++) out->mat[i] = inverse_matrix[i];
  for (int i = 0; i < 3; i++) {
    out->bias_cbrt[i] = cbrtf(opsin_biases[i]);
    out->scaled_bias[i] = opsin_biases[i] * intensity_scale;
  }
  out->intensity_scale = intensity_scale;
}

/* xyb_process (xyb.rs:208-240), in place on three rows */
void jxlo_xyb_to_linear(const JxloXybParams* p, float* row_x, float* row_y, float* row_b, size_t n) {
  for (size_t i = 0; i < n; i++) {
    const float x = row_x[i], y = row_y[i], b = row_b[i];
    float l = y + x - p->bias_cbrt[0];
    float m = y - x - p->bias_cbrt[1];
    float s = b - p->bias_cbrt[2];
    const float l2 = l * l, m2 = m * m, s2 = s * s;
    const float scaled_l = l * p->intensity_scale, scaled_m = m * p->intensity_scale,
                scaled_s = s * p->intensity_scale;
    l = mul_add(l2, scaled_l, p->scaled_bias[0]);
    m = mul_add(m2, scaled_m, p->scaled_bias[1]);
    s = mul_add(s2, scaled_s, p->scaled_bias[2]);
    row_x[i] = mul_add(p->mat[0], l, mul_add(p->mat[1], m, p->mat[2] * s));
    row_y[i] = mul_add(p->mat[3], l, mul_add(p->mat[4], m, p->mat[5] * s));
    row_b[i] = mul_add(p->mat[6], l, mul_add(p->mat[7], m, p->mat[8] * s));
  }
}

/* linear_to_srgb_simd (tf.rs:13-44): |x| < 0.0031308 ? 12.92|x| : P(sqrt|x|)/Q(sqrt|x|), sign restored */
float jxlo_linear_to_srgb1(float x) {
  static const float P[5] = {-5.135152395e-4f, 5.287254571e-3f, 3.903842876e-1f, 1.474205315f, 7.352629620e-1f};
  static const float Q[5] = {1.004519624e-2f, 3.036675394e-1f, 1.340816930f, 9.258482155e-1f, 2.424867759e-2f};
  const float a = fabsf(x);
  float r;
  if (0.0031308f > a) {
    r = a * 12.92f;
  } else {
    const float t = sqrtf(a);
    float yp = P[4], yq = Q[4];
    for (int i = 3; i >= 0; i--) yp = mul_add(yp, t, P[i]);
    for (int i = 3; i >= 0; i--) yq = mul_add(yq, t, Q[i]);
    r = yp / yq;
  }
  return copysignf(r, x);
}
void jxlo_linear_to_srgb(float* v, size_t n) {
  for (size_t i = 0; i < n; i++) v[i] = jxlo_linear_to_srgb1(v[i]);
}

/* f32_to_u8_simd (convert.rs:574-606) for one sample of `channel` at frame position (x, y); the
 * reference's padded table rows make the lookup (x + 23*channel) mod 32 for every SIMD width */
uint8_t jxlo_f32_to_u8(float v, size_t x, size_t y, int channel, int bit_depth) {
  const float max = (float)((1u << bit_depth) - 1u);
  const float dither = kDither[((y + (size_t)channel * 13) % 32) * 32 + (x + (size_t)channel * 23) % 32];
  const float scaled = v * max;
  const float dithered = scaled + dither;
  float clamped = dithered > 0.0f ? dithered : 0.0f; /* max(zero): NaN -> 0 like _mm256_max_ps(v, zero) */
  clamped = clamped < max ? clamped : max;
#if JXLO_FUSED
  return (uint8_t)rintf(clamped); /* _mm256_round_ps(TO_NEAREST_INT): ties to even (avx.rs:609) */
#else
  return (uint8_t)roundf(clamped); /* f32::round: ties away from zero (scalar.rs:199-201) */
#endif
}

/* f32_to_u16_simd (convert.rs:743-761): clamp to [0, 1], scale, round; no dither */
uint16_t jxlo_f32_to_u16(float v, int bit_depth) {
  const float max = (float)((1u << bit_depth) - 1u);
  float clamped = v > 0.0f ? v : 0.0f;
  clamped = clamped < 1.0f ? clamped : 1.0f;
  const float scaled = clamped * max;
#if JXLO_FUSED
  return (uint16_t)rintf(scaled); /* avx.rs:642 */
#else
  return (uint16_t)roundf(scaled); /* scalar.rs:204-206 */
#endif
}

void jxlo_xyb_to_rgb16(const JxloXybParams* p, const float* px, const float* py, const float* pb, size_t w, size_t h,
                       size_t stride, uint16_t* out, size_t out_stride_elems, int out_channels) {
  for (size_t y = 0; y < h; y++) {
    for (size_t x = 0; x < w; x++) {
      float r = px[y * stride + x], g = py[y * stride + x], b = pb[y * stride + x];
      jxlo_xyb_to_linear(p, &r, &g, &b, 1);
      uint16_t* o = out + y * out_stride_elems + x * (size_t)out_channels;
      o[0] = jxlo_f32_to_u16(jxlo_linear_to_srgb1(r), 16);
      o[1] = jxlo_f32_to_u16(jxlo_linear_to_srgb1(g), 16);
      o[2] = jxlo_f32_to_u16(jxlo_linear_to_srgb1(b), 16);
      if (out_channels == 4) o[3] = 65535;
    }
  }
}

/* the three stages on whole planes -> interleaved 8-bit (out_channels = 3: RGB, 4: RGBA with A = 255) */
void jxlo_xyb_to_rgb8(const JxloXybParams* p, const float* px, const float* py, const float* pb, size_t w, size_t h,
                      size_t stride, uint8_t* out, size_t out_stride_bytes, int out_channels) {
  for (size_t y = 0; y < h; y++) {
    for (size_t x = 0; x < w; x++) {
      float r = px[y * stride + x], g = py[y * stride + x], b = pb[y * stride + x];
      jxlo_xyb_to_linear(p, &r, &g, &b, 1);
      const float v[3] = {jxlo_linear_to_srgb1(r), jxlo_linear_to_srgb1(g), jxlo_linear_to_srgb1(b)};
      uint8_t* o = out + y * out_stride_bytes + x * (size_t)out_channels;
      for (int c = 0; c < 3; c++) o[c] = jxlo_f32_to_u8(v[c], x, y, c, 8);
      if (out_channels == 4) o[3] = 255;
    }
  }
}

/* ---- YCbCr frames (JPEG recompression): render/stages/ycbcr.rs:35-78 ---- */
void jxlo_ycbcr_to_rgb(float* cb, float* y, float* cr, size_t n) {
  /* the constants are f32 literals of the products the source spells out */
  const float c128 = 128.0f / 255.0f;
  const float cr_to_r = 1.402f;
  const float cr_to_g = -0.299f * 1.402f / 0.587f;
  const float cb_to_g = -0.114f * 1.772f / 0.587f;
  const float cb_to_b = 1.772f;
  for (size_t i = 0; i < n; i++) {
    const float yv = y[i] + c128, cbv = cb[i], crv = cr[i];
    const float r = mul_add(crv, cr_to_r, yv);
    const float g = mul_add(crv, cr_to_g, mul_add(cbv, cb_to_g, yv));
    const float b = mul_add(cbv, cb_to_b, yv);
    cb[i] = r; /* R -> Cb plane, G -> Y plane, B -> Cr plane (:74-77) */
    y[i] = g;
    cr[i] = b;
  }
}

void jxlo_ycbcr_to_rgb8(const float* pcb, const float* py, const float* pcr, size_t w, size_t h, size_t stride,
                        uint8_t* out, size_t out_stride_bytes, int out_channels) {
  for (size_t y = 0; y < h; y++) {
    for (size_t x = 0; x < w; x++) {
      float v[3] = {pcb[y * stride + x], py[y * stride + x], pcr[y * stride + x]};
      jxlo_ycbcr_to_rgb(&v[0], &v[1], &v[2], 1);
      uint8_t* o = out + y * out_stride_bytes + x * (size_t)out_channels;
      for (int c = 0; c < 3; c++) o[c] = jxlo_f32_to_u8(v[c], x, y, c, 8);
      if (out_channels == 4) o[3] = 255;
    }
  }
}

void jxlo_ycbcr_to_rgb16(const float* pcb, const float* py, const float* pcr, size_t w, size_t h, size_t stride,
                         uint16_t* out, size_t out_stride_elems, int out_channels) {
  for (size_t y = 0; y < h; y++) {
    for (size_t x = 0; x < w; x++) {
      float v[3] = {pcb[y * stride + x], py[y * stride + x], pcr[y * stride + x]};
      jxlo_ycbcr_to_rgb(&v[0], &v[1], &v[2], 1);
      uint16_t* o = out + y * out_stride_elems + x * (size_t)out_channels;
      for (int c = 0; c < 3; c++) o[c] = jxlo_f32_to_u16(v[c], 16);
      if (out_channels == 4) o[3] = 65535;
    }
  }
}
