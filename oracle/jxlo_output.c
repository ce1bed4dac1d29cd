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

/* ---- FromLinearStage, the other transfer functions (render/stages/from_linear.rs:57-112) ----
 * The SIMD-dispatched curves (BT.709, PQ, gamma) use mul_add (= FMA on the FMA back-ends); HLG runs through the
 * scalar helpers: plain mul/add Horner steps (util/rational_poly.rs:13-17, fast_math.rs:80-137) and a true fused
 * f32::mul_add in the luminance mix (color/tf.rs:385). */
#include "tf_constants.inc"

static inline float ratpoly_simd(float x, const float* p, int np, const float* q, int nq) { /* rational_poly.rs:20-35 */
  float yp = p[np - 1], yq = q[nq - 1];
  for (int i = np - 2; i >= 0; i--) yp = mul_add(yp, x, p[i]);
  for (int i = nq - 2; i >= 0; i--) yq = mul_add(yq, x, q[i]);
  return yp / yq;
}
static inline float ratpoly_scalar(float x, const float* p, int np, const float* q, int nq) { /* :13-17 */
  float yp = p[np - 1], yq = q[nq - 1];
  for (int i = np - 2; i >= 0; i--) yp = yp * x + p[i];
  for (int i = nq - 2; i >= 0; i--) yq = yq * x + q[i];
  return yp / yq;
}
static inline float bits_f(int32_t b) { float f; memcpy(&f, &b, 4); return f; }
static inline int32_t f_bits(float f) { int32_t b; memcpy(&b, &f, 4); return b; }

static inline float fast_log2f_any(float x, int simd) { /* fast_math.rs:127-149 */
  const int32_t x_bits = f_bits(x);
  const int32_t exp_bits = (int32_t)((uint32_t)x_bits - 0x3f2aaaabu);
  const int32_t exp_shifted = exp_bits >> 23;
  const float mantissa = bits_f((int32_t)((uint32_t)x_bits - ((uint32_t)exp_shifted << 23)));
  const float exp_val = (float)exp_shifted;
  const float m1 = mantissa - 1.0f;
  return (simd ? ratpoly_simd(m1, kTf_LOG2F_P, 3, kTf_LOG2F_Q, 3) : ratpoly_scalar(m1, kTf_LOG2F_P, 3, kTf_LOG2F_Q, 3)) + exp_val;
}
static inline float fast_pow2f_any(float x, int simd) { /* fast_math.rs:79-114 */
  const float x_floor = floorf(x);
  const float e = bits_f((int32_t)(((uint32_t)((int32_t)x_floor + 127)) << 23));
  const float frac = x - x_floor;
  float num = frac + kTf_POW2F_NUMER[0];
  float den;
  if (simd) {
    num = mul_add(num, frac, kTf_POW2F_NUMER[1]);
    num = mul_add(num, frac, kTf_POW2F_NUMER[2]);
    num = num * e;
    den = mul_add(kTf_POW2F_DENOM[0], frac, kTf_POW2F_DENOM[1]);
    den = mul_add(den, frac, kTf_POW2F_DENOM[2]);
    den = mul_add(den, frac, kTf_POW2F_DENOM[3]);
  } else {
    num = num * frac + kTf_POW2F_NUMER[1];
    num = num * frac + kTf_POW2F_NUMER[2];
    num = num * e;
    den = kTf_POW2F_DENOM[0] * frac + kTf_POW2F_DENOM[1];
    den = den * frac + kTf_POW2F_DENOM[2];
    den = den * frac + kTf_POW2F_DENOM[3];
  }
  return num / den;
}
float jxlo_fast_powf(float base, float e, int simd) { return fast_pow2f_any(fast_log2f_any(base, simd) * e, simd); }

float jxlo_linear_to_bt709_1(float x) { /* color/tf.rs:115-148 */
  const float a = fabsf(x);
  const float r = (0.018f > a) ? a * 4.5f : ratpoly_simd(sqrtf(a), kTf_BT709_P, 5, kTf_BT709_Q, 5);
  return copysignf(r, x);
}
float jxlo_linear_to_pq_1(float intensity_target, float x) { /* color/tf.rs:288-314 */
  const float y_mult = intensity_target * (1.0f / 10000.0f);
  const float a = fabsf(x);
  const float a_1_4 = sqrtf(sqrtf(a * y_mult));
  const float y_small = ratpoly_simd(a_1_4, kTf_PQ_INV_EOTF_P_SMALL, 5, kTf_PQ_INV_EOTF_Q_SMALL, 5);
  const float y_large = ratpoly_simd(a_1_4, kTf_PQ_INV_EOTF_P, 5, kTf_PQ_INV_EOTF_Q, 5);
  return copysignf((1e-4f > a) ? y_small : y_large, x);
}
float jxlo_linear_to_gamma_1(float g, float x) { /* from_linear.rs:99-110 */
  return copysignf(jxlo_fast_powf(fabsf(x), g, 1), x);
}
/* hlg_display_to_scene + scene_to_hlg (color/tf.rs:379-393, :437-447, :482-497); exponent =
 * (1 - system_gamma) / system_gamma, the host-side scalar tf.rs:442-446 computes from intensity_target */
void jxlo_linear_to_hlg_1(float exponent, const float lum[3], float* r, float* g, float* b) {
  if (!(fabsf(exponent) < 0.1f)) {
    const float mixed = fmaf(*r, lum[0], fmaf(*g, lum[1], *b * lum[2])); /* std f32::mul_add: always fused */
    const float mult = jxlo_fast_powf(mixed, exponent, 0);
    *r *= mult;
    *g *= mult;
    *b *= mult;
  }
  const double HLG_A = 0.17883277, HLG_B = 1.0 - 4.0 * HLG_A, HLG_C = 0.5599107295;
  const float k = (float)(HLG_A * 0.693147180559945309417232121458176568), hb = (float)HLG_B, hc = (float)HLG_C;
  float* v[3] = {r, g, b};
  for (int i = 0; i < 3; i++) {
    const float a = fabsf(*v[i]);
    const float y = (a <= 1.0f / 12.0f) ? sqrtf(3.0f * a) : k * fast_log2f_any(12.0f * a - hb, 0) + hc;
    *v[i] = copysignf(y, *v[i]);
  }
}

/* kind: 0 linear (no stage), 1 sRGB, 2 BT.709, 3 PQ (param = intensity_target), 4 HLG (param = exponent, lum),
 * 5 gamma (param = exponent) */
void jxlo_from_linear(int kind, float param, const float lum[3], float* r, float* g, float* b, size_t n) {
  for (size_t i = 0; i < n; i++) {
    switch (kind) {
      case 1: r[i] = jxlo_linear_to_srgb1(r[i]); g[i] = jxlo_linear_to_srgb1(g[i]); b[i] = jxlo_linear_to_srgb1(b[i]); break;
      case 2: r[i] = jxlo_linear_to_bt709_1(r[i]); g[i] = jxlo_linear_to_bt709_1(g[i]); b[i] = jxlo_linear_to_bt709_1(b[i]); break;
      case 3: r[i] = jxlo_linear_to_pq_1(param, r[i]); g[i] = jxlo_linear_to_pq_1(param, g[i]); b[i] = jxlo_linear_to_pq_1(param, b[i]); break;
      case 4: jxlo_linear_to_hlg_1(param, lum, &r[i], &g[i], &b[i]); break;
      case 5: r[i] = jxlo_linear_to_gamma_1(param, r[i]); g[i] = jxlo_linear_to_gamma_1(param, g[i]); b[i] = jxlo_linear_to_gamma_1(param, b[i]); break;
      default: break;
    }
  }
}

/* XybStage -> FromLinearStage(kind) -> ConvertF32ToU8 / U16, the general form of jxlo_xyb_to_rgb8/16 */
void jxlo_xyb_to_rgb_tf(const JxloXybParams* p, int kind, float param, const float lum[3], const float* px, const float* py,
                        const float* pb, size_t w, size_t h, size_t stride, int bits, void* out, size_t out_stride_elems,
                        int out_channels) {
  for (size_t y = 0; y < h; y++) {
    for (size_t x = 0; x < w; x++) {
      float r = px[y * stride + x], g = py[y * stride + x], b = pb[y * stride + x];
      jxlo_xyb_to_linear(p, &r, &g, &b, 1);
      jxlo_from_linear(kind, param, lum, &r, &g, &b, 1);
      const float v[3] = {r, g, b};
      if (bits == 8) {
        uint8_t* o = (uint8_t*)out + y * out_stride_elems + x * (size_t)out_channels;
        for (int c = 0; c < 3; c++) o[c] = jxlo_f32_to_u8(v[c], x, y, c, 8);
        if (out_channels == 4) o[3] = 255;
      } else {
        uint16_t* o = (uint16_t*)out + y * out_stride_elems + x * (size_t)out_channels;
        for (int c = 0; c < 3; c++) o[c] = jxlo_f32_to_u16(v[c], 16);
        if (out_channels == 4) o[3] = 65535;
      }
    }
  }
}
