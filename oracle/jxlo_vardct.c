/*
 * ORACLE (test infrastructure, see jxlo.h) -- VarDCT reconstruction chain:
 * LF dequant, adaptive LF smoothing, per-group dequant + chroma-from-luma +
 * transform_to_pixels, sigma map, Gaborish, EPF0/1/2, with the render pipeline's
 * per-stage mirror semantics.
 *
 * Reference map:
 *   dequant_lf ................. jxl/src/frame/modular/mod.rs:837-929
 *   adaptive_lf_smoothing ...... jxl/src/frame/adaptive_lf_smoothing.rs:15-125
 *   adjust_quant_bias / dequant_lane / dequant_block
 *                                jxl/src/frame/group.rs:85-177
 *   decode_vardct_group (non-entropy part)
 *                                jxl/src/frame/group.rs:395-396, :454-504, :579-613, :181-253
 *   SigmaSource::new ........... jxl/src/features/epf.rs:35-87
 *   Gaborish ................... jxl/src/render/stages/gaborish.rs:20-27, :83-85
 *   EPF ........................ jxl/src/render/stages/epf/{common,epf0,epf1,epf2}.rs
 *   mirror / edge semantics .... jxl/src/util/mirror.rs:8-19,
 *                                jxl/src/render/simple_pipeline/run_stage.rs:129-146
 *   stage order ................ jxl/src/frame/render.rs:569-622
 */
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
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

void jxlo_default_frame_params(JxloFrameParams* p, int xsize, int ysize) {
  memset(p, 0, sizeof *p);
  p->xsize = xsize;
  p->ysize = ysize;
  p->xsize_blocks = (xsize + 7) / 8;
  p->ysize_blocks = (ysize + 7) / 8;
  p->group_dim = 256;
  p->global_scale = 21845;
  p->quant_lf = 16;
  p->lf_quant_factors[0] = 1.0f / 4096.0f; /* quant_weights.rs:24-30 */
  p->lf_quant_factors[1] = 1.0f / 512.0f;
  p->lf_quant_factors[2] = 1.0f / 256.0f;
  p->quant_biases[0] = 1.0f - 0.05465007330715401f; /* headers/transform_data.rs:30-31 */
  p->quant_biases[1] = 1.0f - 0.07005449891748593f;
  p->quant_biases[2] = 1.0f - 0.049935103337343655f;
  p->quant_biases[3] = 0.145f;
  p->x_qm_scale = 3;
  p->b_qm_scale = 2;
  p->color_factor = 84;
  p->base_correlation_x = 0.0f;
  p->base_correlation_b = 1.0f;
  p->gab = 1;
  for (int c = 0; c < 3; c++) {
    p->gab_w1[c] = 0.115169525f;
    p->gab_w2[c] = 0.061248592f;
  }
  p->epf_iters = 2;
  for (int i = 0; i < 8; i++) p->epf_sharp_lut[i] = (float)i / 7.0f;
  p->epf_sharp_lut[7] = 1.0f;
  p->epf_channel_scale[0] = 40.0f;
  p->epf_channel_scale[1] = 5.0f;
  p->epf_channel_scale[2] = 3.5f;
  p->epf_quant_mul = 0.46f;
  p->epf_pass0_sigma_scale = 0.9f;
  p->epf_pass2_sigma_scale = 6.5f;
  p->epf_border_sad_mul = 2.0f / 3.0f;
  p->do_lf_smoothing = 1;
}

static inline float inv_global_scale(const JxloFrameParams* p) { /* quantizer.rs:79-81 */
  return (float)(1 << 16) / (float)p->global_scale;
}

/* ---------------- K0a ---------------- */
void jxlo_dequant_lf(const JxloFrameParams* p, const int32_t* qy, const int32_t* qx,
                     const int32_t* qb, float mul, size_t n, float* out_x, float* out_y,
                     float* out_b) {
  const float inv_quant_lf = (float)(1 << 16) / ((float)p->global_scale * (float)p->quant_lf);
  const float fac_x = (p->lf_quant_factors[0] * inv_quant_lf) * mul;
  const float fac_y = (p->lf_quant_factors[1] * inv_quant_lf) * mul;
  const float fac_b = (p->lf_quant_factors[2] * inv_quant_lf) * mul;
  const float cfl_x = p->base_correlation_x + (float)p->ytox_lf / (float)p->color_factor;
  const float cfl_b = p->base_correlation_b + (float)p->ytob_lf / (float)p->color_factor;
  for (size_t i = 0; i < n; i++) {
    const float in_x = (float)qx[i] * fac_x;
    const float in_y = (float)qy[i] * fac_y;
    const float in_b = (float)qb[i] * fac_b;
    out_y[i] = in_y;
    out_x[i] = in_y * cfl_x + in_x; /* plain scalar Rust: never contracted */
    out_b[i] = in_y * cfl_b + in_b;
  }
}

void jxlo_dequant_lf_channel(const JxloFrameParams* p, int c, const int32_t* q, float mul, size_t n, float* out) {
  const float inv_quant_lf = (float)(1 << 16) / ((float)p->global_scale * (float)p->quant_lf);
  const float fac = (p->lf_quant_factors[c] * inv_quant_lf) * mul; /* modular/mod.rs:884 */
  for (size_t i = 0; i < n; i++) out[i] = (float)q[i] * fac;
}

/* ---------------- K0b ---------------- */
static const float kWSide = 0.20345139757231578f;
static const float kWCorner = 0.0334829185968739f;

/* rows [y0, y1) of adaptive_lf_smoothing (the image must be larger than 2 x 2) */
static void lf_smoothing_rows(const JxloFrameParams* p, const float* const in[3], int w, int h,
                              float* const out[3], int y0, int y1) {
  const float w_center = 1.0f - 4.0f * (kWSide + kWCorner);
  /* finalize_lf (frame/mod.rs:360-369): inv_quant_lf * quant_factors[c] */
  const float inv_quant_lf = inv_global_scale(p) / (float)p->quant_lf;
  float lf_factors[3];
  for (int c = 0; c < 3; c++) lf_factors[c] = inv_quant_lf * p->lf_quant_factors[c];
  for (int y = y0; y < y1; y++) {
    for (int x = 0; x < w; x++) {
      const size_t i = (size_t)y * w + x;
      if (y == 0 || y == h - 1 || x == 0 || x == w - 1) {
        for (int c = 0; c < 3; c++) out[c][i] = in[c][i];
        continue;
      }
      float gap = 0.5f, mc[3], sm[3];
      for (int c = 0; c < 3; c++) {
        const float* t = in[c] + i - w;
        const float* m = in[c] + i;
        const float* b = in[c] + i + w;
        const float corner = t[-1] + t[1] + b[-1] + b[1];
        const float side = m[-1] + m[1] + t[0] + b[0];
        mc[c] = m[0];
        sm[c] = corner * kWCorner + side * kWSide + mc[c] * w_center;
        const float g = fabsf((mc[c] - sm[c]) / lf_factors[c]);
        gap = gap > g ? gap : g; /* f32::max(gap, g) */
      }
      float factor = 3.0f - 4.0f * gap;
      factor = factor > 0.0f ? factor : 0.0f;
      for (int c = 0; c < 3; c++) out[c][i] = (sm[c] - mc[c]) * factor + mc[c];
    }
  }
}

void jxlo_adaptive_lf_smoothing(const JxloFrameParams* p, const float* const in[3], int w, int h,
                                float* const out[3]) {
  if (h <= 2 || w <= 2) {
    for (int c = 0; c < 3; c++) memcpy(out[c], in[c], sizeof(float) * (size_t)w * h);
    return;
  }
  lf_smoothing_rows(p, in, w, h, out, 0, h);
}

/* ---------------- K3 sigma ---------------- */
static void sigma_map_range(const JxloFrameParams* p, const int32_t* raw_quant, const uint8_t* epf_map,
                            float* inv_sigma, size_t i0, size_t i1) {
  const float kInvSigmaNum = -1.1715728752538099024f;
  const float quant_scale = 1.0f / inv_global_scale(p);
  for (size_t i = i0; i < i1; i++) {
    const float sigma_quant = p->epf_quant_mul / (quant_scale * (float)raw_quant[i] * kInvSigmaNum);
    float sigma = sigma_quant * p->epf_sharp_lut[epf_map[i]];
    sigma = sigma < -1e-4f ? sigma : -1e-4f; /* f32::min */
    inv_sigma[i] = 1.0f / sigma;
  }
}

void jxlo_sigma_map(const JxloFrameParams* p, const int32_t* raw_quant, const uint8_t* epf_map,
                    float* inv_sigma) {
  sigma_map_range(p, raw_quant, epf_map, inv_sigma, 0, (size_t)p->xsize_blocks * p->ysize_blocks);
}

/* ---------------- K1 ---------------- */
static inline float adjust_quant_bias(int32_t q, float bias_c, float bias3) { /* group.rs:85-96 */
  const float quant = (float)q;
  if ((q < 0 ? -q : q) < 2) return quant * bias_c;
  return quant - bias3 / quant;
}

void jxlo_decode_group(const JxloFrameParams* p, int group, const int32_t* coeffs,
                       const uint8_t* transform_map, const int32_t* raw_quant,
                       const int8_t* ytox_map, const int8_t* ytob_map,
                       const float* const lf[3], const float* const tables[17],
                       float* const planes[3], size_t stride) {
  const int gdb = p->group_dim / 8; /* group dim in blocks */
  const int xgroups = (p->xsize + p->group_dim - 1) / p->group_dim;
  const int gx = group % xgroups, gy = group / xgroups;
  const int bx0 = gx * gdb, by0 = gy * gdb;
  const int bw = (p->xsize_blocks - bx0) < gdb ? (p->xsize_blocks - bx0) : gdb;
  const int bh = (p->ysize_blocks - by0) < gdb ? (p->ysize_blocks - by0) : gdb;
  const size_t mstride = (size_t)p->xsize_blocks;
  const size_t cstride = (size_t)((p->xsize_blocks + 7) / 8);
  const size_t gsz = (size_t)p->group_dim * p->group_dim;
  /* group.rs:395-396 */
  const float x_dm = powf(1.0f / 1.25f, (float)p->x_qm_scale - 2.0f);
  const float b_dm = powf(1.0f / 1.25f, (float)p->b_qm_scale - 2.0f);
  const float igs = inv_global_scale(p);
  float* tb[3];
  for (int c = 0; c < 3; c++) tb[c] = (float*)malloc(sizeof(float) * 65536);
  float lfbuf[1024];
  size_t off = 0;
  for (int by = 0; by < bh; by++) {
    for (int bx = 0; bx < bw; bx++) {
      const size_t mi = (size_t)(by0 + by) * mstride + (bx0 + bx);
      const uint8_t raw = transform_map[mi];
      if (raw < 128) continue;
      const int type = raw & 127;
      const int cx = jxlo_covered_blocks_x(type), cy = jxlo_covered_blocks_y(type);
      const size_t n = (size_t)cx * cy * 64;
      /* chroma-from-luma multipliers of the 64x64 tile holding the top-left block (:463-465) */
      const size_t ci = (size_t)((by0 + by) / 8) * cstride + (size_t)((bx0 + bx) / 8);
      const float x_cc = p->base_correlation_x + (float)ytox_map[ci] / (float)p->color_factor;
      const float b_cc = p->base_correlation_b + (float)ytob_map[ci] / (float)p->color_factor;
      /* dequant_block :153-176 */
      const float sdy = igs / (float)(uint32_t)raw_quant[mi];
      const float sdx = sdy * x_dm;
      const float sdb = sdy * b_dm;
      const int tab = jxlo_quant_table_for_type(type);
      const float* m = tables[tab];
      const size_t size = (size_t)jxlo_quant_table_size(tab);
      const int32_t* qx = coeffs + off;
      const int32_t* qy = coeffs + gsz + off;
      const int32_t* qb = coeffs + 2 * gsz + off;
      for (size_t k = 0; k < n; k++) { /* dequant_lane :100-133 */
        const float x_mul = m[k] * sdx;
        const float y_mul = m[size + k] * sdy;
        const float b_mul = m[2 * size + k] * sdb;
        const float dx = adjust_quant_bias(qx[k], p->quant_biases[0], p->quant_biases[3]) * x_mul;
        const float dy = adjust_quant_bias(qy[k], p->quant_biases[1], p->quant_biases[3]) * y_mul;
        const float db = adjust_quant_bias(qb[k], p->quant_biases[2], p->quant_biases[3]) * b_mul;
        tb[0][k] = mul_add(x_cc, dy, dx);
        tb[1][k] = dy;
        tb[2][k] = mul_add(b_cc, dy, db);
      }
      for (int c = 0; c < 3; c++) {
        /* sub-sampled channels only hold the blocks aligned to their sampling (:223-226); bx, by are
         * group-local there, and a group is an even number of blocks */
        const int hs = p->hshift[c], vs = p->vshift[c];
        if ((((bx >> hs) << hs) != bx) || (((by >> vs) << vs) != by)) continue;
        /* LF patch cy x cx (:227-235); the sub-sampled LF samples of an LF group sit in the top-left
         * corner of the group's rectangle of the LF image (:485-504) */
        const int lfg = p->group_dim; /* LF group = group_dim blocks (8 * group_dim pixels) */
        const int lfbx = ((bx0 + bx) / lfg) * lfg, lfby = ((by0 + by) / lfg) * lfg;
        const int lx = lfbx + ((bx0 + bx - lfbx) >> hs), ly = lfby + ((by0 + by - lfby) >> vs);
        for (int y = 0; y < cy; y++)
          for (int x = 0; x < cx; x++) lfbuf[y * cx + x] = lf[c][(size_t)(ly + y) * mstride + (lx + x)];
        jxlo_transform_to_pixels(type, lfbuf, tb[c]);
        /* copy R x C pixels into the plane at the down-sampled origin (:237-250) */
        const int R = cy * 8, C = cx * 8;
        const size_t ox = (size_t)(((bx0 + bx) * 8) >> hs), oy = (size_t)(((by0 + by) * 8) >> vs);
        for (int y = 0; y < R; y++)
          memcpy(planes[c] + (oy + y) * stride + ox, tb[c] + (size_t)y * C, sizeof(float) * C);
      }
      off += n;
    }
  }
  for (int c = 0; c < 3; c++) free(tb[c]);
}

/* ---------------- stage edge semantics ---------------- */
static inline int mirror(int v, int s) { /* util/mirror.rs:8-19 */
  for (;;) {
    if (v < 0)
      v = -v - 1;
    else if (v >= s)
      v = s * 2 - v - 1;
    else
      return v;
  }
}

#define PIX(pl, x, y) ((pl)[(size_t)mirror((y), h) * stride + (size_t)mirror((x), w)])

/* ---------------- chroma upsampling (render/stages/chroma_upsample.rs) ---------------- */
void jxlo_chroma_upsample_h(const float* in, int ws, int hs, size_t in_stride, float* out, size_t out_stride) {
  for (int y = 0; y < hs; y++) {
    const float* r = in + (size_t)y * in_stride;
    for (int x = 0; x < ws; x++) {
      const float prev = r[mirror(x - 1, ws)], cur = r[x], next = r[mirror(x + 1, ws)];
      out[(size_t)y * out_stride + 2 * x] = mul_add(prev, 0.25f, cur * 0.75f);     /* :53 */
      out[(size_t)y * out_stride + 2 * x + 1] = mul_add(next, 0.25f, cur * 0.75f); /* :56 */
    }
  }
}

void jxlo_chroma_upsample_v(const float* in, int ws, int hs, size_t in_stride, float* out, size_t out_stride) {
  for (int y = 0; y < hs; y++) {
    const float* rp = in + (size_t)mirror(y - 1, hs) * in_stride;
    const float* rc = in + (size_t)y * in_stride;
    const float* rn = in + (size_t)mirror(y + 1, hs) * in_stride;
    for (int x = 0; x < ws; x++) {
      out[(size_t)(2 * y) * out_stride + x] = mul_add(rp[x], 0.25f, rc[x] * 0.75f);     /* :137 */
      out[(size_t)(2 * y + 1) * out_stride + x] = mul_add(rn[x], 0.25f, rc[x] * 0.75f); /* :140 */
    }
  }
}

/* ---------------- 2x / 4x / 8x upsampling (render/stages/upsample.rs) ---------------- */
#include "upsampling_weights.inc"

void jxlo_upsample_kernels(int n, const float* weights, float* flat) {
  if (!weights) weights = n == 2 ? kDefaultUpsamplingWeights2 : n == 4 ? kDefaultUpsamplingWeights4 : kDefaultUpsamplingWeights8;
  const int half = n / 2;
  /* upsample.rs:31-50: the weights are the upper triangle of the symmetric (5*half)^2 top-left quadrant */
  for (int i = 0; i < 5 * half; i++) {
    for (int j = 0; j < 5 * half; j++) {
      const int y = i < j ? i : j, x = i < j ? j : i;
      const float wv = weights[5 * half * y - y * (y - 1) / 2 + x - y];
      const int dj = j / 5, di = i / 5, kj = j % 5, ki = i % 5, last = 2 * half - 1;
      /* kernel[a][b][c][d] -> flat[(a*n + b)*25 + c*5 + d] */
      flat[((dj)*n + di) * 25 + kj * 5 + ki] = wv;
      flat[((last - dj) * n + di) * 25 + (4 - kj) * 5 + ki] = wv;
      flat[((dj)*n + (last - di)) * 25 + kj * 5 + (4 - ki)] = wv;
      flat[((last - dj) * n + (last - di)) * 25 + (4 - kj) * 5 + (4 - ki)] = wv;
    }
  }
}

void jxlo_upsample(int n, const float* weights, const float* in, int w, int h, size_t in_stride, float* out,
                   size_t out_stride) {
  float flat[64 * 25];
  jxlo_upsample_kernels(n, weights, flat);
  for (int y = 0; y < h; y++) {
    for (int x = 0; x < w; x++) {
      float win[25], mn, mx;
      for (int ky = 0; ky < 5; ky++)
        for (int kx = 0; kx < 5; kx++)
          win[ky * 5 + kx] = in[(size_t)mirror(y - 2 + ky, h) * in_stride + (size_t)mirror(x - 2 + kx, w)];
      mn = mx = win[0];
      for (int t = 1; t < 25; t++) { /* compute_minmax, :115-172 */
        mn = win[t] < mn ? win[t] : mn;
        mx = win[t] > mx ? win[t] : mx;
      }
      for (int oy = 0; oy < n; oy++) {
        for (int ox = 0; ox < n; ox++) {
          const float* k = flat + (oy * n + ox) * 25;
          /* kernel_conv (:175-215): three accumulators, tap t feeds accumulator t % 3 */
          float acc[3] = {win[0] * k[0], win[1] * k[1], win[2] * k[2]};
          for (int t = 3; t < 25; t++) acc[t % 3] = mul_add(win[t], k[t], acc[t % 3]);
          float v = acc[0] + acc[1] + acc[2];
          v = v > mn ? v : mn; /* .max(minval).min(maxval) */
          v = v < mx ? v : mx;
          out[(size_t)(y * n + oy) * out_stride + (size_t)(x * n + ox)] = v;
        }
      }
    }
  }
}

/* ---------------- K2 ---------------- */
void jxlo_gaborish_rows(const float* in, int w, int h, size_t stride, float w1, float w2,
                        float* out, int y0, int y1) {
  const float total = 1.0f + w1 * 4.0f + w2 * 4.0f; /* gaborish.rs:20-27 */
  const float k0 = 1.0f / total, k1 = w1 / total, k2 = w2 / total;
  for (int y = y0; y < y1; y++) {
    for (int x = 0; x < w; x++) {
      const float p00 = PIX(in, x - 1, y - 1), p01 = PIX(in, x, y - 1), p02 = PIX(in, x + 1, y - 1);
      const float p10 = PIX(in, x - 1, y), p11 = PIX(in, x, y), p12 = PIX(in, x + 1, y);
      const float p20 = PIX(in, x - 1, y + 1), p21 = PIX(in, x, y + 1), p22 = PIX(in, x + 1, y + 1);
      float sum = p11 * k0; /* :83-85 */
      sum = mul_add(k1, p01 + p10 + p21 + p12, sum);
      sum = mul_add(k2, p00 + p02 + p20 + p22, sum);
      out[(size_t)y * stride + x] = sum;
    }
  }
}
void jxlo_gaborish(const float* in, int w, int h, size_t stride, float w1, float w2, float* out) {
  jxlo_gaborish_rows(in, w, h, stride, w1, w2, out, 0, h);
}

/* ---------------- K3 ---------------- */
static const float kMinSigma = -3.90524291751269967465540850526868f; /* lib.rs:28 */

static inline float sad_mul_at(int x, int y, float sm, float bsm) { /* epf/common.rs:31-41 */
  const int xm = x & 7, ym = y & 7;
  return (xm == 0 || xm == 7 || ym == 0 || ym == 7) ? bsm : sm;
}

#define AD(a, b) fabsf((a) - (b))

static void epf0_px(const JxloFrameParams* p, const float* const in[3], int w, int h, size_t stride,
                    int x, int y, float inv_sigma, float* o) {
  /* epf0.rs:87-210.  Offsets (dx,dy) of the 12 neighbours in sads[] order. */
  static const int nb[12][2] = {{0, -2}, {-1, -1}, {0, -1}, {1, -1}, {-2, 0}, {-1, 0},
                                {1, 0},  {2, 0},   {-1, 1}, {0, 1},  {1, 1},  {0, 2}};
  /* plus5 offsets in the order the reference adds them: each row of the table in
   * :165-176 lists, for neighbour n, |c+d - (n+d)| with d running over
   * (0,-1)... in a neighbour-specific order.  We restate the table literally below. */
  float sads[12] = {0};
  for (int c = 0; c < 3; c++) {
    const float* q = in[c];
    const float scale = p->epf_channel_scale[c];
#define P(cx, cy) PIX(q, x + (cx) - 3, y + (cy) - 3)
    const float p30 = P(3, 0), p21 = P(2, 1), p31 = P(3, 1), p41 = P(4, 1), p12 = P(1, 2),
                p22 = P(2, 2), p32 = P(3, 2), p42 = P(4, 2), p52 = P(5, 2), p03 = P(0, 3),
                p13 = P(1, 3), p23 = P(2, 3), p33 = P(3, 3), p43 = P(4, 3), p53 = P(5, 3),
                p63 = P(6, 3), p14 = P(1, 4), p24 = P(2, 4), p34 = P(3, 4), p44 = P(4, 4),
                p54 = P(5, 4), p25 = P(2, 5), p35 = P(3, 5), p45 = P(4, 5), p36 = P(3, 6);
#undef P
    const float d32_30 = AD(p32, p30), d32_21 = AD(p32, p21), d32_31 = AD(p32, p31),
                d32_41 = AD(p32, p41), d32_12 = AD(p32, p12), d32_22 = AD(p32, p22),
                d32_42 = AD(p32, p42), d32_52 = AD(p32, p52), d32_23 = AD(p32, p23),
                d32_34 = AD(p32, p34), d32_43 = AD(p32, p43), d32_33 = AD(p32, p33),
                d23_21 = AD(p23, p21), d23_12 = AD(p23, p12), d23_22 = AD(p23, p22),
                d23_03 = AD(p23, p03), d23_13 = AD(p23, p13), d23_33 = AD(p23, p33),
                d23_43 = AD(p23, p43), d23_14 = AD(p23, p14), d23_24 = AD(p23, p24),
                d23_34 = AD(p23, p34), d23_25 = AD(p23, p25), d33_31 = AD(p33, p31),
                d33_22 = AD(p33, p22), d33_42 = AD(p33, p42), d33_13 = AD(p33, p13),
                d33_43 = AD(p33, p43), d33_53 = AD(p33, p53), d33_24 = AD(p33, p24),
                d33_34 = AD(p33, p34), d33_44 = AD(p33, p44), d33_35 = AD(p33, p35),
                d43_41 = AD(p43, p41), d43_42 = AD(p43, p42), d43_52 = AD(p43, p52),
                d43_53 = AD(p43, p53), d43_63 = AD(p43, p63), d43_34 = AD(p43, p34),
                d43_44 = AD(p43, p44), d43_54 = AD(p43, p54), d43_45 = AD(p43, p45),
                d34_14 = AD(p34, p14), d34_24 = AD(p34, p24), d34_44 = AD(p34, p44),
                d34_54 = AD(p34, p54), d34_25 = AD(p34, p25), d34_35 = AD(p34, p35),
                d34_45 = AD(p34, p45), d34_36 = AD(p34, p36);
    sads[0] = mul_add(scale, d32_30 + d23_21 + d33_31 + d43_41 + d32_34, sads[0]);
    sads[1] = mul_add(scale, d32_21 + d23_12 + d33_22 + d32_43 + d23_34, sads[1]);
    sads[2] = mul_add(scale, d32_31 + d23_22 + d32_33 + d43_42 + d33_34, sads[2]);
    sads[3] = mul_add(scale, d32_41 + d32_23 + d33_42 + d43_52 + d43_34, sads[3]);
    sads[4] = mul_add(scale, d32_12 + d23_03 + d33_13 + d23_43 + d34_14, sads[4]);
    sads[5] = mul_add(scale, d32_22 + d23_13 + d23_33 + d33_43 + d34_24, sads[5]);
    sads[6] = mul_add(scale, d32_42 + d23_33 + d33_43 + d43_53 + d34_44, sads[6]);
    sads[7] = mul_add(scale, d32_52 + d23_43 + d33_53 + d43_63 + d34_54, sads[7]);
    sads[8] = mul_add(scale, d32_23 + d23_14 + d33_24 + d43_34 + d34_25, sads[8]);
    sads[9] = mul_add(scale, d32_33 + d23_24 + d33_34 + d43_44 + d34_35, sads[9]);
    sads[10] = mul_add(scale, d32_43 + d23_34 + d33_44 + d43_54 + d34_45, sads[10]);
    sads[11] = mul_add(scale, d32_34 + d23_25 + d33_35 + d43_45 + d34_36, sads[11]);
  }
  float wsum = 1.0f;
  for (int i = 0; i < 12; i++) {
    float v = mul_add(sads[i], inv_sigma, 1.0f);
    sads[i] = v > 0.0f ? v : 0.0f;
    wsum += sads[i];
  }
  const float inv_w = 1.0f / wsum;
  for (int c = 0; c < 3; c++) {
    float acc = PIX(in[c], x, y);
    for (int i = 11; i >= 0; i--) acc = mul_add(PIX(in[c], x + nb[i][0], y + nb[i][1]), sads[i], acc);
    o[c] = acc * inv_w;
  }
}

static void epf1_px(const JxloFrameParams* p, const float* const in[3], int w, int h, size_t stride,
                    int x, int y, float inv_sigma, float* o) {
  /* epf1.rs:84-146 */
  float sads[4] = {0};
  for (int c = 0; c < 3; c++) {
    const float* q = in[c];
    const float scale = p->epf_channel_scale[c];
#define P(cx, cy) PIX(q, x + (cx) - 2, y + (cy) - 2)
    const float p20 = P(2, 0), p11 = P(1, 1), p21 = P(2, 1), p31 = P(3, 1), p02 = P(0, 2),
                p12 = P(1, 2), p22 = P(2, 2), p32 = P(3, 2), p42 = P(4, 2), p13 = P(1, 3),
                p23 = P(2, 3), p33 = P(3, 3), p24 = P(2, 4);
#undef P
    const float d20_21 = AD(p20, p21), d11_21 = AD(p11, p21), d22_21 = AD(p22, p21),
                d31_21 = AD(p31, p21), d02_12 = AD(p02, p12), d11_12 = AD(p11, p12),
                d12_22 = AD(p22, p12), d31_32 = AD(p31, p32), d22_32 = AD(p22, p32),
                d42_32 = AD(p42, p32), d13_12 = AD(p13, p12), d22_23 = AD(p22, p23),
                d13_23 = AD(p13, p23), d33_23 = AD(p33, p23), d33_32 = AD(p33, p32),
                d24_23 = AD(p24, p23);
    sads[0] = mul_add(d20_21 + d11_12 + d22_21 + d31_32 + d22_23, scale, sads[0]);
    sads[1] = mul_add(d11_21 + d02_12 + d12_22 + d22_32 + d13_23, scale, sads[1]);
    sads[2] = mul_add(d31_21 + d12_22 + d22_32 + d42_32 + d33_23, scale, sads[2]);
    sads[3] = mul_add(d22_21 + d13_12 + d22_23 + d33_32 + d24_23, scale, sads[3]);
  }
  float wsum = 1.0f;
  for (int i = 0; i < 4; i++) {
    float v = mul_add(sads[i], inv_sigma, 1.0f);
    sads[i] = v > 0.0f ? v : 0.0f;
    wsum += sads[i];
  }
  const float inv_w = 1.0f / wsum;
  for (int c = 0; c < 3; c++) {
    const float* q = in[c];
    float acc = PIX(q, x, y);
    acc = mul_add(PIX(q, x, y + 1), sads[3], acc);
    acc = mul_add(PIX(q, x + 1, y), sads[2], acc);
    acc = mul_add(PIX(q, x - 1, y), sads[1], acc);
    acc = mul_add(PIX(q, x, y - 1), sads[0], acc);
    o[c] = acc * inv_w;
  }
}

static void epf2_px(const JxloFrameParams* p, const float* const in[3], int w, int h, size_t stride,
                    int x, int y, float inv_sigma, float* o) {
  /* epf2.rs:84-136 */
  static const int nb[4][2] = {{0, -1}, {-1, 0}, {1, 0}, {0, 1}};
  const float xc = PIX(in[0], x, y), yc = PIX(in[1], x, y), bc = PIX(in[2], x, y);
  float wacc = 1.0f, xa = xc, ya = yc, ba = bc;
  for (int i = 0; i < 4; i++) {
    const float cx = PIX(in[0], x + nb[i][0], y + nb[i][1]);
    const float cy = PIX(in[1], x + nb[i][0], y + nb[i][1]);
    const float cb = PIX(in[2], x + nb[i][0], y + nb[i][1]);
    const float sad =
        mul_add(AD(cx, xc), p->epf_channel_scale[0],
                mul_add(AD(cy, yc), p->epf_channel_scale[1], AD(cb, bc) * p->epf_channel_scale[2]));
    float wgt = mul_add(sad, inv_sigma, 1.0f);
    wgt = wgt > 0.0f ? wgt : 0.0f;
    wacc += wgt;
    xa = mul_add(wgt, cx, xa);
    ya = mul_add(wgt, cy, ya);
    ba = mul_add(wgt, cb, ba);
  }
  const float inv_w = 1.0f / wacc;
  o[0] = xa * inv_w;
  o[1] = ya * inv_w;
  o[2] = ba * inv_w;
}

void jxlo_epf_rows(int stage, const JxloFrameParams* p, const float* const in[3], int w, int h,
                   size_t stride, const float* inv_sigma, size_t sigma_stride,
                   float* const out[3], int y0, int y1) {
  /* sigma_scale per stage: frame/render.rs:597-621 */
  const float sigma_scale =
      stage == 0 ? p->epf_pass0_sigma_scale : (stage == 1 ? 1.0f : p->epf_pass2_sigma_scale);
  const float sm = sigma_scale * 1.65f;
  const float bsm = sm * p->epf_border_sad_mul;
  for (int y = y0; y < y1; y++) {
    for (int x = 0; x < w; x++) {
      const float sigma = inv_sigma[(size_t)(y / 8) * sigma_stride + (size_t)(x / 8)];
      const size_t i = (size_t)y * stride + x;
      if (sigma < kMinSigma) { /* MIN_SIGMA > sigma: pass through */
        for (int c = 0; c < 3; c++) out[c][i] = in[c][i];
        continue;
      }
      const float is = sigma * sad_mul_at(x, y, sm, bsm);
      float o[3];
      if (stage == 0)
        epf0_px(p, in, w, h, stride, x, y, is, o);
      else if (stage == 1)
        epf1_px(p, in, w, h, stride, x, y, is, o);
      else
        epf2_px(p, in, w, h, stride, x, y, is, o);
      for (int c = 0; c < 3; c++) out[c][i] = o[c];
    }
  }
}
void jxlo_epf(int stage, const JxloFrameParams* p, const float* const in[3], int w, int h,
              size_t stride, const float* inv_sigma, size_t sigma_stride, float* const out[3]) {
  jxlo_epf_rows(stage, p, in, w, h, stride, inv_sigma, sigma_stride, out, 0, h);
}

/* ---------------- threaded whole-chain driver (cpu_baseline harness) ---------------- */
typedef struct {
  void (*fn)(void*, int, int);
  void* arg;
  int n;
  int next;
  pthread_mutex_t mu;
} Pool;

static void* pool_worker(void* v) {
  Pool* pl = (Pool*)v;
  for (;;) {
    pthread_mutex_lock(&pl->mu);
    int i = pl->next++;
    pthread_mutex_unlock(&pl->mu);
    if (i >= pl->n) return NULL;
    pl->fn(pl->arg, i, pl->n);
  }
}

static void run_parallel(int nthreads, int n, void (*fn)(void*, int, int), void* arg) {
  Pool pl;
  pl.fn = fn;
  pl.arg = arg;
  pl.n = n;
  pl.next = 0;
  pthread_mutex_init(&pl.mu, NULL);
  if (nthreads <= 1) {
    pool_worker(&pl);
  } else {
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * nthreads);
    for (int i = 0; i < nthreads; i++) pthread_create(&th[i], NULL, pool_worker, &pl);
    for (int i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
    free(th);
  }
  pthread_mutex_destroy(&pl.mu);
}

typedef struct {
  const JxloFrameParams* p;
  const int32_t* coeffs;
  const uint8_t* transform_map;
  const int32_t* raw_quant;
  const int8_t *ytox, *ytob;
  const float* lf[3];
  const float* const* tables;
  float* const* planes;
  size_t stride;
  /* stage job */
  int stage; /* -1 gaborish */
  const float* sin[3];
  float* sout[3];
  const float* sigma;
  int rows_per_job;
  /* LF smoothing / sigma jobs */
  const float* lf_in[3];
  float* lf_out[3];
  const uint8_t* epf_map;
  float* sigma_out;
} Job;

static void lf_job(void* v, int i, int n) {
  (void)n;
  Job* j = (Job*)v;
  const int h = j->p->ysize_blocks, w = j->p->xsize_blocks;
  int y0 = i * 16, y1 = y0 + 16;
  if (y1 > h) y1 = h;
  lf_smoothing_rows(j->p, j->lf_in, w, h, j->lf_out, y0, y1);
}

static void sigma_job(void* v, int i, int n) {
  (void)n;
  Job* j = (Job*)v;
  const int h = j->p->ysize_blocks, w = j->p->xsize_blocks;
  int y0 = i * 16, y1 = y0 + 16;
  if (y1 > h) y1 = h;
  sigma_map_range(j->p, j->raw_quant, j->epf_map, j->sigma_out, (size_t)y0 * w, (size_t)y1 * w);
}

static void group_job(void* v, int i, int n) {
  (void)n;
  Job* j = (Job*)v;
  const size_t gsz = (size_t)j->p->group_dim * j->p->group_dim;
  jxlo_decode_group(j->p, i, j->coeffs + (size_t)i * 3 * gsz, j->transform_map, j->raw_quant,
                    j->ytox, j->ytob, j->lf, j->tables, j->planes, j->stride);
}

static void stage_job(void* v, int i, int n) {
  (void)n;
  Job* j = (Job*)v;
  const int h = j->p->ysize, w = j->p->xsize;
  int y0 = i * j->rows_per_job, y1 = y0 + j->rows_per_job;
  if (y1 > h) y1 = h;
  if (j->stage == -2) {
    for (int c = 0; c < 3; c++)
      for (int y = y0; y < y1; y++)
        memcpy(j->sout[c] + (size_t)y * j->stride, j->sin[c] + (size_t)y * j->stride, sizeof(float) * (size_t)w);
  } else if (j->stage < 0) {
    for (int c = 0; c < 3; c++)
      jxlo_gaborish_rows(j->sin[c], w, h, j->stride, j->p->gab_w1[c], j->p->gab_w2[c], j->sout[c],
                         y0, y1);
  } else {
    jxlo_epf_rows(j->stage, j->p, j->sin, w, h, j->stride, j->sigma, (size_t)j->p->xsize_blocks,
                  j->sout, y0, y1);
  }
}

void jxlo_vardct_frame(const JxloFrameParams* p, const int32_t* coeffs,
                       const uint8_t* transform_map, const int32_t* raw_quant,
                       const uint8_t* epf_map, const int8_t* ytox_map, const int8_t* ytob_map,
                       float* const lf[3], const float* const tables[17], float* const planes[3],
                       float* const tmp[3], size_t stride, int num_threads) {
  const int bw = p->xsize_blocks, bh = p->ysize_blocks;
  Job j;
  memset(&j, 0, sizeof j);
  j.p = p;
  j.raw_quant = raw_quant;
  j.epf_map = epf_map;
  const int lf_jobs = (bh + 15) / 16;
  if (p->do_lf_smoothing && bw > 2 && bh > 2) {
    float* sm[3];
    for (int c = 0; c < 3; c++) {
      sm[c] = (float*)malloc(sizeof(float) * (size_t)bw * bh);
      j.lf_in[c] = lf[c];
      j.lf_out[c] = sm[c];
    }
    run_parallel(num_threads, lf_jobs, lf_job, &j);
    for (int c = 0; c < 3; c++) {
      memcpy(lf[c], sm[c], sizeof(float) * (size_t)bw * bh);
      free(sm[c]);
    }
  }
  float* sigma = (float*)malloc(sizeof(float) * (size_t)bw * bh);
  j.sigma_out = sigma;
  if (p->epf_iters > 0) run_parallel(num_threads, lf_jobs, sigma_job, &j);
  j.coeffs = coeffs;
  j.transform_map = transform_map;
  j.raw_quant = raw_quant;
  j.ytox = ytox_map;
  j.ytob = ytob_map;
  for (int c = 0; c < 3; c++) j.lf[c] = lf[c];
  j.tables = tables;
  j.planes = planes;
  j.stride = stride;
  j.sigma = sigma;
  const int xg = (p->xsize + p->group_dim - 1) / p->group_dim;
  const int yg = (p->ysize + p->group_dim - 1) / p->group_dim;
  run_parallel(num_threads, xg * yg, group_job, &j);
  /* chroma upsampling first (frame/render.rs:569-576): horizontal, then vertical, per channel; the
   * channel's image is ceil(size / 2^shift) samples, mirrored at its own edges */
  for (int c = 0; c < 3; c++) {
    const int hs = p->hshift[c], vs = p->vshift[c];
    if (!hs && !vs) continue;
    int cw = (p->xsize + (1 << hs) - 1) >> hs, chh = (p->ysize + (1 << vs) - 1) >> vs;
    if (hs) {
      jxlo_chroma_upsample_h(planes[c], cw, chh, stride, tmp[c], stride);
      cw *= 2;
      for (int y = 0; y < chh; y++) memcpy(planes[c] + (size_t)y * stride, tmp[c] + (size_t)y * stride, sizeof(float) * cw);
    }
    if (vs) {
      jxlo_chroma_upsample_v(planes[c], cw, chh, stride, tmp[c], stride);
      chh *= 2;
      for (int y = 0; y < chh; y++) memcpy(planes[c] + (size_t)y * stride, tmp[c] + (size_t)y * stride, sizeof(float) * cw);
    }
  }
  /* stage list of frame/render.rs:569-622: gaborish, epf0 (iters>=3), epf1 (>=1), epf2 (>=2);
   * result always ends in planes[] */
  float* cur[3] = {planes[0], planes[1], planes[2]};
  float* oth[3] = {tmp[0], tmp[1], tmp[2]};
  int stages[4], ns = 0;
  if (p->gab) stages[ns++] = -1;
  if (p->epf_iters >= 3) stages[ns++] = 0;
  if (p->epf_iters >= 1) stages[ns++] = 1;
  if (p->epf_iters >= 2) stages[ns++] = 2;
  j.rows_per_job = 16;
  const int njobs = (p->ysize + j.rows_per_job - 1) / j.rows_per_job;
  for (int s = 0; s < ns; s++) {
    j.stage = stages[s];
    for (int c = 0; c < 3; c++) {
      j.sin[c] = cur[c];
      j.sout[c] = oth[c];
    }
    run_parallel(num_threads, njobs, stage_job, &j);
    for (int c = 0; c < 3; c++) {
      float* t = cur[c];
      cur[c] = oth[c];
      oth[c] = t;
    }
  }
  if (cur[0] != planes[0]) {
    j.stage = -2; /* copy rows of cur back into planes, in parallel like the stages */
    for (int c = 0; c < 3; c++) {
      j.sin[c] = cur[c];
      j.sout[c] = planes[c];
    }
    run_parallel(num_threads, njobs, stage_job, &j);
  }
  free(sigma);
}
