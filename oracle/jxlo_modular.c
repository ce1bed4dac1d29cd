/*
 * ORACLE (test infrastructure, see jxlo.h) -- Modular inverse transforms,
 * whole-plane form (the GPU path does not tile; neighbour-tile borders of
 * step.rs collapse to the "no neighbour" branches).
 *
 * Reference map:
 *   RCT ............. jxl/src/frame/modular/transforms/rct.rs:14-157
 *   Palette ......... jxl/src/frame/modular/transforms/palette.rs:24-199
 *   Squeeze ......... jxl/src/frame/modular/transforms/squeeze.rs:107-194 (tendency, unsqueeze),
 *                     :389-437 (hsqueeze_scalar), :576-644 (vsqueeze_scalar)
 * All arithmetic is wrapping i32 (jxl_simd/src/scalar.rs Wrapping<i32>); the
 * scalar squeeze definition uses i64 intermediates and truncating division.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "jxlo.h"

static inline int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static inline int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static inline int32_t sar(int32_t a, int s) { return a >> s; } /* arithmetic on gcc/clang */

void jxlo_rct(int32_t* p0, int32_t* p1, int32_t* p2, size_t n, int op, int perm) {
  for (size_t i = 0; i < n; i++) {
    int32_t v0 = p0[i], v1 = p1[i], v2 = p2[i];
    int32_t w0 = v0, w1 = v1, w2 = v2;
    switch (op) { /* rct.rs:22-45 */
      case 0: break;
      case 1: w2 = wadd(v2, v0); break;
      case 2: w1 = wadd(v1, v0); break;
      case 3: w1 = wadd(v1, v0); w2 = wadd(v2, v0); break;
      case 4: w1 = wadd(v1, sar(wadd(v0, v2), 1)); break;
      case 5: {
        int32_t t2 = wadd(v0, v2);
        w1 = wadd(v1, sar(wadd(v0, t2), 1));
        w2 = t2;
        break;
      }
      case 6: {
        int32_t y = v0, co = v1, cg = v2;
        y = wsub(y, sar(cg, 1));
        int32_t g = wadd(cg, y);
        y = wsub(y, sar(co, 1));
        int32_t r = wadd(y, co);
        w0 = r; w1 = g; w2 = y;
        break;
      }
    }
    /* permutation rct.rs:132-156: buffer contents after the pointer swaps */
    int32_t o0, o1, o2;
    switch (perm) {
      default:
      case 0: o0 = w0; o1 = w1; o2 = w2; break;             /* Rgb */
      case 1: o1 = w0; o2 = w1; o0 = w2; break;             /* Gbr: out[1,2,0] = in[0,1,2] */
      case 2: o2 = w0; o0 = w1; o1 = w2; break;             /* Brg: out[2,0,1] = in[0,1,2] */
      case 3: o0 = w0; o2 = w1; o1 = w2; break;             /* Rbg */
      case 4: o1 = w0; o0 = w1; o2 = w2; break;             /* Grb */
      case 5: o2 = w0; o1 = w1; o0 = w2; break;             /* Bgr */
    }
    p0[i] = o0; p1[i] = o1; p2[i] = o2;
  }
}

/* ---- palette ---- */
static const int16_t kDelta[72][3] = { /* palette.rs:46-119 (spec data) */
#include "delta_palette.inc"
};
static int32_t delta_entry(int idx, int c) { return kDelta[idx][c]; }

int32_t jxlo_palette_value(const int32_t* palette, size_t palette_stride, int64_t index, int c,
                           int palette_size, int bit_depth) { /* palette.rs:39-199 */
  if (index < 0) {
    if (c >= 3) return 0;
    uint64_t i = (uint64_t)(-(index + 1));
    i %= 1 + 2 * (72 - 1);
    static const int mult[2] = {-1, 1};
    int32_t result = delta_entry((int)((i + 1) >> 1), c) * mult[i & 1];
    if (bit_depth > 8) result *= 1 << (bit_depth - 8);
    return result;
  }
  uint64_t i = (uint64_t)index;
  const uint64_t ps = (uint64_t)palette_size;
  if (ps <= i && i < ps + 64) {
    if (c >= 3) return 0;
    i -= ps;
    i >>= c * 2;
    int sh = bit_depth - 3;
    if (sh < 0) sh = 0;
    return (int32_t)(((i % 4) * (uint64_t)((1 << bit_depth) - 1)) >> 2) + (1 << sh);
  } else if (ps + 64 <= i) {
    if (c >= 3) return 0;
    i -= ps + 64;
    if (c == 1) i /= 5;
    if (c == 2) i /= 25;
    return (int32_t)(((i % 5) * (uint64_t)((1 << bit_depth) - 1)) >> 2);
  }
  return palette[(size_t)c * palette_stride + i];
}

void jxlo_palette(const int32_t* index, size_t n, const int32_t* palette, int num_colors,
                  size_t palette_stride, int nb_channels, int bit_depth, int32_t* out) {
  for (int c = 0; c < nb_channels; c++)
    for (size_t i = 0; i < n; i++)
      out[(size_t)c * n + i] =
          jxlo_palette_value(palette, palette_stride, (int64_t)index[i], c, num_colors, bit_depth);
}

/* Predictor::predict_one without the weighted predictor (modular/predict.rs:152-198) on the neighbourhood
 * PredictionData::get builds (:96-137), i64 arithmetic, `/` truncating like Rust's */
static int64_t predict_one(int predictor, const int32_t* out, int w, int x, int y) {
  const int32_t* row = out + (size_t)y * w;
  const int32_t* row_top = out + (size_t)(y > 0 ? y - 1 : 0) * w;
  const int32_t* row_toptop = out + (size_t)(y > 1 ? y - 2 : 0) * w;
  const int64_t left = x > 0 ? row[x - 1] : (y > 0 ? row_top[0] : 0);
  const int64_t top = y > 0 ? row_top[x] : left;
  const int64_t topleft = (x > 0 && y > 0) ? row_top[x - 1] : left;
  const int64_t topright = (x + 1 < w && y > 0) ? row_top[x + 1] : top;
  const int64_t leftleft = x > 1 ? row[x - 2] : left;
  const int64_t toptop = y > 1 ? row_toptop[x] : top;
  const int64_t toprightright = (x + 2 < w && y > 0) ? row_top[x + 2] : topright;
  switch (predictor) {
    case 0: return 0;
    case 1: return left;
    case 2: return top;
    case 3: return (top + left) / 2;
    case 4: { /* select, :191-198 */
      const int64_t p = left + top - topleft;
      const int64_t dl = p - left < 0 ? left - p : p - left, dt = p - top < 0 ? top - p : p - top;
      return dl < dt ? left : top;
    }
    case 5: { /* clamped_gradient, :139-146 */
      const int64_t mn = left < top ? left : top, mx = left < top ? top : left;
      const int64_t grad = left + top - topleft;
      const int64_t gmax = topleft < mn ? mx : grad;
      return topleft > mx ? mn : gmax;
    }
    case 7: return topright;
    case 8: return topleft;
    case 9: return leftleft;
    case 10: return (left + topleft) / 2;
    case 11: return (top + topleft) / 2;
    case 12: return (top + topright) / 2;
    case 13: return (6 * top - 2 * toptop + 7 * left + leftleft + toprightright + 3 * topright + 8) / 16;
    default: return 0; /* 6 = Weighted: not restated (its own branch of the step, palette.rs:200-227) */
  }
}

/* do_palette_step_general, the branch with delta entries and / or a predictor other than Zero and Weighted
 * (palette.rs:228-251): raster order per channel, entries below num_deltas are added to the prediction from the
 * already reconstructed neighbours */
void jxlo_palette_delta(const int32_t* index, int w, int h, const int32_t* palette, int num_colors, int num_deltas,
                        size_t palette_stride, int nb_channels, int bit_depth, int predictor, int32_t* out) {
  const size_t n = (size_t)w * h;
  for (int c = 0; c < nb_channels; c++) {
    int32_t* o = out + (size_t)c * n;
    for (int y = 0; y < h; y++) {
      for (int x = 0; x < w; x++) {
        const int32_t idx = index[(size_t)y * w + x];
        const int32_t entry = jxlo_palette_value(palette, palette_stride, (int64_t)idx, c, num_colors + num_deltas, bit_depth);
        int32_t val = entry;
        if (idx < num_deltas) val = (int32_t)(uint32_t)(uint64_t)(predict_one(predictor, o, w, x, y) + (int64_t)entry);
        o[(size_t)y * w + x] = val;
      }
    }
  }
}

/* ---- the self-correcting ("Weighted") predictor, modular/predict.rs:201-519 ----
 * State per channel: two rows of the four sub-predictors' error sums (pred_errors_buffer) and of the final
 * prediction's signed error (error), double-buffered by row parity; error[half + 0] stays 0 and position x lives at
 * error[half + x + 1].  The interface keeps the reference's two calls: predict (predict_and_property, :312-470) and
 * update (update_errors, :472-517). */
#define WP_EXTRA_BITS 3
#define WP_ROUND (((1 << WP_EXTRA_BITS) >> 1) - 1)
static const uint32_t kWpDivLookup[64] = { /* (1 << 24) / (i + 1), :206-213 */
    16777216, 8388608, 5592405, 4194304, 3355443, 2796202, 2396745, 2097152, 1864135, 1677721, 1525201, 1398101,
    1290555,  1198372, 1118481, 1048576, 986895,  932067,  883011,  838860,  798915,  762600,  729444,  699050,
    671088,   645277,  621378,  599186,  578524,  559240,  541200,  524288,  508400,  493447,  479349,  466033,
    453438,   441505,  430185,  419430,  409200,  399457,  390167,  381300,  372827,  364722,  356962,  349525,
    342392,   335544,  328965,  322638,  316551,  310689,  305040,  299593,  294337,  289262,  284359,  279620,
    275036,   270600,  266305,  262144};
struct jxlo_wp_state {
  int xsize;
  uint32_t (*pe)[4]; /* pred_errors_buffer, 2 * (xsize + 1) entries */
  int32_t* error;    /* 2 * (xsize + 1) entries */
  int64_t prediction[4];
  int64_t pred;
  uint8_t w[4], p1c, p2c, p3c[5];
};
static int wp_ilog2_u64(uint64_t v) { return 63 - __builtin_clzll(v); }
/* header: p1c, p2c, p3ca..p3ce, w0..w3 (headers/modular.rs:16-66) */
jxlo_wp_state* jxlo_wp_new(const uint32_t header[11], int xsize) {
  jxlo_wp_state* s = (jxlo_wp_state*)calloc(1, sizeof(*s));
  s->xsize = xsize;
  s->pe = calloc((size_t)2 * (xsize + 1), sizeof(*s->pe));
  s->error = calloc((size_t)2 * (xsize + 1), sizeof(int32_t));
  s->p1c = (uint8_t)header[0];
  s->p2c = (uint8_t)header[1];
  for (int i = 0; i < 5; i++) s->p3c[i] = (uint8_t)header[2 + i];
  for (int i = 0; i < 4; i++) s->w[i] = (uint8_t)header[7 + i];
  return s;
}
void jxlo_wp_free(jxlo_wp_state* s) {
  if (!s) return;
  free(s->pe);
  free(s->error);
  free(s);
}
static inline int64_t wp_abs(int64_t v) { return v < 0 ? -v : v; }
/* neighbours = top, left, topright, topleft, toptop (PredictionData); returns the prediction, *property = the
 * largest-magnitude neighbouring error */
int64_t jxlo_wp_predict(jxlo_wp_state* s, int x, int y, const int32_t nb[5], int32_t* property) {
  const int half = s->xsize + 1;
  const int cur_row = (y & 1) ? 0 : half, prev_row = (y & 1) ? half : 0;
  const int pos_ne = x + 1 < s->xsize ? x + 1 : x;
  const int pos_nw = x > 0 ? x - 1 : 0;
  const uint32_t* err_n = s->pe[prev_row + x];
  const uint32_t* err_ne = s->pe[prev_row + pos_ne];
  const uint32_t* err_nw = s->pe[prev_row + pos_nw];
  uint32_t wk[4];
  for (int k = 0; k < 4; k++) {
    const uint32_t err = err_n[k] + err_ne[k] + err_nw[k]; /* wrapping */
    int shift = wp_ilog2_u64((uint64_t)err + 1) - 5;
    if (shift < 0) shift = 0;
    const uint32_t div = kWpDivLookup[err >> shift];
    wk[k] = 4u + (((uint32_t)s->w[k] * div) >> shift);
  }
  const int64_t te_w = s->error[cur_row + x];
  const int64_t te_n = s->error[prev_row + 1 + x];
  const int64_t te_nw = s->error[prev_row + 1 + pos_nw];
  const int64_t sum_wn = te_n + te_w;
  const int64_t te_ne = s->error[prev_row + 1 + pos_ne];
  int64_t p = te_w;
  if (wp_abs(te_n) > wp_abs(p)) p = te_n;
  if (wp_abs(te_nw) > wp_abs(p)) p = te_nw;
  if (wp_abs(te_ne) > wp_abs(p)) p = te_ne;
  const int64_t n = (int64_t)nb[0] << WP_EXTRA_BITS, w = (int64_t)nb[1] << WP_EXTRA_BITS;
  const int64_t ne = (int64_t)nb[2] << WP_EXTRA_BITS, nw = (int64_t)nb[3] << WP_EXTRA_BITS;
  const int64_t nn = (int64_t)nb[4] << WP_EXTRA_BITS;
  const int64_t p0 = w + ne - n;
  const int64_t p1 = n - (((sum_wn + te_ne) * (int64_t)s->p1c) >> 5);
  const int64_t p2 = w - (((sum_wn + te_nw) * (int64_t)s->p2c) >> 5);
  const int64_t p3 = n - ((te_nw * (int64_t)s->p3c[0] + te_n * (int64_t)s->p3c[1] + te_ne * (int64_t)s->p3c[2] +
                           (nn - n) * (int64_t)s->p3c[3] + (nw - w) * (int64_t)s->p3c[4]) >> 5);
  const int log_weight = wp_ilog2_u64((uint64_t)wk[0] + wk[1] + wk[2] + wk[3]);
  const int64_t w0s = (int64_t)wk[0] >> (log_weight - 4), w1s = (int64_t)wk[1] >> (log_weight - 4);
  const int64_t w2s = (int64_t)wk[2] >> (log_weight - 4), w3s = (int64_t)wk[3] >> (log_weight - 4);
  const int64_t weight_sum = w0s + w1s + w2s + w3s;
  const int64_t sum = (weight_sum >> 1) - 1 + w0s * p0 + w1s * p1 + w2s * p2 + w3s * p3;
  int64_t pred = (sum * (int64_t)kWpDivLookup[weight_sum - 1]) >> 24;
  if (((te_n ^ te_w) | (te_n ^ te_nw)) <= 0) {
    const int64_t mx = w > (ne > n ? ne : n) ? w : (ne > n ? ne : n);
    const int64_t mn = w < (ne < n ? ne : n) ? w : (ne < n ? ne : n);
    const int64_t lo = mx < pred ? mx : pred;
    pred = mn > lo ? mn : lo;
  }
  s->prediction[0] = p0; s->prediction[1] = p1; s->prediction[2] = p2; s->prediction[3] = p3;
  s->pred = pred;
  if (property) *property = (int32_t)p;
  return (pred + WP_ROUND) >> WP_EXTRA_BITS;
}
void jxlo_wp_update(jxlo_wp_state* s, int32_t correct_val, int x, int y) {
  const int half = s->xsize + 1;
  const int cur_row = (y & 1) ? 0 : half, prev_row = (y & 1) ? half : 0;
  const int64_t val = (int64_t)correct_val << WP_EXTRA_BITS;
  s->error[cur_row + x + 1] = (int32_t)(s->pred - val);
  uint32_t e[4];
  for (int k = 0; k < 4; k++) e[k] = (uint32_t)((wp_abs(s->prediction[k] - val) + WP_ROUND) >> WP_EXTRA_BITS);
  for (int k = 0; k < 4; k++) s->pe[cur_row + x][k] = e[k];
  for (int k = 0; k < 4; k++) s->pe[prev_row + x + 1][k] += e[k];
}
/* do_palette_step_general, predictor == Weighted branch (palette.rs:200-227): every pixel runs the predictor and
 * updates its errors, delta entries (index < num_deltas) are added to the prediction */
void jxlo_palette_delta_wp(const int32_t* index, int w, int h, const int32_t* palette, int num_colors, int num_deltas,
                           size_t palette_stride, int nb_channels, int bit_depth, const uint32_t wp_header[11],
                           int32_t* out) {
  const size_t n = (size_t)w * h;
  for (int c = 0; c < nb_channels; c++) {
    int32_t* o = out + (size_t)c * n;
    jxlo_wp_state* st = jxlo_wp_new(wp_header, w);
    for (int y = 0; y < h; y++) {
      for (int x = 0; x < w; x++) {
        const int32_t idx = index[(size_t)y * w + x];
        const int32_t entry = jxlo_palette_value(palette, palette_stride, (int64_t)idx, c, num_colors + num_deltas, bit_depth);
        /* PredictionData::get (predict.rs:96-128) */
        const int32_t* row = o + (size_t)y * w;
        const int32_t* row_top = o + (size_t)(y > 0 ? y - 1 : 0) * w;
        const int32_t* row_toptop = o + (size_t)(y > 1 ? y - 2 : 0) * w;
        const int32_t left = x > 0 ? row[x - 1] : (y > 0 ? row_top[0] : 0);
        const int32_t top = y > 0 ? row_top[x] : left;
        const int32_t topleft = (x > 0 && y > 0) ? row_top[x - 1] : left;
        const int32_t topright = (x + 1 < w && y > 0) ? row_top[x + 1] : top;
        const int32_t toptop = y > 1 ? row_toptop[x] : top;
        const int32_t nb[5] = {top, left, topright, topleft, toptop};
        const int64_t pred = jxlo_wp_predict(st, x, y, nb, NULL);
        const int32_t val = idx < num_deltas ? (int32_t)(uint32_t)(uint64_t)(pred + (int64_t)entry) : entry;
        o[(size_t)y * w + x] = val;
        jxlo_wp_update(st, val, x, y);
      }
    }
    jxlo_wp_free(st);
  }
}

/* ---- squeeze ---- */
int64_t jxlo_smooth_tendency(int64_t b, int64_t a, int64_t n) { /* squeeze.rs:143-168 */
  int64_t diff = 0;
  if (b >= a && a >= n) {
    diff = (4 * b - 3 * n - a + 6) / 12;
    if (diff - (diff & 1) > 2 * (b - a)) diff = 2 * (b - a) + 1;
    if (diff + (diff & 1) > 2 * (a - n)) diff = 2 * (a - n);
  } else if (b <= a && a <= n) {
    diff = (4 * b - 3 * n - a - 6) / 12;
    if (diff + (diff & 1) < 2 * (b - a)) diff = 2 * (b - a) - 1;
    if (diff - (diff & 1) < 2 * (a - n)) diff = 2 * (a - n);
  }
  return diff;
}

/* 32-bit formulation used by the SIMD back-ends (squeeze.rs:107-141); the
 * reference requires it to equal the scalar definition wherever nothing overflows. */
int32_t jxlo_smooth_tendency_i32(int32_t a, int32_t b, int32_t c) {
  const int32_t a_b = wsub(a, b), b_c = wsub(b, c), a_c = wsub(a, c);
  const int32_t abs_a_b = a_b < 0 ? wsub(0, a_b) : a_b;
  const int32_t abs_b_c = b_c < 0 ? wsub(0, b_c) : b_c;
  const int32_t abs_a_c = a_c < 0 ? wsub(0, a_c) : a_c;
  const int non_monotonic = (a_b ^ b_c) < 0;
  int skip = (a_b != 0) && non_monotonic;   /* a_b.eq_zero().andnot(non_monotonic) */
  skip = (b_c != 0) && skip;                /* b_c.eq_zero().andnot(skip) */
  const int32_t abs_a_b_3 = (int32_t)(((int64_t)abs_a_b * 0x55555556LL) >> 32);
  int32_t x = sar(wadd(wadd(2, abs_a_c), abs_a_b_3), 2);
  const int32_t abs_a_b_2_add_x = wadd((int32_t)((uint32_t)abs_a_b << 1), x & 1);
  if (x > abs_a_b_2_add_x) x = wadd((int32_t)((uint32_t)abs_a_b << 1), 1);
  const int32_t abs_b_c_2 = (int32_t)((uint32_t)abs_b_c << 1);
  if (wadd(x, x & 1) > abs_b_c_2) x = abs_b_c_2;
  if (skip) x = 0; /* maskz */
  return a_c < 0 ? wsub(0, x) : x;
}

static inline void unsqueeze(int32_t avg, int32_t res, int32_t next_avg, int32_t prev, int32_t* a,
                             int32_t* b) { /* squeeze.rs:187-194 */
  const int64_t tendency = jxlo_smooth_tendency((int64_t)prev, (int64_t)avg, (int64_t)next_avg);
  const int64_t diff = (int64_t)res + tendency;
  const int64_t aa = (int64_t)avg + (diff / 2);
  const int64_t bb = aa - diff;
  *a = (int32_t)aa;
  *b = (int32_t)bb;
}

/* unsqueeze_impl (squeeze.rs:171-185): the 32-bit form the SIMD back-ends run on all but the remainder columns /
 * rows (hsqueeze_impl :197-290, vsqueeze_impl :483-574) -- Wrapping<i32> lanes, tendency from smooth_tendency_impl,
 * diff / 2 as (diff + sign bit) >> 1.  Equal to the scalar definition wherever nothing wraps; beyond that the
 * reference's two forms give different answers for the same input. */
static inline void unsqueeze_simd(int32_t avg, int32_t res, int32_t next_avg, int32_t prev, int32_t* a, int32_t* b) {
  const int32_t tendency = jxlo_smooth_tendency_i32(prev, avg, next_avg);
  const int32_t diff = wadd(res, tendency);
  const int32_t sign = (int32_t)((uint32_t)diff >> 31);
  const int32_t diff_2 = sar(wadd(diff, sign), 1);
  *a = wadd(avg, diff_2);
  *b = wsub(*a, diff);
}
#define unsqueeze_any(avg, res, next, prev, a, b) \
  (simd ? unsqueeze_simd(avg, res, next, prev, a, b) : unsqueeze(avg, res, next, prev, a, b))

static void unsqueeze_h_impl(const int32_t* avg, size_t avg_stride, const int32_t* res, size_t res_stride,
                             int out_w, int h, int32_t* out, size_t out_stride, int simd) {
  const int w = out_w / 2; /* residual width; avg width = out_w - w */
  if (out_w == 0 || h == 0) return;
  if (w == 0) { /* do_hsqueeze_step :468-476 */
    for (int y = 0; y < h; y++) out[(size_t)y * out_stride] = avg[(size_t)y * avg_stride];
    return;
  }
  const int has_tail = out_w & 1;
  for (int y = 0; y < h; y++) { /* hsqueeze_scalar :408-437, out_prev/in_next_avg = None */
    const int32_t* ar = avg + (size_t)y * avg_stride;
    const int32_t* rr = res + (size_t)y * res_stride;
    int32_t* o = out + (size_t)y * out_stride;
    int32_t prev_b = ar[0];
    const int x_end = has_tail ? w : w - 1;
    for (int x = 0; x < x_end; x++) {
      int32_t a, b;
      unsqueeze_any(ar[x], rr[x], ar[x + 1], prev_b, &a, &b);
      o[2 * x] = a;
      o[2 * x + 1] = b;
      prev_b = b;
    }
    if (!has_tail) {
      int32_t a, b;
      unsqueeze_any(ar[w - 1], rr[w - 1], ar[w - 1], prev_b, &a, &b);
      o[2 * w - 2] = a;
      o[2 * w - 1] = b;
    } else {
      o[2 * w] = ar[w];
    }
  }
}

static void unsqueeze_v_impl(const int32_t* avg, size_t avg_stride, const int32_t* res, size_t res_stride,
                             int w, int out_h, int32_t* out, size_t out_stride, int simd) {
  const int h = out_h / 2;
  if (out_h == 0 || w == 0) return;
  if (h == 0) { /* do_vsqueeze_step :672-675 */
    memcpy(out, avg, sizeof(int32_t) * w);
    return;
  }
  const int has_tail = out_h & 1;
  for (int y = 0; y < h; y++) { /* vsqueeze_scalar :591-640 */
    const int32_t* ar = avg + (size_t)y * avg_stride;
    const int32_t* rr = res + (size_t)y * res_stride;
    const int32_t* an = (has_tail || y < h - 1) ? avg + (size_t)(y + 1) * avg_stride : ar;
    const int32_t* pb = y == 0 ? ar : out + (size_t)(2 * y - 1) * out_stride;
    for (int x = 0; x < w; x++) {
      int32_t a, b;
      unsqueeze_any(ar[x], rr[x], an[x], pb[x], &a, &b);
      out[(size_t)(2 * y) * out_stride + x] = a;
      out[(size_t)(2 * y + 1) * out_stride + x] = b;
    }
  }
  if (has_tail) memcpy(out + (size_t)(2 * h) * out_stride, avg + (size_t)h * avg_stride, sizeof(int32_t) * w);
}

void jxlo_unsqueeze_h(const int32_t* avg, size_t avg_stride, const int32_t* res, size_t res_stride,
                      int out_w, int h, int32_t* out, size_t out_stride) {
  unsqueeze_h_impl(avg, avg_stride, res, res_stride, out_w, h, out, out_stride, 0);
}
void jxlo_unsqueeze_v(const int32_t* avg, size_t avg_stride, const int32_t* res, size_t res_stride,
                      int w, int out_h, int32_t* out, size_t out_stride) {
  unsqueeze_v_impl(avg, avg_stride, res, res_stride, w, out_h, out, out_stride, 0);
}
/* the same drivers with EVERY step in the SIMD back-ends' wrapping 32-bit form (a reference build runs it on the bulk
 * of a plane and the scalar form on the remainder columns / rows; the two agree for samples below 2^28) */
void jxlo_unsqueeze_h_simd(const int32_t* avg, size_t avg_stride, const int32_t* res, size_t res_stride,
                           int out_w, int h, int32_t* out, size_t out_stride) {
  unsqueeze_h_impl(avg, avg_stride, res, res_stride, out_w, h, out, out_stride, 1);
}
void jxlo_unsqueeze_v_simd(const int32_t* avg, size_t avg_stride, const int32_t* res, size_t res_stride,
                           int w, int out_h, int32_t* out, size_t out_stride) {
  unsqueeze_v_impl(avg, avg_stride, res, res_stride, w, out_h, out, out_stride, 1);
}

/* ---- smooth unsqueeze: what a squeeze step runs when its residual channel has not arrived (all-zero), i.e. the
 * progressive previews (transforms/step.rs:138-150 picks the kind, :841-851 dispatches).  The average channel is
 * upsampled by a 5x5 float kernel instead of the integer tendency recurrence. ---- */
#ifndef JXLO_FUSED
#define JXLO_FUSED 1
#endif
static inline float sm_mul_add(float a, float b, float c) {
#if JXLO_FUSED
  return fmaf(a, b, c);
#else
  return (a * b) + c;
#endif
}
typedef struct { uint8_t n; float w; } SmTap;
/* Four partial sums of (up to) four taps, each from zero in the listed order, joined as (a + b) + (c + d), then
 * + copysign(0.5, sum) and a truncating convert: convolve_2d_simd squeeze.rs:686-812, convolve_1d_simd :814-888.
 * n[k] is the 5x5 neighbourhood, k = 5 * row + col (the vertical variant transposes it, :1160-1188). */
#define W2 0.62646443f
#define W10 0.24413736f
#define W18 0.06118795f
#define W26 -0.01328634f
#define W34 -0.03355509f
#define W50 -0.02015225f
#define W58 -0.01033307f
#define W74 -0.00056067f
static const SmTap kSm2d[4][16] = {
    {{1, W58}, {2, W50}, {3, W74}, {5, W58}, {6, W18}, {7, W10}, {8, W34}, {10, W50}, {11, W10}, {12, W2}, {13, W26},
     {0, 0.f}, {15, W74}, {16, W34}, {17, W26}, {18, W50}},
    {{1, W74}, {2, W50}, {3, W58}, {6, W34}, {7, W10}, {8, W18}, {9, W58}, {11, W26}, {12, W2}, {13, W10}, {14, W50},
     {0, 0.f}, {16, W50}, {17, W26}, {18, W34}, {19, W74}},
    {{5, W74}, {6, W34}, {7, W26}, {8, W50}, {10, W50}, {11, W10}, {12, W2}, {13, W26}, {15, W58}, {16, W18},
     {17, W10}, {0, 0.f}, {18, W34}, {21, W58}, {22, W50}, {23, W74}},
    {{6, W50}, {7, W26}, {8, W34}, {9, W74}, {11, W26}, {12, W2}, {13, W10}, {14, W50}, {16, W34}, {17, W10},
     {18, W18}, {0, 0.f}, {19, W58}, {21, W74}, {22, W50}, {23, W58}}};
#define V1 0.69472290f
#define V9 0.27861324f
#define V17 0.07666797f
#define V25 -0.00778371f
#define V41 -0.03143468f
#define V49 -0.02150597f
#define V65 -0.00434251f
#define V73 -0.00078780f
static const SmTap kSm1d[2][16] = {
    {{1, V73}, {2, V65}, {5, V65}, {6, V25}, {7, V17}, {8, V41}, {10, V49}, {11, V9}, {12, V1}, {13, V25}, {15, V65},
     {16, V25}, {17, V17}, {18, V41}, {21, V73}, {22, V65}},
    {{2, V65}, {3, V73}, {6, V41}, {7, V17}, {8, V25}, {9, V65}, {11, V25}, {12, V1}, {13, V9}, {14, V49}, {16, V41},
     {17, V17}, {18, V25}, {19, V65}, {22, V65}, {23, V73}}};
/* a {0, 0.f} entry marks a three-tap partial sum (the 2-D kernel's third): it is skipped, not accumulated */
/* as_i32 differs between the reference's back-ends: scalar (`as i32`, jxl_simd/src/scalar.rs:178), NEON (vcvtq_s32_f32,
 * aarch64/neon.rs:400) and wasm truncate -- with the +-0.5 that is round-half-away, the evident intent -- while the
 * x86 ones use cvtps (x86_64/avx.rs:580, sse42.rs:472, avx512.rs:638), round-to-nearest-even ON TOP of the +-0.5.
 * cvt_rne = 0 restates the former (what the product implements), 1 the x86 behaviour (kept to show the difference). */
static int32_t sm_eval(const SmTap t[16], const float n[25], int cvt_rne) {
  float part[4];
  for (int g = 0; g < 4; g++) {
    float acc = 0.f;
    for (int k = 0; k < 4; k++) {
      const SmTap tp = t[4 * g + k];
      if (tp.w == 0.f) continue;
      acc = sm_mul_add(n[tp.n], tp.w, acc);
    }
    part[g] = acc;
  }
  const float sum = (part[0] + part[1]) + (part[2] + part[3]);
  const float biased = sum + copysignf(0.5f, sum);
  return cvt_rne ? (int32_t)lrintf(biased) : (int32_t)biased;
}
void jxlo_smooth_convolve_2d(const float n[25], int cvt_rne, int32_t out[4]) {
  for (int i = 0; i < 4; i++) out[i] = sm_eval(kSm2d[i], n, cvt_rne);
}
void jxlo_smooth_convolve_1d(const float n[25], int cvt_rne, int32_t out[2]) {
  for (int i = 0; i < 2; i++) out[i] = sm_eval(kSm1d[i], n, cvt_rne);
}
/* TiledChannelView::load_row_to_scratch (step.rs:372-420) seen from one sample: rows mirror without repeating the
 * edge twice ( -1 -> 0, -2 -> 1, h -> h - 1 ), columns clamp. */
static float sm_sample(const int32_t* in, size_t stride, int w, int h, int x, int y) {
  const int yy = h == 1 ? 0 : (y < 0 ? -y - 1 : (y >= h ? 2 * h - 1 - y : y));
  const int xx = x < 0 ? 0 : (x >= w ? w - 1 : x);
  return (float)in[(size_t)yy * stride + xx];
}
/* kind 0: smooth_h_unsqueeze (squeeze.rs:1010-1105), 1: smooth_v_unsqueeze (:1120-1225), 2: smooth_2d_unsqueeze
 * (:908-1003).  `in` is the whole average channel (in_w x in_h); the out_w x out_h output rectangle sits at (x0, y0)
 * of the output channel (Rect of the grid tile; (0, 0) for a whole channel). */
void jxlo_smooth_unsqueeze(int kind, const int32_t* in, size_t in_stride, int in_w, int in_h, int x0, int y0,
                           int32_t* out, size_t out_stride, int out_w, int out_h, int cvt_rne) {
  const int fx = kind != 1, fy = kind != 0; /* which axes double */
  const int in_xs = fx ? out_w / 2 : out_w, in_ys = fy ? out_h / 2 : out_h;
  if (in_xs == 0 || in_ys == 0) return;
  const int cx0 = fx ? x0 / 2 : x0, cy0 = fy ? y0 / 2 : y0;
  const int ny = fy ? (out_h + 1) / 2 : out_h, nx = fx ? (out_w + 1) / 2 : out_w;
  for (int iy = 0; iy < ny; iy++) {
    for (int ix = 0; ix < nx; ix++) {
      float n[25];
      for (int r = 0; r < 5; r++)
        for (int c = 0; c < 5; c++) {
          const float v = sm_sample(in, in_stride, in_w, in_h, cx0 + ix + c - 2, cy0 + iy + r - 2);
          n[kind == 1 ? 5 * c + r : 5 * r + c] = v;
        }
      if (kind == 2) {
        int32_t o[4];
        jxlo_smooth_convolve_2d(n, cvt_rne, o);
        for (int k = 0; k < 4; k++) {
          const int ox = 2 * ix + (k & 1), oy = 2 * iy + (k >> 1);
          if (ox < out_w && oy < out_h) out[(size_t)oy * out_stride + ox] = o[k];
        }
      } else {
        int32_t o[2];
        jxlo_smooth_convolve_1d(n, cvt_rne, o);
        for (int k = 0; k < 2; k++) {
          const int ox = kind == 0 ? 2 * ix + k : ix, oy = kind == 0 ? iy : 2 * iy + k;
          if (ox < out_w && oy < out_h) out[(size_t)oy * out_stride + ox] = o[k];
        }
      }
    }
  }
}

/* ---- the stages between Modular channels and the rest of the pipeline (render/stages/convert.rs) ---- */
/* ConvertI32ToU8Stage (:642-715; the builder's replacement for ModularToF32 + F32ToU8 when the output depth is a
 * multiple of the channel's, render/builder.rs:152-170): wrapping multiply, clamp to [0, max], low byte */
void jxlo_i32_to_u8(const int32_t* in, size_t n, int32_t multiplier, int32_t max, uint8_t* out) {
  for (size_t i = 0; i < n; i++) {
    const int32_t scaled = (int32_t)((uint32_t)in[i] * (uint32_t)multiplier);
    const int32_t zeroclip = scaled < 0 ? 0 : scaled;
    const int32_t clip = scaled > max ? max : zeroclip;
    out[i] = (uint8_t)clip;
  }
}
/* ConvertModularToF32Stage, integer samples (:488-533): val * (1 / (2^bits - 1)) */
void jxlo_modular_to_f32(const int32_t* in, size_t n, int bits, float* out) {
  const float scale = 1.0f / (float)((1ull << bits) - 1);
  for (size_t i = 0; i < n; i++) out[i] = (float)in[i] * scale;
}
/* ConvertModularToF32Stage, floating-point samples: int_to_float_generic (:436-486), which also reproduces the two
 * fast paths of int_to_float (:416-433) -- binary32 bit for bit, binary16 like the hardware conversion except that a
 * signalling NaN keeps its payload */
void jxlo_float_samples_to_f32(const int32_t* in, size_t n, uint32_t bits, uint32_t exp_bits, float* out) {
  const int exp_bias = (1 << (exp_bits - 1)) - 1;
  const uint32_t sign_shift = bits - 1, mant_bits = bits - exp_bits - 1, mant_shift = 23 - mant_bits;
  for (size_t i = 0; i < n; i++) {
    uint32_t f = (uint32_t)in[i], r;
    const int signbit = (f >> sign_shift) != 0;
    f &= (uint32_t)((1ull << sign_shift) - 1);
    if (f == 0) {
      r = signbit ? 0x80000000u : 0u;
    } else {
      int exp = (int)(f >> mant_bits);
      uint32_t mantissa = f & ((1u << mant_bits) - 1u);
      if (exp == (1 << exp_bits) - 1) {
        r = (signbit ? 0x80000000u : 0u) | 0xffu << 23 | mantissa << mant_shift;
      } else {
        mantissa <<= mant_shift;
        if (exp == 0 && exp_bits < 8) {
          while ((mantissa & 0x800000u) == 0) {
            mantissa <<= 1;
            exp -= 1;
          }
          exp += 1;
          mantissa &= 0x7fffffu;
        }
        exp -= exp_bias;
        exp += 127;
        r = (signbit ? 0x80000000u : 0u) | (uint32_t)exp << 23 | mantissa;
      }
    }
    memcpy(&out[i], &r, 4);
  }
}
/* ConvertModularXYBToF32Stage (:306-343): channels arrive as Y, X, B; B carries B - Y */
void jxlo_modular_xyb_to_f32(const int32_t* y, const int32_t* x, const int32_t* b, size_t n, const float scale[3],
                             float* ox, float* oy, float* ob) {
  for (size_t i = 0; i < n; i++) {
    ox[i] = (float)x[i] * scale[0];
    oy[i] = (float)y[i] * scale[1];
    ob[i] = ((float)b[i] + (float)y[i]) * scale[2];
  }
}
