/*
 * ORACLE (test infrastructure, see jxlo.h) -- restatement of jxl_transforms.
 *
 * Reference map:
 *   1-D IDCT recursion ......... jxl_transforms/src/idct_large.rs:251-310 (generic form);
 *                                idct2.rs:17-25, idct4.rs:17-37, idct8.rs:17-77 are the same
 *                                recursion unrolled (checked operation by operation)
 *   2-D IDCT drivers ........... idct2d.rs:11-426, idct_large.rs:387-537
 *   reinterpreting DCT ......... reinterpreting_dct{2,4,8,16,32}.rs, reinterpreting_dct2d.rs
 *   transform_to_pixels ........ transform.rs:14-32 (idct2_top_block), :295-374 (AFV),
 *                                :377-664 (27-way switch)
 *   f64 definitions ............ tests.rs:24-173
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "jxlo.h"

#ifndef JXLO_FUSED
#define JXLO_FUSED 1
#endif

int jxlo_is_fused(void) { return JXLO_FUSED; }

/* mul_add / neg_mul_add of the SIMD layer.  `fused` is a per-call-site property:
 * with JXLO_FUSED=1 we model the x86 AVX2 back-end, where shapes routed through
 * `ScalarDescriptor` or `maybe_downgrade_128bit()` (SSE4.2: sse42.rs:403-409
 * computes this*mul+add unfused) stay unfused (idct2d.rs:341-366,
 * reinterpreting_dct2d.rs:535-600). */
static inline float mul_add(float a, float b, float c, int fused) {
#if JXLO_FUSED
  if (fused) return fmaf(a, b, c);
#else
  (void)fused;
#endif
  return (a * b) + c;
}
static inline float neg_mul_add(float a, float b, float c, int fused) {
#if JXLO_FUSED
  if (fused) return fmaf(-a, b, c);
#else
  (void)fused;
#endif
  return c - (a * b);
}

/* ------------------------------------------------------------------------- */
/* transform table: transform_map.rs:97-116                                   */
static const int kCoveredX[JXLO_NUM_TRANSFORMS] = {1, 1, 1, 1, 2, 4, 1, 2,  1,  4, 2,  4,  1, 1,
                                                   1, 1, 1, 1, 8, 4, 8, 16, 8, 16, 32, 16, 32};
static const int kCoveredY[JXLO_NUM_TRANSFORMS] = {1, 1, 1, 1, 2, 4, 2,  1,  4, 1, 4,  2,  1, 1,
                                                   1, 1, 1, 1, 8, 8, 4, 16, 16, 8, 32, 32, 16};
int jxlo_covered_blocks_x(int t) { return kCoveredX[t]; }
int jxlo_covered_blocks_y(int t) { return kCoveredY[t]; }

/* ------------------------------------------------------------------------- */
/* constant tables                                                            */
static float g_w[9][128];      /* g_w[log2 n][i] = 1/(2 cos((2i+1) pi / 2n)) */
static float g_scale[6][32];   /* g_scale[log2 n][i], 6-decimal constants of the reference */
static int g_init = 0;

static int ilog2(int n) {
  int l = 0;
  while ((1 << l) < n) l++;
  return l;
}

static void init_tables(void) {
  if (g_init) return;
  for (int l = 2; l <= 8; l++) {
    int n = 1 << l;
    for (int i = 0; i < n / 2; i++) {
      g_w[l][i] = (float)(1.0 / (2.0 * cos((2.0 * i + 1.0) * M_PI / (2.0 * n))));
    }
  }
  /* reinterpreting_dctN.rs return tuples: 1/(n cos(i pi/16n) cos(i pi/8n) cos(i pi/4n)),
   * written in the reference with 6 decimals (e.g. reinterpreting_dct4.rs:35-38) */
  for (int l = 0; l <= 5; l++) {
    int n = 1 << l;
    for (int i = 0; i < n; i++) {
      double s = 1.0 / (n * cos(i * M_PI / (16.0 * n)) * cos(i * M_PI / (8.0 * n)) *
                        cos(i * M_PI / (4.0 * n)));
      char buf[64];
      snprintf(buf, sizeof buf, "%.6f", s);
      g_scale[l][i] = strtof(buf, NULL);
    }
  }
  g_init = 1;
}

const float* jxlo_idct_weights(int n) {
  init_tables();
  return g_w[ilog2(n)];
}
const float* jxlo_rdct_scales(int n) {
  init_tables();
  return g_scale[ilog2(n)];
}

/* ------------------------------------------------------------------------- */
/* 1-D IDCT, recursive even/odd split (idct_large.rs:284-309).  x has n entries,
 * scratch has n entries.                                                      */
static void idct_rec(float* x, int n, float* scratch, int fused) {
  if (n == 2) { /* idct2.rs:17-25 */
    float a = x[0] + x[1];
    float b = x[0] - x[1];
    x[0] = a;
    x[1] = b;
    return;
  }
  const int h = n / 2;
  float* even = scratch;
  float* odd = scratch + h;
  for (int i = 0; i < h; i++) {
    even[i] = x[2 * i];
    odd[i] = x[2 * i + 1];
  }
  idct_rec(even, h, x, fused);
  for (int i = h - 1; i >= 1; i--) odd[i] += odd[i - 1];
  odd[0] *= (float)M_SQRT2;
  idct_rec(odd, h, x, fused);
  const float* w = g_w[ilog2(n)];
  for (int i = 0; i < h; i++) {
    x[i] = mul_add(odd[i], w[i], even[i], fused);
    x[n - 1 - i] = neg_mul_add(odd[i], w[i], even[i], fused);
  }
}

static void idct1d_impl(float* data, int n, int stride, int fused) {
  float x[256], scratch[256 + 128 + 64 + 32 + 16 + 8 + 4 + 2];
  /* scratch for the recursion: each level uses its caller's x as scratch, so two
   * buffers of n suffice (as in idct_impl_inner(first_half, data)). */
  for (int i = 0; i < n; i++) x[i] = data[(size_t)i * stride];
  idct_rec(x, n, scratch, fused);
  for (int i = 0; i < n; i++) data[(size_t)i * stride] = x[i];
}

void jxlo_idct1d(float* data, int n, int stride) {
  init_tables();
  idct1d_impl(data, n, stride, 1);
}

/* 1-D forward "reinterpreting" DCT (reinterpreting_dct8.rs:10-100 is the n=8
 * instance; the structure below reproduces v8..v44 of that listing for n=8 and
 * the 2/4/16/32 listings likewise):
 *   even = DCT(x[i] + x[n-1-i]);  odd = DCT((x[i] - x[n-1-i]) * w_n[i]);
 *   odd[0] = odd[0]*sqrt2 + odd[1];  odd[i] += odd[i+1];  interleave.
 * The top-level n==2 case has no w multiply (reinterpreting_dct2.rs:13-24).    */
static void rdct_rec(float* x, int n, float* scratch, int fused) {
  if (n == 1) return;
  if (n == 2) {
    float a = x[0] + x[1];
    float b = x[0] - x[1];
    x[0] = a;
    x[1] = b;
    return;
  }
  const int h = n / 2;
  float* even = scratch;
  float* odd = scratch + h;
  const float* w = g_w[ilog2(n)];
  for (int i = 0; i < h; i++) even[i] = x[i] + x[n - 1 - i];
  for (int i = 0; i < h; i++) odd[i] = (x[i] - x[n - 1 - i]) * w[i];
  rdct_rec(even, h, x, fused);
  rdct_rec(odd, h, x, fused);
  odd[0] = mul_add(odd[0], (float)M_SQRT2, odd[1], fused);
  for (int i = 1; i + 1 < h; i++) odd[i] = odd[i] + odd[i + 1];
  for (int i = 0; i < h; i++) {
    x[2 * i] = even[i];
    x[2 * i + 1] = odd[i];
  }
}

static void rdct1d_impl(float* data, int n, int stride, int fused) {
  float x[32], scratch[64];
  for (int i = 0; i < n; i++) x[i] = data[(size_t)i * stride];
  rdct_rec(x, n, scratch, fused);
  const float* s = g_scale[ilog2(n)];
  for (int i = 0; i < n; i++) data[(size_t)i * stride] = x[i] * s[i];
}

void jxlo_rdct1d(float* data, int n, int stride) {
  init_tables();
  rdct1d_impl(data, n, stride, 1);
}

/* ------------------------------------------------------------------------- */
/* 2-D IDCT.  Semantics of idct2d_square / _wide / _thin (idct_large.rs:387-501),
 * identical for the <=32 drivers in idct2d.rs: the horizontal (u -> x) transform
 * runs first, then the vertical (v -> y) one; SIMD transposes are layout only.  */
void jxlo_idct2d(float* data, int rows, int cols) {
  init_tables();
  const int fused = (rows > 4 && cols > 4); /* idct2d.rs:341-366: 2x2 scalar, 4x4/4x8/8x4 128-bit */
  if (rows < cols) {
    /* in[v*cols + u] */
    for (int v = 0; v < rows; v++) idct1d_impl(data + (size_t)v * cols, cols, 1, fused);
    for (int x = 0; x < cols; x++) idct1d_impl(data + x, rows, cols, fused);
    return;
  }
  /* in[u*rows + v] (transposed) */
  for (int v = 0; v < rows; v++) idct1d_impl(data + v, cols, rows, fused);
  float* t = (float*)malloc(sizeof(float) * (size_t)rows * cols);
  for (int x = 0; x < cols; x++)
    for (int v = 0; v < rows; v++) t[(size_t)v * cols + x] = data[(size_t)x * rows + v];
  for (int x = 0; x < cols; x++) idct1d_impl(t + x, rows, cols, fused);
  memcpy(data, t, sizeof(float) * (size_t)rows * cols);
  free(t);
}

/* LLF from LF (reinterpreting_dct2d.rs).  wide: rows (horizontal) first, then
 * columns (:110-136); square/thin: columns (vertical) first, then the other
 * axis, result left transposed (:140-213).                                    */
void jxlo_rdct2d(float* lf, int rows, int cols, float* out) {
  init_tables();
  const int mn = rows < cols ? rows : cols;
  const int mx = rows < cols ? cols : rows;
  const int fused = mn > 4; /* reinterpreting_dct2d.rs:535-600 */
  const size_t ostride = (size_t)8 * mx;
  if (rows < cols) {
    for (int y = 0; y < rows; y++) rdct1d_impl(lf + (size_t)y * cols, cols, 1, fused);
    for (int x = 0; x < cols; x++) rdct1d_impl(lf + x, rows, cols, fused);
    for (int y = 0; y < rows; y++)
      for (int x = 0; x < cols; x++) out[y * ostride + x] = lf[(size_t)y * cols + x];
    return;
  }
  for (int x = 0; x < cols; x++) rdct1d_impl(lf + x, rows, cols, fused);
  float t[32 * 32];
  for (int y = 0; y < rows; y++)
    for (int x = 0; x < cols; x++) t[(size_t)x * rows + y] = lf[(size_t)y * cols + x];
  /* t is cols x rows; transform along its first index */
  for (int v = 0; v < rows; v++) rdct1d_impl(t + v, cols, rows, fused);
  for (int u = 0; u < cols; u++)
    for (int v = 0; v < rows; v++) out[u * ostride + v] = t[(size_t)u * rows + v];
}

/* ------------------------------------------------------------------------- */
/* 8x8 special transforms                                                      */
static void idct2_top_block(int s, const float* in, float* out) { /* transform.rs:14-32 */
  const int num = s / 2;
  for (int y = 0; y < num; y++) {
    for (int x = 0; x < num; x++) {
      float c00 = in[y * 8 + x];
      float c01 = in[y * 8 + num + x];
      float c10 = in[(y + num) * 8 + x];
      float c11 = in[(y + num) * 8 + num + x];
      float r00 = c00 + c01 + c10 + c11;
      float r01 = c00 + c01 - c10 - c11;
      float r10 = c00 - c01 + c10 - c11;
      float r11 = c00 - c01 - c10 + c11;
      out[y * 2 * 8 + x * 2] = r00;
      out[y * 2 * 8 + x * 2 + 1] = r01;
      out[(y * 2 + 1) * 8 + x * 2] = r10;
      out[(y * 2 + 1) * 8 + x * 2 + 1] = r11;
    }
  }
}

static const float kAfvBasis[256] = {
#include "afv_basis.inc"
};

static void afv_idct4x4(const float* coeffs, float* pixels) { /* transform.rs:295-303 */
  for (int i = 0; i < 16; i++) {
    float pixel = 0.0f;
    for (int j = 0; j < 16; j++) pixel += coeffs[j] * kAfvBasis[j * 16 + i];
    pixels[i] = pixel;
  }
}

static void afv_transform(int kind, const float* c, float* pixels) { /* transform.rs:306-374 */
  const int afv_x = kind & 1, afv_y = kind / 2;
  const float block00 = c[0], block01 = c[1], block10 = c[8];
  const float dcs[3] = {(block00 + block10 + block01) * 4.0f, block00 + block10 - block01,
                        block00 - block10};
  float coeff[16], block[32];
  for (int iy = 0; iy < 4; iy++)
    for (int ix = 0; ix < 4; ix++)
      coeff[iy * 4 + ix] = (ix == 0 && iy == 0) ? dcs[0] : c[iy * 2 * 8 + ix * 2];
  afv_idct4x4(coeff, block);
  for (int iy = 0; iy < 4; iy++) {
    const int by = afv_y == 1 ? 3 - iy : iy;
    for (int ix = 0; ix < 4; ix++) {
      const int bx = afv_x == 1 ? 3 - ix : ix;
      pixels[(iy + afv_y * 4) * 8 + afv_x * 4 + ix] = block[by * 4 + bx];
    }
  }
  for (int iy = 0; iy < 4; iy++)
    for (int ix = 0; ix < 4; ix++)
      block[iy * 4 + ix] = (ix == 0 && iy == 0) ? dcs[1] : c[iy * 2 * 8 + ix * 2 + 1];
  jxlo_idct2d(block, 4, 4);
  for (int iy = 0; iy < 4; iy++)
    for (int ix = 0; ix < 4; ix++)
      pixels[(iy + afv_y * 4) * 8 + (1 - afv_x) * 4 + ix] = block[iy * 4 + ix];
  for (int iy = 0; iy < 4; iy++)
    for (int ix = 0; ix < 8; ix++)
      block[iy * 8 + ix] = (ix == 0 && iy == 0) ? dcs[2] : c[(1 + iy * 2) * 8 + ix];
  jxlo_idct2d(block, 4, 8);
  for (int iy = 0; iy < 4; iy++)
    for (int ix = 0; ix < 8; ix++) pixels[(iy + (1 - afv_y) * 4) * 8 + ix] = block[iy * 8 + ix];
}

void jxlo_transform_to_pixels(int type, float* lf, float* buf) { /* transform.rs:377-664 */
  init_tables();
  const int cx = kCoveredX[type], cy = kCoveredY[type];
  float c[64];
  switch (type) {
    case 0: /* DCT */
      buf[0] = lf[0];
      jxlo_idct2d(buf, 8, 8);
      return;
    case 1: { /* IDENTITY :530-571 */
      buf[0] = lf[0];
      memcpy(c, buf, sizeof c);
      const float b00 = c[0], b01 = c[1], b10 = c[8], b11 = c[9];
      const float dcs[4] = {b00 + b01 + b10 + b11, b00 + b01 - b10 - b11, b00 - b01 + b10 - b11,
                            b00 - b01 - b10 + b11};
      for (int y = 0; y < 2; y++) {
        for (int x = 0; x < 2; x++) {
          const float block_dc = dcs[y * 2 + x];
          float residual_sum = 0.0f;
          for (int iy = 0; iy < 4; iy++)
            for (int ix = 0; ix < 4; ix++) {
              if (ix == 0 && iy == 0) continue;
              residual_sum += c[(y + iy * 2) * 8 + x + ix * 2];
            }
          const int pivot = (4 * y + 1) * 8 + 4 * x + 1;
          buf[pivot] = block_dc - residual_sum * (1.0f / 16.0f);
          for (int iy = 0; iy < 4; iy++)
            for (int ix = 0; ix < 4; ix++) {
              if (ix == 1 && iy == 1) continue;
              buf[(y * 4 + iy) * 8 + x * 4 + ix] = c[(y + iy * 2) * 8 + x + ix * 2] + buf[pivot];
            }
          buf[y * 4 * 8 + x * 4] = c[(y + 2) * 8 + x + 2] + buf[pivot];
        }
      }
      return;
    }
    case 2: { /* DCT2X2 :572-578 */
      buf[0] = lf[0];
      memcpy(c, buf, sizeof c);
      idct2_top_block(2, c, buf);
      idct2_top_block(4, buf, c);
      idct2_top_block(8, c, buf);
      return;
    }
    case 3: { /* DCT4X4 :579-612 */
      buf[0] = lf[0];
      memcpy(c, buf, sizeof c);
      const float b00 = c[0], b01 = c[1], b10 = c[8], b11 = c[9];
      const float dcs[4] = {b00 + b01 + b10 + b11, b00 + b01 - b10 - b11, b00 - b01 + b10 - b11,
                            b00 - b01 - b10 + b11};
      for (int y = 0; y < 2; y++)
        for (int x = 0; x < 2; x++) {
          float block[16];
          block[0] = dcs[y * 2 + x];
          for (int iy = 0; iy < 4; iy++)
            for (int ix = 0; ix < 4; ix++) {
              if (ix == 0 && iy == 0) continue;
              block[iy * 4 + ix] = c[(y + iy * 2) * 8 + x + ix * 2];
            }
          jxlo_idct2d(block, 4, 4);
          for (int iy = 0; iy < 4; iy++)
            for (int ix = 0; ix < 4; ix++) buf[(y * 4 + iy) * 8 + x * 4 + ix] = block[iy * 4 + ix];
        }
      return;
    }
    case 13: { /* DCT8X4 :613-637 */
      buf[0] = lf[0];
      memcpy(c, buf, sizeof c);
      const float dcs[2] = {c[0] + c[8], c[0] - c[8]};
      for (int x = 0; x < 2; x++) {
        float block[32];
        for (int iy = 0; iy < 4; iy++)
          for (int ix = 0; ix < 8; ix++)
            block[iy * 8 + ix] = (ix == 0 && iy == 0) ? dcs[x] : c[(x + iy * 2) * 8 + ix];
        jxlo_idct2d(block, 8, 4);
        for (int iy = 0; iy < 8; iy++)
          for (int ix = 0; ix < 4; ix++) buf[iy * 8 + x * 4 + ix] = block[iy * 4 + ix];
      }
      return;
    }
    case 12: { /* DCT4X8 :638-662 */
      buf[0] = lf[0];
      memcpy(c, buf, sizeof c);
      const float dcs[2] = {c[0] + c[8], c[0] - c[8]};
      for (int y = 0; y < 2; y++) {
        float block[32];
        for (int iy = 0; iy < 4; iy++)
          for (int ix = 0; ix < 8; ix++)
            block[iy * 8 + ix] = (ix == 0 && iy == 0) ? dcs[y] : c[(y + iy * 2) * 8 + ix];
        jxlo_idct2d(block, 4, 8);
        for (int iy = 0; iy < 4; iy++)
          for (int ix = 0; ix < 8; ix++) buf[(y * 4 + iy) * 8 + ix] = block[iy * 8 + ix];
      }
      return;
    }
    case 14:
    case 15:
    case 16:
    case 17: /* AFV0..3 :510-529 */
      buf[0] = lf[0];
      memcpy(c, buf, sizeof c);
      afv_transform(type - 14, c, buf);
      return;
    default:
      /* all remaining types: LLF from LF, then the R x C IDCT (:391-509) */
      jxlo_rdct2d(lf, cy, cx, buf);
      jxlo_idct2d(buf, cy * 8, cx * 8);
      return;
  }
}

/* ------------------------------------------------------------------------- */
/* f64 definitions (tests.rs:24-173)                                           */
static double alpha(int u) { return u == 0 ? M_SQRT1_2 : 1.0; }

void jxlo_slow_dct1d(const double* in, int n, double* out) { /* tests.rs:24-61, one column */
  for (int u = 0; u < n; u++) {
    double sum = 0.0;
    for (int y = 0; y < n; y++)
      sum += alpha(u) * cos((y + 0.5) * u * M_PI / n) * M_SQRT2 * in[y];
    out[u] = sum;
  }
}
void jxlo_slow_idct1d(const double* in, int n, double* out) { /* tests.rs:63-102 */
  for (int y = 0; y < n; y++) {
    double sum = 0.0;
    for (int u = 0; u < n; u++)
      sum += alpha(u) * cos((y + 0.5) * u * M_PI / n) * M_SQRT2 * in[u];
    out[y] = sum;
  }
}

/* transform along the first index of an r x c row-major matrix */
static void apply_cols(const double* in, int r, int c, double* out,
                       void (*f)(const double*, int, double*)) {
  double* a = (double*)malloc(sizeof(double) * r);
  double* b = (double*)malloc(sizeof(double) * r);
  for (int x = 0; x < c; x++) {
    for (int y = 0; y < r; y++) a[y] = in[(size_t)y * c + x];
    f(a, r, b);
    for (int y = 0; y < r; y++) out[(size_t)y * c + x] = b[y];
  }
  free(a);
  free(b);
}
static void transpose_d(const double* in, int r, int c, double* out) {
  for (int y = 0; y < r; y++)
    for (int x = 0; x < c; x++) out[(size_t)x * r + y] = in[(size_t)y * c + x];
}

void jxlo_slow_idct2d(const double* in, int rows, int cols, double* out) { /* tests.rs:119-132 */
  size_t n = (size_t)rows * cols;
  double* a = (double*)malloc(sizeof(double) * n);
  double* b = (double*)malloc(sizeof(double) * n);
  if (rows < cols) {
    transpose_d(in, rows, cols, a); /* cols x rows */
  } else {
    memcpy(a, in, sizeof(double) * n); /* reinterpreted as cols x rows chunks of `rows` */
  }
  apply_cols(a, cols, rows, b, jxlo_slow_idct1d); /* idct along the size-`cols` index */
  transpose_d(b, cols, rows, a);                  /* rows x cols */
  apply_cols(a, rows, cols, out, jxlo_slow_idct1d);
  free(a);
  free(b);
}

static void slow_scales(int n, double* s) { /* tests.rs:134-143 */
  for (int i = 0; i < n; i++)
    s[i] = cos((double)i / (16 * n) * M_PI) * cos((double)i / (8 * n) * M_PI) *
           cos((double)i / (4 * n) * M_PI) * n;
}

void jxlo_slow_rdct2d(const double* in, int rows, int cols, double* out) { /* tests.rs:145-173 */
  size_t n = (size_t)rows * cols;
  double* a = (double*)malloc(sizeof(double) * n);
  double* b = (double*)malloc(sizeof(double) * n);
  double rs[32], cs[32];
  apply_cols(in, rows, cols, a, jxlo_slow_dct1d); /* dct1: rows x cols */
  transpose_d(a, rows, cols, b);                  /* cols x rows */
  apply_cols(b, cols, rows, a, jxlo_slow_dct1d);  /* dct2: cols x rows */
  slow_scales(rows, rs);
  slow_scales(cols, cs);
  if (rows < cols) {
    transpose_d(a, cols, rows, out); /* rows x cols */
    for (int y = 0; y < rows; y++)
      for (int x = 0; x < cols; x++) out[(size_t)y * cols + x] /= rs[y] * cs[x];
  } else {
    for (int y = 0; y < cols; y++)
      for (int x = 0; x < rows; x++) out[(size_t)y * rows + x] = a[(size_t)y * rows + x] / (rs[x] * cs[y]);
  }
  free(a);
  free(b);
}
