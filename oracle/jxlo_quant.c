/*
 * ORACLE (test infrastructure, see jxlo.h) -- default dequantisation matrices and
 * the natural coefficient order.  These are host-computed *inputs* of the hot
 * path; restated so that the synthetic workloads
 * use the same tables the reference would, and pinned by the 891 libjxl samples
 * of jxl/src/frame/quant_weights.rs:1231-2137.
 *
 * Reference map: quant_weights.rs:378-856 (library parameters, spec data),
 * :894-1079 (compute_table), :1138-1200 (get_quant_weights, interpolate, mult),
 * :321-343 (type -> table), :1128-1132 (table sizes); coeff_order.rs:66-120.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "jxlo.h"

/* quant_weights.rs:1128-1132 */
static const int kReqX[JXLO_NUM_QUANT_TABLES] = {1, 1, 1, 1, 2, 4, 1, 1, 2, 1, 1, 8, 4, 16, 8, 32, 16};
static const int kReqY[JXLO_NUM_QUANT_TABLES] = {1, 1, 1, 1, 2, 4, 2, 4, 4, 1, 1, 8, 8, 16, 16, 32, 32};
/* quant_weights.rs:321-343 */
static const int kTableForType[JXLO_NUM_TRANSFORMS] = {0, 1, 2,  3,  4,  5,  6,  6,  7,  7,  8,  8, 9, 9,
                                                       10, 10, 10, 10, 11, 12, 12, 13, 14, 14, 15, 16, 16};

int jxlo_quant_table_for_type(int type) { return kTableForType[type]; }
int jxlo_quant_table_size(int t) { return kReqX[t] * kReqY[t] * 64; }

#define MAX_BANDS 17
typedef struct {
  int num_bands;
  float params[3][MAX_BANDS];
} DctParams;

static float mult(float v) { return v > 0.0f ? 1.0f + v : 1.0f / (1.0f - v); } /* :1194-1200 */

static float interpolate_vec(float scaled_pos, const float* array) { /* :1177-1184 */
  float idxf = floorf(scaled_pos);
  float frac = scaled_pos - idxf;
  int idx = (int)idxf;
  float a = array[idx];
  float b = array[idx + 1];
  return powf(b / a, frac) * a;
}

static float interpolate(float pos, float max, const float* array, int len) { /* :1186-1192 */
  float scaled_pos = pos * (float)(len - 1) / max;
  int idx = (int)scaled_pos;
  float a = array[idx];
  float b = array[idx + 1];
  return a * powf(b / a, scaled_pos - (float)idx);
}

static void get_quant_weights(int rows, int cols, const DctParams* p, float* out) { /* :1138-1175 */
  for (int c = 0; c < 3; c++) {
    float bands[MAX_BANDS] = {0};
    bands[0] = p->params[c][0];
    for (int i = 1; i < p->num_bands; i++) bands[i] = bands[i - 1] * mult(p->params[c][i]);
    float scale = (float)(p->num_bands - 1) / ((float)M_SQRT2 + 1e-6f);
    float rcpcol = scale / (float)(cols - 1);
    float rcprow = scale / (float)(rows - 1);
    for (int y = 0; y < rows; y++) {
      float dy = (float)y * rcprow;
      float dy2 = dy * dy;
      for (int x = 0; x < cols; x++) {
        float dx = (float)x * rcpcol;
        float scaled_distance = sqrtf(dx * dx + dy2);
        float weight = p->num_bands == 1 ? bands[0] : interpolate_vec(scaled_distance, bands);
        out[(size_t)c * cols * rows + (size_t)y * cols + x] = weight;
      }
    }
  }
}

/* ---- library parameters (spec data; quant_weights.rs:378-856).  Literals are
 * written as doubles and narrowed once, like Rust's f32 literal parsing;
 * products such as 0.9 * 26629.07 are evaluated in f32 as in the reference. ---- */
#define F(x) ((float)(x))
static DctParams mk(int n, const double a[3][MAX_BANDS]) {
  DctParams p;
  memset(&p, 0, sizeof p);
  p.num_bands = n;
  for (int c = 0; c < 3; c++)
    for (int i = 0; i < n; i++) p.params[c][i] = F(a[c][i]);
  return p;
}

static DctParams params_dct8(void) {
  static const double a[3][MAX_BANDS] = {{3150.0, 0.0, -0.4, -0.4, -0.4, -2.0},
                                         {560.0, 0.0, -0.3, -0.3, -0.3, -0.3},
                                         {512.0, -2.0, -1.0, 0.0, -1.0, -2.0}};
  return mk(6, a);
}
static DctParams params_dct4x4(void) {
  static const double a[3][MAX_BANDS] = {
      {2200.0, 0.0, 0.0, 0.0}, {392.0, 0.0, 0.0, 0.0}, {112.0, -0.25, -0.25, -0.5}};
  return mk(4, a);
}
static DctParams params_dct16(void) {
  static const double a[3][MAX_BANDS] = {
      {8996.8725711814115328, -1.3000777393353804, -0.49424529824571225, -0.439093774457103443,
       -0.6350101832695744, -0.90177264050827612, -1.6162099239887414},
      {3191.48366296844234752, -0.67424582104194355, -0.80745813428471001, -0.44925837484843441,
       -0.35865440981033403, -0.31322389111877305, -0.37615025315725483},
      {1157.50408145487200256, -2.0531423165804414, -1.4, -0.50687130033378396,
       -0.42708730624733904, -1.4856834539296244, -4.9209142884401604}};
  return mk(7, a);
}
static DctParams params_dct32(void) {
  static const double a[3][MAX_BANDS] = {
      {15718.40830982518931456, -1.025, -0.98, -0.9012, -0.4, -0.48819395464, -0.421064, -0.27},
      {7305.7636810695983104, -0.8041958212306401, -0.7633036457487539, -0.55660379990111464,
       -0.49785304658857626, -0.43699592683512467, -0.40180866526242109, -0.27321683125358037},
      {3803.53173721215041536, -3.060733579805728, -2.0413270132490346, -2.0235650159727417,
       -0.5495389509954993, -0.4, -0.4, -0.3}};
  return mk(8, a);
}
static DctParams params_dct8x16(void) {
  static const double a[3][MAX_BANDS] = {{7240.7734393502, -0.7, -0.7, -0.2, -0.2, -0.2, -0.5},
                                         {1448.15468787004, -0.5, -0.5, -0.5, -0.2, -0.2, -0.2},
                                         {506.854140754517, -1.4, -0.2, -0.5, -0.5, -1.5, -3.6}};
  return mk(7, a);
}
static DctParams params_dct8x32(void) {
  static const double a[3][MAX_BANDS] = {
      {16283.2494710648897, -1.7812845336559429, -1.6309059012653515, -1.0382179034313539, -0.85,
       -0.7, -0.9, -1.2360638576849587},
      {5089.15750884921511936, -0.320049391452786891, -0.35362849922161446, -0.30340000000000003,
       -0.61, -0.5, -0.5, -0.6},
      {3397.77603275308720128, -0.321327362693153371, -0.34507619223117997, -0.70340000000000003,
       -0.9, -1.0, -1.0, -1.1754605576265209}};
  return mk(8, a);
}
static DctParams params_dct16x32(void) {
  static const double a[3][MAX_BANDS] = {
      {13844.97076442300573, -0.97113799999999995, -0.658, -0.42026, -0.22712, -0.2206, -0.226,
       -0.6},
      {4798.964084220744293, -0.61125308982767057, -0.83770786552491361, -0.79014862079498627,
       -0.2692727459704829, -0.38272769465388551, -0.22924222653091453, -0.20719098826199578},
      {1807.236946760964614, -1.2, -1.2, -0.7, -0.7, -0.7, -0.4, -0.5}};
  return mk(8, a);
}
static DctParams params_dct4x8(void) {
  static const double a[3][MAX_BANDS] = {
      {2198.050556016380522, -0.96269623020744692, -0.76194253026666783, -0.6551140670773547},
      {764.3655248643528689, -0.92630200888366945, -0.9675229603596517, -0.27845290869168118},
      {527.107573587542228, -1.4594385811273854, -1.450082094097871593, -1.5843722511996204}};
  return mk(4, a);
}
/* 64x64-and-up family: first band = mulf * basef (f32 product), rest shared (:635-856) */
static DctParams params_large(double mul, const double base[3]) {
  static const double tail[3][7] = {
      {-1.025, -0.78, -0.65012, -0.19041574084286472, -0.20819395464, -0.421064,
       -0.32733845535848671},
      {-0.3041958212306401, -0.3633036457487539, -0.35660379990111464, -0.3443074455424403,
       -0.33699592683512467, -0.30180866526242109, -0.27321683125358037},
      {-1.2, -1.2, -0.8, -0.7, -0.7, -0.4, -0.5}};
  DctParams p;
  memset(&p, 0, sizeof p);
  p.num_bands = 8;
  for (int c = 0; c < 3; c++) {
    p.params[c][0] = F(mul) * F(base[c]);
    for (int i = 0; i < 7; i++) p.params[c][i + 1] = F(tail[c][i]);
  }
  return p;
}
static const double kBaseSquare[3] = {26629.073922049845, 9311.3238710010046, 4992.2486445538634};
static const double kBaseRect[3] = {23629.073922049845, 8611.3238710010046, 4492.2486445538634};

static void finish(float* w, size_t n) { /* :1071-1077 */
  for (size_t i = 0; i < n; i++) w[i] = 1.0f / w[i];
}

int jxlo_library_dequant_table(int t, float* weights) { /* compute_table :894-1079 */
  const int wrows = 8 * kReqX[t], wcols = 8 * kReqY[t];
  const size_t num = (size_t)wrows * wcols;
  memset(weights, 0, sizeof(float) * 3 * num);
  DctParams p;
  switch (t) {
    case 0: p = params_dct8(); get_quant_weights(wrows, wcols, &p, weights); break;
    case 1: { /* Identity :387-395, :904-913 */
      static const float xyb[3][3] = {{280.0f, 3160.0f, 3160.0f}, {60.0f, 864.0f, 864.0f}, {18.0f, 200.0f, 200.0f}};
      for (int c = 0; c < 3; c++) {
        for (int i = 0; i < 64; i++) weights[64 * c + i] = xyb[c][0];
        weights[64 * c + 1] = xyb[c][1];
        weights[64 * c + 8] = xyb[c][1];
        weights[64 * c + 9] = xyb[c][2];
      }
      break;
    }
    case 2: { /* Dct2 :396-404, :914-944 */
      static const float xyb[3][6] = {{3840.0f, 2560.0f, 1280.0f, 640.0f, 480.0f, 300.0f},
                                      {960.0f, 640.0f, 320.0f, 180.0f, 140.0f, 120.0f},
                                      {640.0f, 320.0f, 128.0f, 64.0f, 32.0f, 16.0f}};
      for (int c = 0; c < 3; c++) {
        float* w = weights + c * 64;
        w[0] = (float)0xBAD;
        w[1] = xyb[c][0];
        w[8] = xyb[c][0];
        w[9] = xyb[c][1];
        for (int y = 0; y < 2; y++)
          for (int x = 0; x < 2; x++) {
            w[y * 8 + x + 2] = xyb[c][2];
            w[(y + 2) * 8 + x] = xyb[c][2];
          }
        for (int y = 0; y < 2; y++)
          for (int x = 0; x < 2; x++) w[(y + 2) * 8 + x + 2] = xyb[c][3];
        for (int y = 0; y < 4; y++)
          for (int x = 0; x < 4; x++) {
            w[y * 8 + x + 4] = xyb[c][4];
            w[(y + 4) * 8 + x] = xyb[c][4];
          }
        for (int y = 0; y < 4; y++)
          for (int x = 0; x < 4; x++) w[(y + 4) * 8 + x + 4] = xyb[c][5];
      }
      break;
    }
    case 3: { /* Dct4 :405-414, :945-959 (xyb_mul all 1) */
      float w44[3 * 16];
      p = params_dct4x4();
      get_quant_weights(4, 4, &p, w44);
      for (int c = 0; c < 3; c++) {
        for (int y = 0; y < 8; y++)
          for (int x = 0; x < 8; x++) weights[c * num + y * 8 + x] = w44[c * 16 + (y / 2) * 4 + (x / 2)];
        weights[c * num + 1] /= 1.0f;
        weights[c * num + 8] /= 1.0f;
        weights[c * num + 9] /= 1.0f;
      }
      break;
    }
    case 4: p = params_dct16(); get_quant_weights(wrows, wcols, &p, weights); break;
    case 5: p = params_dct32(); get_quant_weights(wrows, wcols, &p, weights); break;
    case 6: p = params_dct8x16(); get_quant_weights(wrows, wcols, &p, weights); break;
    case 7: p = params_dct8x32(); get_quant_weights(wrows, wcols, &p, weights); break;
    case 8: p = params_dct16x32(); get_quant_weights(wrows, wcols, &p, weights); break;
    case 9: { /* Dct4x8 :573-597, :960-972 */
      float w48[3 * 32];
      p = params_dct4x8();
      get_quant_weights(4, 8, &p, w48);
      for (int c = 0; c < 3; c++) {
        for (int y = 0; y < 8; y++)
          for (int x = 0; x < 8; x++) weights[c * num + y * 8 + x] = w48[c * 32 + (y / 2) * 8 + x];
        weights[c * num + 8] /= 1.0f;
      }
      break;
    }
    case 10: { /* AFV :599-633, :984-1069 */
      static const float afvw[3][9] = {{3072.0f, 3072.0f, 256.0f, 256.0f, 256.0f, 414.0f, 0.0f, 0.0f, 0.0f},
                                       {1024.0f, 1024.0f, 50.0f, 50.0f, 50.0f, 58.0f, 0.0f, 0.0f, 0.0f},
                                       {384.0f, 384.0f, 12.0f, 12.0f, 12.0f, 22.0f, -0.25f, -0.25f, -0.25f}};
      static const double freqs_d[16] = {0xBAD, 0xBAD, 0.8517778890324296, 5.37778436506804,
                                         0xBAD, 0xBAD, 4.734747904497923, 5.449245381693219,
                                         1.6598270267479331, 4.0, 7.275749096817861, 10.423227632456525,
                                         2.662932286148962, 7.630657783650829, 8.962388608184032,
                                         12.97166202570235};
      float w48[3 * 32], w44[3 * 16];
      DctParams p48 = params_dct4x8(), p44 = params_dct4x4();
      get_quant_weights(4, 8, &p48, w48);
      get_quant_weights(4, 4, &p44, w44);
      const float lo = F(0.8517778890324296);
      const float hi = F(12.97166202570235) - lo + 1e-6f;
      for (int c = 0; c < 3; c++) {
        float bands[4];
        bands[0] = afvw[c][5];
        for (int i = 1; i < 4; i++) bands[i] = bands[i - 1] * mult(afvw[c][i + 5]);
        float* w = weights + c * 64;
        w[0] = 1.0f;
        w[1 * 8 + 0] = afvw[c][0]; /* set(x=0,y=1) */
        w[0 * 8 + 1] = afvw[c][1];
        w[2 * 8 + 0] = afvw[c][2];
        w[0 * 8 + 2] = afvw[c][3];
        w[2 * 8 + 2] = afvw[c][4];
        for (int y = 0; y < 4; y++)
          for (int x = 0; x < 4; x++) {
            if (x < 2 && y < 2) continue;
            w[(2 * y) * 8 + 2 * x] = interpolate(F(freqs_d[y * 4 + x]) - lo, hi, bands, 4);
          }
        for (int y = 0; y < 4; y++)
          for (int x = 0; x < 8; x++) {
            if (x == 0 && y == 0) continue;
            weights[c * num + (2 * y + 1) * 8 + x] = w48[c * 32 + y * 8 + x];
          }
        for (int y = 0; y < 4; y++)
          for (int x = 0; x < 4; x++) {
            if (x == 0 && y == 0) continue;
            weights[c * num + (2 * y) * 8 + 2 * x + 1] = w44[c * 16 + y * 4 + x];
          }
      }
      break;
    }
    case 11: p = params_large(0.9, kBaseSquare); get_quant_weights(wrows, wcols, &p, weights); break;
    case 12: p = params_large(0.65, kBaseRect); get_quant_weights(wrows, wcols, &p, weights); break;
    case 13: p = params_large(1.8, kBaseSquare); get_quant_weights(wrows, wcols, &p, weights); break;
    case 14: p = params_large(1.3, kBaseRect); get_quant_weights(wrows, wcols, &p, weights); break;
    case 15: p = params_large(3.6, kBaseSquare); get_quant_weights(wrows, wcols, &p, weights); break;
    case 16: p = params_large(2.6, kBaseRect); get_quant_weights(wrows, wcols, &p, weights); break;
    default: return -1;
  }
  finish(weights, 3 * num);
  return 0;
}

/* coeff_order.rs:66-120 */
void jxlo_natural_coeff_order(int type, uint32_t* out) {
  int cx = jxlo_covered_blocks_x(type), cy = jxlo_covered_blocks_y(type);
  if (cx < cy) { int t = cx; cx = cy; cy = t; } /* orders exist for the cx >= cy member of a pair */
  const int xsize = cx * 8;
  const int xs = cx / cy;
  const int xsm = xs - 1;
  int xss = 0;
  while ((1 << xss) < xs) xss++;
  int cur = cx * cy;
  for (int i = 0; i < xsize; i++) {
    for (int j = 0; j <= i; j++) {
      int x = j, y = i - j;
      if (i % 2 != 0) { int t = x; x = y; y = t; }
      if ((y & xsm) != 0) continue;
      y >>= xss;
      int val;
      if (x < cx && y < cy) {
        val = y * cx + x;
      } else {
        val = cur++;
      }
      out[val] = (uint32_t)(y * xsize + x);
    }
  }
  for (int ir = 1; ir < xsize; ir++) {
    int ip = xsize - ir;
    int i = ip - 1;
    for (int j = 0; j <= i; j++) {
      int x = xsize - 1 - (i - j);
      int y = xsize - 1 - j;
      if (i % 2 != 0) { int t = x; x = y; y = t; }
      if ((y & xsm) != 0) continue;
      y >>= xss;
      out[cur++] = (uint32_t)(y * xsize + x);
    }
  }
}
