/* TEST INFRASTRUCTURE ONLY (see jxlo.h).  Restatement of the reference's noise synthesis:
 *   Xorshift128Plus            jxl/src/util/xorshift128plus.rs:9-71
 *   render_noise_for_group     jxl/src/frame/decode.rs:578-668   (random planes)
 *   ConvolveNoiseStage         jxl/src/render/stages/noise.rs:32-86
 *   Noise::strength            jxl/src/features/noise.rs:21-41
 *   AddNoiseStage              jxl/src/render/stages/noise.rs:140-189 */
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
  return a * b + c;
#endif
}

static uint64_t split_mix_64(uint64_t z) { /* :67-71 */
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

void jxlo_xorshift_seed(uint64_t seed, JxloXorshift* r) { /* new_with_seed, :17-30 */
  r->s0[0] = split_mix_64(seed + 0x9E3779B97F4A7C15ull);
  r->s1[0] = split_mix_64(r->s0[0]);
  for (int i = 1; i < 8; i++) {
    r->s0[i] = split_mix_64(r->s1[i - 1]);
    r->s1[i] = split_mix_64(r->s0[i]);
  }
}

void jxlo_xorshift_seeds(uint32_t a, uint32_t b, uint32_t c, uint32_t d, JxloXorshift* r) { /* :32-48 */
  r->s0[0] = split_mix_64((((uint64_t)a << 32) + b) + 0x9E3779B97F4A7C15ull);
  r->s1[0] = split_mix_64((((uint64_t)c << 32) + d) + 0x9E3779B97F4A7C15ull);
  for (int i = 1; i < 8; i++) {
    r->s0[i] = split_mix_64(r->s0[i - 1]);
    r->s1[i] = split_mix_64(r->s1[i - 1]);
  }
}

void jxlo_xorshift_fill(JxloXorshift* r, uint64_t out[8]) { /* :50-65 */
  for (int i = 0; i < 8; i++) {
    uint64_t new_s1 = r->s0[i];
    r->s0[i] = r->s1[i];
    out[i] = new_s1 + r->s0[i];
    new_s1 ^= new_s1 << 23;
    new_s1 ^= r->s0[i] ^ (new_s1 >> 18) ^ (r->s0[i] >> 5);
    r->s1[i] = new_s1;
  }
}

static inline float bits_to_float(uint32_t bits) { /* decode.rs:601 */
  const uint32_t u = (bits >> 9) | 0x3F800000u;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

/* The three random planes of a w x h image (the size after upsampling): one generator per
 * group_dim x group_dim tile, seeded with the frame indices and the tile's top-left corner, shared by the three
 * channels in turn, 16 floats per fill (decode.rs:612-668). */
void jxlo_noise_generate(uint32_t visible_frame_index, uint32_t nonvisible_frame_index, int w, int h, int group_dim,
                         float* const out[3], size_t stride) {
  for (int y0 = 0; y0 < h; y0 += group_dim) {
    for (int x0 = 0; x0 < w; x0 += group_dim) {
      JxloXorshift rng;
      jxlo_xorshift_seeds(visible_frame_index, nonvisible_frame_index, (uint32_t)x0, (uint32_t)y0, &rng);
      const int sw = w - x0 < group_dim ? w - x0 : group_dim, sh = h - y0 < group_dim ? h - y0 : group_dim;
      for (int c = 0; c < 3; c++) {
        for (int y = 0; y < sh; y++) {
          float* row = out[c] + (size_t)(y0 + y) * stride + x0;
          for (int b = 0; b < (sw + 15) / 16; b++) {
            uint64_t batch[8];
            jxlo_xorshift_fill(&rng, batch);
            const int n = sw - b * 16 < 16 ? sw - b * 16 : 16;
            for (int i = 0; i < n; i++) {
              const uint64_t v = batch[i / 2];
              row[b * 16 + i] = bits_to_float((i & 1) ? (uint32_t)(v >> 32) : (uint32_t)v);
            }
          }
        }
      }
    }
  }
}

static inline int mirror(int v, int s) {
  for (;;) {
    if (v < 0) v = -v - 1;
    else if (v >= s) v = s * 2 - v - 1;
    else return v;
  }
}

void jxlo_noise_convolve(const float* in, int w, int h, size_t stride, float* out, size_t out_stride) {
  for (int y = 0; y < h; y++) {
    const float* r[5];
    for (int k = 0; k < 5; k++) r[k] = in + (size_t)mirror(y - 2 + k, h) * stride;
    for (int x = 0; x < w; x++) {
      int xs[5];
      for (int k = 0; k < 5; k++) xs[k] = mirror(x - 2 + k, w);
      float others = 0.0f; /* noise.rs:61-76: this order */
      for (int i = 0; i < 5; i++) {
        others += r[0][xs[i]];
        others += r[1][xs[i]];
        others += r[3][xs[i]];
        others += r[4][xs[i]];
      }
      others += r[2][xs[0]];
      others += r[2][xs[1]];
      others += r[2][xs[3]];
      others += r[2][xs[4]];
      out[(size_t)y * out_stride + x] = mul_add(others, 0.16f, r[2][xs[2]] * -3.84f);
    }
  }
}

float jxlo_noise_strength(const float lut[8], float vx) { /* features/noise.rs:21-41 */
  const float k_scale = 6.0f;
  const float sv = vx * k_scale;
  const float scaled = sv > 0.0f ? sv : 0.0f; /* f32::max(0.0, x): NaN -> 0 */
  const float pre_floor = floorf(scaled), pre_frac = scaled - pre_floor;
  const float floor_x = scaled >= k_scale + 1.0f ? k_scale : pre_floor;
  const float frac_x = scaled >= k_scale + 1.0f ? 1.0f : pre_frac;
  const int fi = (int)floor_x;
  const float low = lut[fi], hi = lut[fi + 1];
  float v = (hi - low) * frac_x + low; /* scalar Rust: never contracted */
  v = v < 0.0f ? 0.0f : v;
  return v > 1.0f ? 1.0f : v;
}

void jxlo_noise_add(const float lut[8], float ytox, float ytob, float* px, float* py, float* pb, const float* rr,
                    const float* rg, const float* rc, size_t n) { /* noise.rs:151-188 */
  int all_zero = 1;
  for (int i = 0; i < 8; i++) all_zero &= lut[i] == 0.0f;
  if (all_zero) return;
  const float norm_const = 0.22f, k_rg_corr = 0.9921875f, k_rgn_corr = 0.0078125f;
  for (size_t i = 0; i < n; i++) {
    const float vx = px[i], vy = py[i];
    const float in_g = vy - vx, in_r = vy + vx;
    const float sg = jxlo_noise_strength(lut, in_g * 0.5f), sr = jxlo_noise_strength(lut, in_r * 0.5f);
    const float ar = rr[i] * norm_const, ag = rg[i] * norm_const, ac = rc[i] * norm_const;
    const float red = sr * (k_rgn_corr * ar + k_rg_corr * ac);
    const float green = sg * (k_rgn_corr * ag + k_rg_corr * ac);
    const float rgn = red + green;
    px[i] += ytox * rgn + red - green;
    py[i] += rgn;
    pb[i] += ytob * rgn;
  }
}
