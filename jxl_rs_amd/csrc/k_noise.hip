// Noise synthesis: the random planes of render_noise_for_group (jxl/src/frame/decode.rs:578-668) from the
// reference's Xorshift128Plus (jxl/src/util/xorshift128plus.rs:9-71), ConvolveNoiseStage and AddNoiseStage
// (jxl/src/render/stages/noise.rs:32-86, :140-189), Noise::strength (jxl/src/features/noise.rs:21-41).
//
// Generation.  The reference runs ONE generator per 256x256 tile of the (upsampled) image -- 8 independent
// xorshift128+ lanes, 16 floats per fill, the three channels one after the other -- i.e. 12288 dependent
// steps per tile.  xorshift128+ is linear over GF(2): the state after k steps is T^k applied to the seed
// state, so a thread can start anywhere in the stream.  `pow2` holds T^(2^j) as 128 columns of 128 bits; a
// thread = (tile, channel, chunk of 32 rows, lane): it derives the tile's seed, jumps (c*rows + r0) * fills
// per row ahead by the binary expansion of that count, and then produces its 32 rows -- identical bits, 96
// times the parallelism.
#include "jxlh_internal.h"

namespace jxlh {
namespace {

constexpr int kNoiseTile = 256;       // group_dim (frame/decode.rs:588)
constexpr int kNoiseChunkRows = 32;
constexpr int kNoiseChunks = kNoiseTile / kNoiseChunkRows;

__device__ __forceinline__ uint64_t split_mix_64(uint64_t z) {  // xorshift128plus.rs:67-71
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__device__ __forceinline__ float bits_to_float(uint32_t bits) {  // decode.rs:601
  return __uint_as_float((bits >> 9) | 0x3F800000u);
}

__global__ __launch_bounds__(256) void k_noise_generate(float* __restrict__ o0, float* __restrict__ o1,
                                                        float* __restrict__ o2, size_t stride, int w, int h, int tiles_x,
                                                        int tile_y0, int ntiles, uint32_t visible, uint32_t nonvisible,
                                                        const uint2* __restrict__ pow2 /* [16][128][2 x u64 as 4 x u32] */) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int lane = idx & 7;
  int rest = idx >> 3;
  const int chunk = rest % kNoiseChunks;
  rest /= kNoiseChunks;
  const int c = rest % 3;
  const int tile = rest / 3;
  if (tile >= ntiles) return;
  const int x0 = (tile % tiles_x) * kNoiseTile, y0 = (tile_y0 + tile / tiles_x) * kNoiseTile;
  const int sw = min(kNoiseTile, w - x0), sh = min(kNoiseTile, h - y0);
  const int r0 = chunk * kNoiseChunkRows, r1 = min(sh, r0 + kNoiseChunkRows);
  if (r0 >= sh) return;
  const int bpr = (sw + 15) / 16;
  // new_with_seeds (xorshift128plus.rs:32-48): lane i has gone through i + 1 mixing rounds
  uint64_t s0 = (((uint64_t)visible << 32) + nonvisible) + 0x9E3779B97F4A7C15ull;
  uint64_t s1 = (((uint64_t)(uint32_t)x0 << 32) + (uint32_t)y0) + 0x9E3779B97F4A7C15ull;
  for (int i = 0; i <= lane; i++) {
    s0 = split_mix_64(s0);
    s1 = split_mix_64(s1);
  }
  // jump ahead by k fills: state <- T^k state
  uint32_t k = (uint32_t)((c * sh + r0) * bpr);
  const ulonglong2* __restrict__ cols = reinterpret_cast<const ulonglong2*>(pow2);
  for (int j = 0; k != 0; j++, k >>= 1) {
    if (!(k & 1)) continue;
    const ulonglong2* __restrict__ m = cols + j * 128;
    uint64_t n0 = 0, n1 = 0;
#pragma unroll 8
    for (int b = 0; b < 64; b++) {
      const uint64_t ma = 0 - ((s0 >> b) & 1), mb = 0 - ((s1 >> b) & 1);
      const ulonglong2 ca = m[b], cb = m[64 + b];
      n0 ^= (ca.x & ma) ^ (cb.x & mb);
      n1 ^= (ca.y & ma) ^ (cb.y & mb);
    }
    s0 = n0;
    s1 = n1;
  }
  float* __restrict__ out = c == 0 ? o0 : c == 1 ? o1 : o2;
  for (int r = r0; r < r1; r++) {
    float* __restrict__ row = out + (size_t)(y0 + r) * stride + x0 + 2 * lane;
    for (int b = 0; b < bpr; b++) {
      uint64_t n1 = s0;  // fill (xorshift128plus.rs:50-65)
      s0 = s1;
      const uint64_t bits = n1 + s0;
      n1 ^= n1 << 23;
      n1 ^= s0 ^ (n1 >> 18) ^ (s0 >> 5);
      s1 = n1;
      const int x = b * 16 + 2 * lane;
      const float lo = bits_to_float((uint32_t)bits), hi = bits_to_float((uint32_t)(bits >> 32));
      if (x + 1 < sw) {
        *reinterpret_cast<float2*>(row + b * 16) = make_float2(lo, hi);
      } else if (x < sw) {
        row[b * 16] = lo;
      }
    }
  }
}

__device__ __forceinline__ int mirror_idx(int v, int s) {
  while (v < 0 || v >= s) v = v < 0 ? -v - 1 : 2 * s - v - 1;
  return v;
}

// ConvolveNoiseStage on a 5x5 window held row-major in win[25] (noise.rs:57-80: this summation order)
__device__ __forceinline__ float convolve25(const float (&win)[25]) {
  float others = 0.0f;
#pragma unroll
  for (int i = 0; i < 5; i++) {
    others += win[0 * 5 + i];
    others += win[1 * 5 + i];
    others += win[3 * 5 + i];
    others += win[4 * 5 + i];
  }
  others += win[2 * 5 + 0];
  others += win[2 * 5 + 1];
  others += win[2 * 5 + 3];
  others += win[2 * 5 + 4];
  return __builtin_fmaf(others, 0.16f, win[2 * 5 + 2] * -3.84f);
}

struct NoiseLut {
  float v[8];
};

__device__ __forceinline__ float noise_strength(const NoiseLut& lut, float vx) {  // features/noise.rs:21-41
  constexpr float kScale = 6.0f;
  const float sv = vx * kScale;
  const float scaled = sv > 0.0f ? sv : 0.0f;
  const float pre_floor = __builtin_floorf(scaled), pre_frac = scaled - pre_floor;
  const bool top = scaled >= kScale + 1.0f;
  const float floor_x = top ? kScale : pre_floor;
  const float frac_x = top ? 1.0f : pre_frac;
  const int fi = (int)floor_x;
  float low = lut.v[0], hi = lut.v[1];
#pragma unroll
  for (int i = 1; i < 7; i++) {  // select instead of indexing the kernel-argument array per lane
    low = fi == i ? lut.v[i] : low;
    hi = fi == i ? lut.v[i + 1] : hi;
  }
  float v = (hi - low) * frac_x + low;
  v = v < 0.0f ? 0.0f : v;
  return v > 1.0f ? 1.0f : v;
}

__device__ __forceinline__ void add_noise(const NoiseLut& lut, float ytox, float ytob, float rnd_r, float rnd_g,
                                          float rnd_c, float& vx, float& vy, float& vb) {  // noise.rs:163-187
  constexpr float kNorm = 0.22f, kRgCorr = 0.9921875f, kRgnCorr = 0.0078125f;
  const float in_g = vy - vx, in_r = vy + vx;
  const float sg = noise_strength(lut, in_g * 0.5f), sr = noise_strength(lut, in_r * 0.5f);
  const float ar = rnd_r * kNorm, ag = rnd_g * kNorm, ac = rnd_c * kNorm;
  const float red = sr * (kRgnCorr * ar + kRgCorr * ac);
  const float green = sg * (kRgnCorr * ag + kRgCorr * ac);
  const float rg = red + green;
  vx += ytox * rg + red - green;
  vy += rg;
  vb += ytob * rg;
}

// ConvolveNoise x3 + AddNoise in one pass: a thread walks R rows of one column, sliding three 5x5 windows.
// MODE 0: frame path (planes updated in place).  MODE 1: convolution only, first channel, out-of-place (hook).
template <int MODE>
__global__ __launch_bounds__(256) void k_noise_apply(const float* __restrict__ n0, const float* __restrict__ n1,
                                                     const float* __restrict__ n2, size_t nstride, float* __restrict__ p0,
                                                     float* __restrict__ p1, float* __restrict__ p2, size_t pstride, int w,
                                                     int h, int y_begin, int y_end, int rows_per_thread, NoiseLut lut,
                                                     float ytox, float ytob) {
  constexpr int NC = MODE == 0 ? 3 : 1;
  const int x = blockIdx.x * 256 + threadIdx.x;
  const int ys = y_begin + blockIdx.y * rows_per_thread;
  if (x >= w || ys >= y_end) return;
  const float* __restrict__ np[3] = {n0, n1, n2};
  float win[NC][25];
  int xs[5];
#pragma unroll
  for (int k = 0; k < 5; k++) xs[k] = mirror_idx(x - 2 + k, w);
#pragma unroll
  for (int c = 0; c < NC; c++)
#pragma unroll
    for (int ky = 0; ky < 4; ky++) {
      const float* __restrict__ row = np[c] + (size_t)mirror_idx(ys - 2 + ky, h) * nstride;
#pragma unroll
      for (int kx = 0; kx < 5; kx++) win[c][(ky + 1) * 5 + kx] = row[xs[kx]];
    }
  const int ye = min(y_end, ys + rows_per_thread);
#pragma unroll 1
  for (int y = ys; y < ye; y++) {
    float conv[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
#pragma unroll
      for (int t = 0; t < 20; t++) win[c][t] = win[c][t + 5];
      const float* __restrict__ row = np[c] + (size_t)mirror_idx(y + 2, h) * nstride;
#pragma unroll
      for (int kx = 0; kx < 5; kx++) win[c][20 + kx] = row[xs[kx]];
      conv[c] = convolve25(win[c]);
    }
    const size_t pi = (size_t)y * pstride + x;
    if constexpr (MODE == 0) {
      float vx = p0[pi], vy = p1[pi], vb = p2[pi];
      add_noise(lut, ytox, ytob, conv[0], conv[1], conv[2], vx, vy, vb);
      p0[pi] = vx;
      p1[pi] = vy;
      p2[pi] = vb;
    } else {
      p0[pi] = conv[0];
    }
  }
}

// AddNoiseStage alone on tight arrays (hook)
__global__ void k_noise_add(float* __restrict__ px, float* __restrict__ py, float* __restrict__ pb,
                            const float* __restrict__ rr, const float* __restrict__ rg, const float* __restrict__ rc,
                            size_t n, NoiseLut lut, float ytox, float ytob) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float vx = px[i], vy = py[i], vb = pb[i];
  add_noise(lut, ytox, ytob, rr[i], rg[i], rc[i], vx, vy, vb);
  px[i] = vx;
  py[i] = vy;
  pb[i] = vb;
}

}  // namespace

// T^(2^j), j = 0..15, of one xorshift128+ lane as 128 columns of {lo, hi} (column b = image of state bit b;
// bits 0..63 = s0, 64..127 = s1); computed on the host once
void xorshift_jump_table(uint64_t out[16][128][2]) {
  auto step = [](uint64_t& s0, uint64_t& s1) {
    uint64_t n1 = s0;
    s0 = s1;
    n1 ^= n1 << 23;
    n1 ^= s0 ^ (n1 >> 18) ^ (s0 >> 5);
    s1 = n1;
  };
  for (int b = 0; b < 128; b++) {
    uint64_t s0 = b < 64 ? 1ull << b : 0, s1 = b < 64 ? 0 : 1ull << (b - 64);
    step(s0, s1);
    out[0][b][0] = s0;
    out[0][b][1] = s1;
  }
  for (int j = 1; j < 16; j++) {  // square: column b of M^2 = M applied to column b of M
    for (int b = 0; b < 128; b++) {
      const uint64_t v0 = out[j - 1][b][0], v1 = out[j - 1][b][1];
      uint64_t r0 = 0, r1 = 0;
      for (int t = 0; t < 64; t++) {
        if ((v0 >> t) & 1) { r0 ^= out[j - 1][t][0]; r1 ^= out[j - 1][t][1]; }
        if ((v1 >> t) & 1) { r0 ^= out[j - 1][64 + t][0]; r1 ^= out[j - 1][64 + t][1]; }
      }
      out[j][b][0] = r0;
      out[j][b][1] = r1;
    }
  }
}

// random planes for the tile rows [tile_y0, tile_y1) of a w x h image
void launch_noise_generate(hipStream_t s, float* const out[3], size_t stride, int w, int h, int tile_y0, int tile_y1,
                           uint32_t visible, uint32_t nonvisible, const void* jump_table_dev) {
  const int tiles_x = (w + kNoiseTile - 1) / kNoiseTile;
  const int ntiles = tiles_x * (tile_y1 - tile_y0);
  if (ntiles <= 0) return;
  const long threads = (long)ntiles * 3 * kNoiseChunks * 8;
  hipLaunchKernelGGL(k_noise_generate, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, out[0], out[1], out[2],
                     stride, w, h, tiles_x, tile_y0, ntiles, visible, nonvisible,
                     static_cast<const uint2*>(jump_table_dev));
}

void launch_noise_apply(hipStream_t s, const float* const noise[3], size_t nstride, float* const planes[3], size_t pstride,
                        int w, int h, int y0, int y1, const float lut[8], float ytox, float ytob) {
  if (w <= 0 || y1 <= y0) return;
  NoiseLut l;
  for (int i = 0; i < 8; i++) l.v[i] = lut[i];
  constexpr int kRows = 8;
  const dim3 grid((w + 255) / 256, (y1 - y0 + kRows - 1) / kRows);
  hipLaunchKernelGGL(k_noise_apply<0>, grid, dim3(256), 0, s, noise[0], noise[1], noise[2], nstride, planes[0], planes[1],
                     planes[2], pstride, w, h, y0, y1, kRows, l, ytox, ytob);
}

void launch_noise_convolve(hipStream_t s, const float* in, float* out, int w, int h) {
  if (w <= 0 || h <= 0) return;
  NoiseLut l = {};
  constexpr int kRows = 8;
  const dim3 grid((w + 255) / 256, (h + kRows - 1) / kRows);
  hipLaunchKernelGGL(k_noise_apply<1>, grid, dim3(256), 0, s, in, in, in, (size_t)w, out, out, out, (size_t)w, w, h, 0, h,
                     kRows, l, 0.0f, 0.0f);
}

void launch_noise_add(hipStream_t s, float* const planes[3], const float* const rnd[3], size_t n, const float lut[8],
                      float ytox, float ytob) {
  if (n == 0) return;
  NoiseLut l;
  for (int i = 0; i < 8; i++) l.v[i] = lut[i];
  hipLaunchKernelGGL(k_noise_add, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, planes[0], planes[1], planes[2],
                     rnd[0], rnd[1], rnd[2], n, l, ytox, ytob);
}

}  // namespace jxlh
