// Chroma upsampling of sub-sampled channels (JPEG recompressions): HorizontalChromaUpsample and
// VerticalChromaUpsample of the reference (jxl/src/render/stages/chroma_upsample.rs:31-63, :108-147), run
// in the order of frame/render.rs:569-576 (horizontal, then vertical) in ONE pass: a thread takes one
// sample of the sub-sampled channel and produces the 2x1 / 1x2 / 2x2 output samples that depend on it.
// Edges: the pipeline mirrors a stage's input at the borders of the (sub-sampled) channel image,
// ceil(size / 2^shift) samples (render/low_memory_pipeline/render_group.rs:389-476, util/mirror.rs:8-19).
// Arithmetic: out = mul_add(neighbour, 0.25, centre * 0.75) -- one rounded product, one FMA.
#include "jxlh_internal.h"

namespace jxlh {
namespace {

__device__ __forceinline__ int mirror_idx(int v, int s) {  // util/mirror.rs:8-19
  while (v < 0 || v >= s) v = v < 0 ? -v - 1 : 2 * s - v - 1;
  return v;
}

__device__ __forceinline__ float blend(float neighbour, float centre) {
  return __builtin_fmaf(neighbour, 0.25f, centre * 0.75f);
}

// layouts: raster or the 8x8-tiled one K1 writes for the fused filters
template <bool HS, bool VS>
__global__ __launch_bounds__(256) void k_chroma_upsample(const float* __restrict__ src, float* __restrict__ dst,
                                                         const PixLayout slay, const PixLayout dlay, int cw, int ch,
                                                         int sy0, int sy1,
                                                         int out_w, int out_h) {
  const int sx = blockIdx.x * 64 + (threadIdx.x & 63);
  const int sy = sy0 + blockIdx.y * 4 + (threadIdx.x >> 6);
  if (sx >= cw || sy >= sy1) return;
  constexpr int NR = VS ? 3 : 1;
  const int rows[3] = {VS ? mirror_idx(sy - 1, ch) : sy, sy, VS ? mirror_idx(sy + 1, ch) : sy};
  const int xp = mirror_idx(sx - 1, cw), xn = mirror_idx(sx + 1, cw);
  float h[NR][2];  // the horizontal stage's output samples (2*sx, 2*sx + 1) of each row
#pragma unroll
  for (int r = 0; r < NR; r++) {
    const int y = VS ? rows[r] : sy;
    const float cur = src[slay.at(sx, y)];
    if (HS) {
      h[r][0] = blend(src[slay.at(xp, y)], cur);
      h[r][1] = blend(src[slay.at(xn, y)], cur);
    } else {
      h[r][0] = cur;
    }
  }
  constexpr int NX = HS ? 2 : 1;
#pragma unroll
  for (int i = 0; i < NX; i++) {
    const int ox = HS ? 2 * sx + i : sx;
    if (ox >= out_w) continue;
    if (VS) {
      const float up = blend(h[0][i], h[1][i]), down = blend(h[2][i], h[1][i]);
      if (2 * sy < out_h) dst[dlay.at(ox, 2 * sy)] = up;
      if (2 * sy + 1 < out_h) dst[dlay.at(ox, 2 * sy + 1)] = down;
    } else {
      dst[dlay.at(ox, sy)] = h[0][i];
    }
  }
}

}  // namespace

void launch_chroma_upsample(hipStream_t s, const float* src, float* dst, const PixLayout& slay,
                            const PixLayout& dlay, int hshift, int vshift,
                            int cw, int ch, int sy0, int sy1, int out_w, int out_h) {
  if (cw <= 0 || sy1 <= sy0 || (!hshift && !vshift)) return;
  const dim3 grid((cw + 63) / 64, (sy1 - sy0 + 3) / 4), block(256);
  if (hshift && vshift)
    hipLaunchKernelGGL((k_chroma_upsample<true, true>), grid, block, 0, s, src, dst, slay, dlay, cw, ch, sy0, sy1, out_w, out_h);
  else if (hshift)
    hipLaunchKernelGGL((k_chroma_upsample<true, false>), grid, block, 0, s, src, dst, slay, dlay, cw, ch, sy0, sy1, out_w, out_h);
  else
    hipLaunchKernelGGL((k_chroma_upsample<false, true>), grid, block, 0, s, src, dst, slay, dlay, cw, ch, sy0, sy1, out_w, out_h);
}

}  // namespace jxlh
