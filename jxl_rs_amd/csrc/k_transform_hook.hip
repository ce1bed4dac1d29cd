// Stage-level test hook: transform_to_pixels on n independent varblocks of one type
// (jxl_transforms/src/transform.rs:666-677), the device analogue of the reference's
// jxl_transforms unit tests.  Uses exactly the cores the frame kernel K1 uses
// (varblock_core.h / varblock_large.h); only the staging differs (coefficients arrive
// already dequantised as f32).
#include "varblock_core.h"
#include "varblock_large.h"

namespace jxlh {
namespace {

template <class S>
__global__ __launch_bounds__(64) void k_t2p_dct(uint32_t n, const float* __restrict__ coeffs,
                                                const float* __restrict__ lf, float* __restrict__ pixels) {
  __shared__ __attribute__((aligned(16))) float buf[S::kTile];
  const int lane = threadIdx.x;
  constexpr int CY = S::R / 8, CX = S::C / 8;
  const uint32_t nbatches = (n + S::NB - 1) / S::NB;
  for (uint32_t batch = blockIdx.x; batch < nbatches; batch += gridDim.x) {
    const uint32_t base = batch * S::NB;
    const int nb = (int)min((uint32_t)S::NB, n - base);
#pragma unroll
    for (int j = 0; j < S::E / 4; j++) {
      const int fl = (j * 64 + lane) * 4;
      const int b = fl / S::N, k = fl % S::N;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (b < nb) v = *reinterpret_cast<const float4*>(coeffs + (size_t)(base + b) * S::N + k);
      if constexpr (S::kWide) {
        buf[m_addr<S>(b, k)] = v.x;
        buf[m_addr<S>(b, k + 1)] = v.y;
        buf[m_addr<S>(b, k + 2)] = v.z;
        buf[m_addr<S>(b, k + 3)] = v.w;
      } else {
        *reinterpret_cast<float4*>(buf + m_addr<S>(b, k)) = v;
      }
    }
    wave_sync();
    idct_batch<S>(
        buf, nb, lane, [&](int b, int y, int x) { return lf[(size_t)(base + b) * (CY * CX) + y * CX + x]; },
        [&](int b, int x, int yb, const float(&v)[8]) {
#pragma unroll
          for (int i = 0; i < 8; i++) pixels[(size_t)(base + b) * S::N + (yb * 8 + i) * S::C + x] = v[i];
        });
  }
}

__global__ __launch_bounds__(64) void k_t2p_special(int type, uint32_t n, const float* __restrict__ coeffs,
                                                    const float* __restrict__ lf, float* __restrict__ pixels) {
  __shared__ float buf[2 * kSpecNB * kSpecPitch];
  const int lane = threadIdx.x;
  float* tin = buf;
  float* tout = buf + kSpecNB * kSpecPitch;
  const uint32_t nbatches = (n + kSpecNB - 1) / kSpecNB;
  for (uint32_t batch = blockIdx.x; batch < nbatches; batch += gridDim.x) {
    const uint32_t base = batch * kSpecNB;
    const int nb = (int)min((uint32_t)kSpecNB, n - base);
    for (int i = lane; i < nb * 64; i += 64) tin[(i / 64) * kSpecPitch + (i % 64)] = coeffs[(size_t)base * 64 + i];
    wave_sync();
    if (lane < nb) {
      float* c = tin + lane * kSpecPitch;
      c[0] = lf[base + lane];
      special_8x8(type, c, tout + lane * kSpecPitch);
    }
    wave_sync();
    for (int i = lane; i < nb * 64; i += 64) pixels[(size_t)base * 64 + i] = tout[(i / 64) * kSpecPitch + (i % 64)];
    wave_sync();
  }
}

__global__ __launch_bounds__(kLargeThreads) void k_t2p_large(int type, uint32_t n, const float* __restrict__ coeffs,
                                                             const float* __restrict__ lf,
                                                             float* __restrict__ pixels) {
  __shared__ __attribute__((aligned(16))) float lds[kLargeWaves * kLargeTile + 2048];
  const int cx = covered_x(type), cy = covered_y(type);
  const size_t N = (size_t)cx * cy * 64;
  for (uint32_t blk = blockIdx.x; blk < n; blk += gridDim.x) {
    const float* c = coeffs + blk * N;
    large_varblock_channel(
        type, [&](int k) { return c[k]; }, lf + (size_t)blk * cx * cy, cx, pixels + blk * N,
        PixLayout{cx * 8, 8 * cx * 8, 0}, lds, threadIdx.x);
  }
}

template <class S>
void launch_dct(hipStream_t s, uint32_t n, const float* coeffs, const float* lf, float* pixels) {
  const uint32_t nbatches = (n + S::NB - 1) / S::NB;
  hipLaunchKernelGGL(k_t2p_dct<S>, dim3(min(nbatches, 4096u)), dim3(64), 0, s, n, coeffs, lf, pixels);
}

}  // namespace

void launch_transform_to_pixels(hipStream_t s, int type, uint32_t n, const float* coeffs, const float* lf,
                                float* pixels) {
  if (n == 0) return;
  switch (type) {
    case 0: launch_dct<Shape<8, 8>>(s, n, coeffs, lf, pixels); break;
    case 4: launch_dct<Shape<16, 16>>(s, n, coeffs, lf, pixels); break;
    case 5: launch_dct<Shape<32, 32>>(s, n, coeffs, lf, pixels); break;
    case 6: launch_dct<Shape<16, 8>>(s, n, coeffs, lf, pixels); break;
    case 7: launch_dct<Shape<8, 16>>(s, n, coeffs, lf, pixels); break;
    case 8: launch_dct<Shape<32, 8, 4>>(s, n, coeffs, lf, pixels); break;
    case 9: launch_dct<Shape<8, 32>>(s, n, coeffs, lf, pixels); break;
    case 10: launch_dct<Shape<32, 16>>(s, n, coeffs, lf, pixels); break;
    case 11: launch_dct<Shape<16, 32>>(s, n, coeffs, lf, pixels); break;
    case 1: case 2: case 3: case 12: case 13: case 14: case 15: case 16: case 17: {
      const uint32_t nbatches = (n + kSpecNB - 1) / kSpecNB;
      hipLaunchKernelGGL(k_t2p_special, dim3(min(nbatches, 4096u)), dim3(64), 0, s, type, n, coeffs, lf, pixels);
      break;
    }
    default:
      hipLaunchKernelGGL(k_t2p_large, dim3(min(n, 2048u)), dim3(kLargeThreads), 0, s, type, n, coeffs, lf, pixels);
      break;
  }
}

}  // namespace jxlh
