// Twelve separately allocated plane-sized buffers: the time of writing three of them together (the K1-like write half of
// tools/placement_probe.hip) for every triple -- how much better than a sequential set is the best combination?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/triple_probe tools/triple_probe.hip && /tmp/triple_probe [buffers]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x)                                                             \
  do {                                                                       \
    hipError_t e = (x);                                                      \
    if (e != hipSuccess) {                                                   \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                 \
      exit(1);                                                               \
    }                                                                        \
  } while (0)
constexpr size_t kPlane = (size_t)8192 * 8192;
__global__ __launch_bounds__(256) void write3(float* p0, float* p1, float* p2) {
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = gridDim.x * 4;
  float* planes[3] = {p0, p1, p2};
  for (int b = wave; b < 1024 * 128; b += nwaves)
#pragma unroll
    for (int c = 0; c < 3; c++) {
      float4* dst = reinterpret_cast<float4*>(planes[c] + (size_t)b * 512);
      dst[lane] = make_float4(1.f, 2.f, 3.f, (float)b);
      dst[lane + 64] = make_float4(1.f, 2.f, 3.f, (float)c);
    }
}
__global__ __launch_bounds__(256) void k1_like(const int* __restrict__ coeffs, float* p0, float* p1, float* p2) {
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = gridDim.x * 4;
  float* planes[3] = {p0, p1, p2};
  for (int b = wave; b < 1024 * 128; b += nwaves) {
    const int g = b >> 7, i = b & 127;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const int4* src = reinterpret_cast<const int4*>(coeffs + (size_t)g * 3 * 65536 + (size_t)c * 65536 + i * 512);
      const int4 a = src[lane], q = src[lane + 64];
      float4* dst = reinterpret_cast<float4*>(planes[c] + (size_t)b * 512);
      dst[lane] = make_float4((float)a.x, (float)a.y, (float)a.z, (float)a.w);
      dst[lane + 64] = make_float4((float)q.x, (float)q.y, (float)q.z, (float)q.w);
    }
  }
}
__global__ __launch_bounds__(256) void filter_like(const float* __restrict__ p0, const float* __restrict__ p1,
                                                   const float* __restrict__ p2, float* t0, float* t1, float* t2) {
  const size_t n4 = kPlane / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 a = reinterpret_cast<const float4*>(p0)[i], b = reinterpret_cast<const float4*>(p1)[i],
                 c = reinterpret_cast<const float4*>(p2)[i];
    reinterpret_cast<float4*>(t0)[i] = make_float4(a.x + b.x, a.y, a.z, a.w);
    reinterpret_cast<float4*>(t1)[i] = make_float4(b.x + c.x, b.y, b.z, b.w);
    reinterpret_cast<float4*>(t2)[i] = make_float4(c.x + a.x, c.y, c.z, c.w);
  }
}
int main(int argc, char** argv) {
  const int nb = argc > 1 ? atoi(argv[1]) : 12;
  std::vector<float*> bufs(nb);
  for (int i = 0; i < nb; i++) {
    CHECK(hipMalloc(&bufs[i], kPlane * 4));
    float* pad;
    CHECK(hipMalloc(&pad, (size_t)(argc > 2 ? atoi(argv[2]) : 512) << 20));  // spacer
  }
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  auto timeit = [&](float* a, float* b, float* c) {
    float ms;
    hipLaunchKernelGGL(write3, dim3(2048), dim3(256), 0, 0, a, b, c);
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 4; r++) hipLaunchKernelGGL(write3, dim3(2048), dim3(256), 0, 0, a, b, c);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 4;
  };
  std::vector<float> all;
  float best = 1e9f;
  int bi = 0, bj = 0, bk = 0;
  for (int i = 0; i < nb; i++)
    for (int j = i + 1; j < nb; j++)
      for (int k = j + 1; k < nb; k++) {
        const float t = timeit(bufs[i], bufs[j], bufs[k]);
        all.push_back(t);
        if (t < best) best = t, bi = i, bj = j, bk = k;
      }
  printf("sequential triples (i, i+1, i+2):");
  for (int i = 0; i + 2 < nb; i++) printf(" %.4f", timeit(bufs[i], bufs[i + 1], bufs[i + 2]));
  printf("\n");
  std::sort(all.begin(), all.end());
  printf("%zu triples: min %.4f (%d %d %d)  p10 %.4f  median %.4f  p90 %.4f  max %.4f\n", all.size(), all[0], bi, bj, bk,
         all[all.size() / 10], all[all.size() / 2], all[all.size() * 9 / 10], all.back());
  printf("best triple again: %.4f %.4f\n", timeit(bufs[bi], bufs[bj], bufs[bk]), timeit(bufs[bi], bufs[bj], bufs[bk]));
  {
    int* coeffs[2];
    for (int c = 0; c < 2; c++) {
      CHECK(hipMalloc(&coeffs[c], (size_t)1024 * 3 * 65536 * 4));
      CHECK(hipMemset(coeffs[c], 0, (size_t)1024 * 3 * 65536 * 4));
    }
    auto time_k1 = [&](int* cf, float* a, float* b, float* c) {
      float ms;
      hipLaunchKernelGGL(k1_like, dim3(2048), dim3(256), 0, 0, cf, a, b, c);
      CHECK(hipEventRecord(e0));
      for (int r = 0; r < 4; r++) hipLaunchKernelGGL(k1_like, dim3(2048), dim3(256), 0, 0, cf, a, b, c);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      return ms / 4;
    };
    auto time_f = [&](float* a, float* b, float* c, float* x, float* y, float* z) {
      float ms;
      hipLaunchKernelGGL(filter_like, dim3(4096), dim3(256), 0, 0, a, b, c, x, y, z);
      CHECK(hipEventRecord(e0));
      for (int r = 0; r < 4; r++) hipLaunchKernelGGL(filter_like, dim3(4096), dim3(256), 0, 0, a, b, c, x, y, z);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      return ms / 4;
    };
    // sequential sets: planes (6s, 6s+2, 6s+4), tmp (6s+1, 6s+3, 6s+5)
    for (int s6 = 0; s6 + 5 < nb; s6 += 6)
      printf("sequential set at %d: k1-like %.4f %.4f  filter-like %.4f\n", s6, time_k1(coeffs[0], bufs[s6], bufs[s6 + 2], bufs[s6 + 4]),
             time_k1(coeffs[1], bufs[s6], bufs[s6 + 2], bufs[s6 + 4]),
             time_f(bufs[s6], bufs[s6 + 2], bufs[s6 + 4], bufs[s6 + 1], bufs[s6 + 3], bufs[s6 + 5]));
    // picked: best write3 triple as planes, then the best filter-like triple (any order) among the rest
    float bestf = 1e9f;
    int t[3] = {0, 0, 0};
    for (int i = 0; i < nb; i++)
      for (int j = 0; j < nb; j++)
        for (int k = 0; k < nb; k++) {
          if (i == j || i == k || j == k || i == bi || i == bj || i == bk || j == bi || j == bj || j == bk || k == bi || k == bj || k == bk) continue;
          if (!(i < j)) continue;  // (halve the work: the third position is free)
          const float f = time_f(bufs[bi], bufs[bj], bufs[bk], bufs[i], bufs[j], bufs[k]);
          if (f < bestf) bestf = f, t[0] = i, t[1] = j, t[2] = k;
        }
    printf("picked: planes %d %d %d, tmp %d %d %d: k1-like %.4f %.4f  filter-like %.4f\n", bi, bj, bk, t[0], t[1], t[2],
           time_k1(coeffs[0], bufs[bi], bufs[bj], bufs[bk]), time_k1(coeffs[1], bufs[bi], bufs[bj], bufs[bk]),
           time_f(bufs[bi], bufs[bj], bufs[bk], bufs[t[0]], bufs[t[1]], bufs[t[2]]));
  }
  // pairs: is there a pairwise structure?
  for (int i = 0; i < nb; i++) {
    printf("with %2d:", i);
    for (int j = 0; j < nb; j++) printf(" %.3f", i == j ? 0.f : timeit(bufs[i], bufs[j], bufs[j]));
    printf("\n");
  }
  return 0;
}
