// Write / read bandwidth of N buffers of 1 GiB each, allocated one after the other: is the device memory uniform?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/vram_map tools/vram_map.hip && /tmp/vram_map [buffers] [MiB each]
#include <hip/hip_runtime.h>

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
__global__ __launch_bounds__(256) void fill(float4* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ __launch_bounds__(256) void sum(const float4* p, size_t n, float* out) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float4 v = p[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 12345.678f) *out = acc;
}
int main(int argc, char** argv) {
  const int nb = argc > 1 ? atoi(argv[1]) : 64;
  const size_t bytes = (size_t)(argc > 2 ? atoi(argv[2]) : 1024) << 20, n = bytes / 16;
  std::vector<float4*> bufs(nb);
  float* out;
  CHECK(hipMalloc(&out, 4));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int i = 0; i < nb; i++) CHECK(hipMalloc(&bufs[i], bytes));
  for (int round = 0; round < 2; round++)
    for (int i = 0; i < nb; i++) {
      float w, r;
      for (int k = 0; k < 2; k++) hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, bufs[i], n);
      CHECK(hipEventRecord(e0));
      for (int k = 0; k < 5; k++) hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, bufs[i], n);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      CHECK(hipEventElapsedTime(&w, e0, e1));
      CHECK(hipEventRecord(e0));
      for (int k = 0; k < 5; k++) hipLaunchKernelGGL(sum, dim3(4096), dim3(256), 0, 0, bufs[i], n, out);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      CHECK(hipEventElapsedTime(&r, e0, e1));
      if (round == 1) printf("buffer %3d at %p: write %.0f GB/s  read %.0f GB/s\n", i, (void*)bufs[i], bytes * 5 / (w * 1e-3) / 1e9, bytes * 5 / (r * 1e-3) / 1e9);
    }
  return 0;
}
