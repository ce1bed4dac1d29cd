// Write-only, read-only and copy bandwidth of the device with float4 grid-stride kernels (plain and nt accesses):
// what a kernel that only WRITES its output (K1 on the slot-bucketed entries: 12 B/px out, ~0.6 B/px in) can reach.
//   hipcc --offload-arch=gfx950 -O3 tools/write_probe.hip -o tools/write_probe && tools/write_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ void k_fill(f4* __restrict__ p, size_t n, float v) {
  const f4 x = {v, v, v, v};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    if (NT) __builtin_nontemporal_store(x, p + i);
    else p[i] = x;
  }
}
template <bool NT>
__global__ void k_read(const f4* __restrict__ p, size_t n, float* out) {
  f4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    acc += NT ? __builtin_nontemporal_load(p + i) : p[i];
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) *out = 1.0f;
}
template <bool NT>
__global__ void k_copy(const f4* __restrict__ a, f4* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const f4 v = NT ? __builtin_nontemporal_load(a + i) : a[i];
    if (NT) __builtin_nontemporal_store(v, b + i);
    else b[i] = v;
  }
}
int main() {
  const size_t bytes = (size_t)3 * 8192 * 8192 * 4;  // one 8K frame's planes: 805 MB
  f4 *a, *b;
  float* flag;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&flag, 4);
  hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
  const size_t n = bytes / 16;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch, double gb) {
    for (int i = 0; i < 20; i++) launch();
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
      hipEventRecord(e0);
      for (int i = 0; i < 10; i++) launch();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms / 10 < best) best = ms / 10;
    }
    printf("%-28s %.4f ms  %.0f GB/s\n", name, best, gb / (best * 1e-3));
  };
  const double gb = bytes / 1e9;
  for (int grid : {2048, 8192, 65536}) {
    printf("grid %d x 256\n", grid);
    run("  write plain", [&] { hipLaunchKernelGGL(k_fill<false>, dim3(grid), dim3(256), 0, 0, a, n, 1.0f); }, gb);
    run("  write nt", [&] { hipLaunchKernelGGL(k_fill<true>, dim3(grid), dim3(256), 0, 0, a, n, 1.0f); }, gb);
    run("  read plain", [&] { hipLaunchKernelGGL(k_read<false>, dim3(grid), dim3(256), 0, 0, a, n, flag); }, gb);
    run("  read nt", [&] { hipLaunchKernelGGL(k_read<true>, dim3(grid), dim3(256), 0, 0, a, n, flag); }, gb);
    run("  copy plain (r + w bytes)", [&] { hipLaunchKernelGGL(k_copy<false>, dim3(grid), dim3(256), 0, 0, a, b, n); }, 2 * gb);
    run("  copy nt (r + w bytes)", [&] { hipLaunchKernelGGL(k_copy<true>, dim3(grid), dim3(256), 0, 0, a, b, n); }, 2 * gb);
  }
  return 0;
}
