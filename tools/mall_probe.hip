// mall_probe -- what the 256 MiB Infinity Cache (MALL) buys a producer -> consumer kernel pair on MI355X.
//
// The reconstruction chain is K1 (coefficients -> planes) followed by the fused filters (planes -> result):
// 12 B/px in, 12 B/px intermediate written, ~15 B/px intermediate read, 12 B/px out.  If the frame is processed
// in bands whose intermediate fits the MALL, the intermediate traffic never has to reach HBM.  This probe measures
// the ceiling of that idea with pure copy kernels:
//   1. float4 copy / read-only / write-only bandwidth against the working-set size
//   2. a banded src -> ring -> dst pipeline (A: src band -> ring slot, B: ring slot -> dst band) against the band
//      size, ring = 2 slots, one stream and two streams, compared with the unbanded two-pass form
// Build: hipcc --offload-arch=gfx950 -O3 tools/mall_probe.hip -o tools/mall_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x)                                                                      \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));                \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)

__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ src, float* __restrict__ sink, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float4 v = src[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 12345.678f) *sink = acc;
}
__global__ __launch_bounds__(256) void k_write(float4* __restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    dst[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

static int grid_for(size_t n) {
  size_t g = (n + 256 * 8 - 1) / (256 * 8);
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (int)g;
}

int main(int argc, char** argv) {
  CHK(hipSetDevice(0));
  const size_t MB = 1 << 20;
  const size_t total = 2048 * MB;
  float4 *a, *b, *ring;
  float* sink;
  CHK(hipMalloc(&a, total));
  CHK(hipMalloc(&b, total));
  CHK(hipMalloc(&ring, total));
  CHK(hipMalloc(&sink, 4));
  CHK(hipMemset(a, 1, total));
  CHK(hipMemset(b, 0, total));
  CHK(hipMemset(ring, 0, total));
  hipStream_t s0, s1;
  CHK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CHK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  auto timed = [&](auto&& body, int reps) {
    body();
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0, s0));
    for (int r = 0; r < reps; r++) body();
    CHK(hipEventRecord(e1, s0));
    CHK(hipEventSynchronize(e1));
    CHK(hipDeviceSynchronize());
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
  };
  printf("== 1. bandwidth vs working set (one kernel repeated on the same buffers), GB/s\n");
  printf("%8s %10s %10s %10s\n", "MB", "copy(r+w)", "read", "write");
  for (size_t mb : {8, 16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048}) {
    const size_t n = mb * MB / 16;
    const int reps = mb <= 128 ? 50 : 10;
    // copy: src mb/2 -> dst mb/2 so that the working set is mb
    const float c = timed([&] { hipLaunchKernelGGL(k_copy, dim3(grid_for(n / 2)), dim3(256), 0, s0, a, b, n / 2); }, reps);
    const float r = timed([&] { hipLaunchKernelGGL(k_read, dim3(grid_for(n)), dim3(256), 0, s0, a, sink, n); }, reps);
    const float w = timed([&] { hipLaunchKernelGGL(k_write, dim3(grid_for(n)), dim3(256), 0, s0, b, n); }, reps);
    printf("%8zu %10.0f %10.0f %10.0f\n", mb, mb * MB / (c * 1e-3) / 1e9, mb * MB / (r * 1e-3) / 1e9,
           mb * MB / (w * 1e-3) / 1e9);
  }
  // ---- 2. banded pipeline: D bytes src -> ring -> dst
  const size_t D = 768 * MB;  // one 8K frame: 805 MB of coefficients / planes
  printf("== 2. src -> intermediate -> dst, %zu MB per leg (ideal traffic with a cached intermediate: 2 x %zu MB)\n",
         D / MB, D / MB);
  {
    const size_t n = D / 16;
    const float t = timed(
        [&] {
          hipLaunchKernelGGL(k_copy, dim3(grid_for(n)), dim3(256), 0, s0, a, ring, n);
          hipLaunchKernelGGL(k_copy, dim3(grid_for(n)), dim3(256), 0, s0, ring, b, n);
        },
        10);
    printf("unbanded two passes: %.3f ms  (%.0f GB/s over 4 legs, %.0f GB/s counting src+dst only)\n", t,
           4.0 * D / (t * 1e-3) / 1e9, 2.0 * D / (t * 1e-3) / 1e9);
    const float t1 = timed([&] { hipLaunchKernelGGL(k_copy, dim3(grid_for(n)), dim3(256), 0, s0, a, b, n); }, 10);
    printf("single copy src -> dst: %.3f ms (%.0f GB/s)\n", t1, 2.0 * D / (t1 * 1e-3) / 1e9);
  }
  std::vector<hipEvent_t> evA(256), evB(256);
  for (auto& e : evA) CHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (auto& e : evB) CHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  printf("%8s %6s %12s %12s %12s\n", "band MB", "slots", "1 stream ms", "2 streams ms", "GB/s(src+dst)");
  for (size_t band_mb : {12, 24, 48, 96, 192}) {
    for (int slots : {2, 3}) {
      const size_t band = band_mb * MB, nb = D / band, n = band / 16;
      const float t1 = timed(
          [&] {
            for (size_t i = 0; i < nb; i++) {
              float4* slot = ring + (i % slots) * n;
              hipLaunchKernelGGL(k_copy, dim3(grid_for(n)), dim3(256), 0, s0, a + i * n, slot, n);
              hipLaunchKernelGGL(k_copy, dim3(grid_for(n)), dim3(256), 0, s0, slot, b + i * n, n);
            }
          },
          10);
      // two streams: A(i) on s0, B(i) on s1 after A(i); A(i) waits for B(i - slots) (slot free)
      const float t2 = timed(
          [&] {
            for (size_t i = 0; i < nb; i++) {
              float4* slot = ring + (i % slots) * n;
              if (i >= (size_t)slots) CHK(hipStreamWaitEvent(s0, evB[i - slots], 0));
              hipLaunchKernelGGL(k_copy, dim3(grid_for(n)), dim3(256), 0, s0, a + i * n, slot, n);
              CHK(hipEventRecord(evA[i], s0));
              CHK(hipStreamWaitEvent(s1, evA[i], 0));
              hipLaunchKernelGGL(k_copy, dim3(grid_for(n)), dim3(256), 0, s1, slot, b + i * n, n);
              CHK(hipEventRecord(evB[i], s1));
            }
            CHK(hipStreamWaitEvent(s0, evB[nb - 1], 0));
          },
          10);
      printf("%8zu %6d %12.3f %12.3f %12.0f\n", band_mb, slots, t1, t2, 2.0 * D / (std::min(t1, t2) * 1e-3) / 1e9);
    }
  }
  return 0;
}
