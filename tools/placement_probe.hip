// Does the placement of a frame's plane buffers show in a kernel that only MOVES bytes the way K1's 8x8 class does?
// K candidate sets of {3 planes, 3 tmp planes} + one coefficient buffer, allocated in the library's order; per set: the
// time of (a) a K1-like pass (a wavefront reads 2 KB of each channel of a group slab, 256 KB apart, and writes 2 KB of each
// plane at the same tile offset) and (b) a filter-like pass (read the three planes, write the three tmp planes).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/placement_probe tools/placement_probe.hip && /tmp/placement_probe [sets]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kSize = 8192, kGroups = 1024, kGroupArea = 65536;
constexpr size_t kPlane = (size_t)kSize * kSize;

struct Set {
  float* planes[3];
  float* tmp[3];
  float* lf[6];
};

__global__ __launch_bounds__(256) void k1_like(const int* __restrict__ coeffs, float* p0, float* p1, float* p2) {
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = gridDim.x * 4;
  const int nbatches = kGroups * 128;  // 8 blocks of 64 coefficients per batch
  float* planes[3] = {p0, p1, p2};
  for (int b = wave; b < nbatches; b += nwaves) {
    const int g = b >> 7, i = b & 127;  // group, batch inside the group (raster: 32 blocks per row -> 4 batches per row)
    const int gy = g >> 5, gx = g & 31, iy = i >> 2, ix0 = (i & 3) * 8;
    const size_t tile = ((size_t)(gy * 32 + iy) * 1024 + gx * 32 + ix0) * 64;  // floats: 8x8 tiles of the plane, raster
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const int4* src = reinterpret_cast<const int4*>(coeffs + (size_t)g * 3 * kGroupArea + (size_t)c * kGroupArea + i * 512);
      const int4 a = src[lane], q = src[lane + 64];
      float4* dst = reinterpret_cast<float4*>(planes[c] + tile);
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

// the K1-like pass split into its two halves, and the planes carved from ONE allocation
__global__ __launch_bounds__(256) void write3(float* p0, float* p1, float* p2) {
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = gridDim.x * 4;
  float* planes[3] = {p0, p1, p2};
  for (int b = wave; b < kGroups * 128; b += nwaves)
#pragma unroll
    for (int c = 0; c < 3; c++) {
      float4* dst = reinterpret_cast<float4*>(planes[c] + (size_t)b * 512);
      dst[lane] = make_float4(1.f, 2.f, 3.f, (float)b);
      dst[lane + 64] = make_float4(1.f, 2.f, 3.f, (float)c);
    }
}
__global__ __launch_bounds__(256) void read3(const int* __restrict__ coeffs, int* sink) {
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = gridDim.x * 4;
  int acc = 0;
  for (int b = wave; b < kGroups * 128; b += nwaves) {
    const int g = b >> 7, i = b & 127;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const int4* src = reinterpret_cast<const int4*>(coeffs + (size_t)g * 3 * kGroupArea + (size_t)c * kGroupArea + i * 512);
      const int4 a = src[lane], q = src[lane + 64];
      acc += a.x + a.y + a.z + a.w + q.x + q.y + q.z + q.w;
    }
  }
  if (acc == 0x7fffffff) *sink = acc;
}

// the three planes interleaved inside one buffer at a granularity of G floats (G = 512: a batch; 65536: 128 batches)
template <int G>
__global__ __launch_bounds__(256) void write3_interleaved(float* arena) {
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = gridDim.x * 4;
  for (int b = wave; b < kGroups * 128; b += nwaves)
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const size_t off = ((size_t)b * 512 / G * 3 + c) * G + (size_t)b * 512 % G;
      float4* dst = reinterpret_cast<float4*>(arena + off);
      dst[lane] = make_float4(1.f, 2.f, 3.f, (float)b);
      dst[lane + 64] = make_float4(1.f, 2.f, 3.f, (float)c);
    }
}

#define CHECK(x)                                                                 \
  do {                                                                           \
    hipError_t e = (x);                                                          \
    if (e != hipSuccess) {                                                       \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                     \
      exit(1);                                                                   \
    }                                                                            \
  } while (0)

int main(int argc, char** argv) {
  const int nsets = argc > 1 ? atoi(argv[1]) : 8;
  std::vector<Set> sets(nsets);
  std::vector<int*> coeffs(nsets);
  for (int s = 0; s < nsets; s++) {
    for (int c = 0; c < 3; c++) {  // the library's order (abi_frame.hip): plane, tmp (+ one block row), two LF planes
      CHECK(hipMalloc(&sets[s].planes[c], kPlane * 4));
      CHECK(hipMalloc(&sets[s].tmp[c], (kPlane + 8 * kSize) * 4));
      CHECK(hipMalloc(&sets[s].lf[2 * c], (size_t)1024 * 1024 * 4));
      CHECK(hipMalloc(&sets[s].lf[2 * c + 1], (size_t)1024 * 1024 * 4));
    }
    CHECK(hipMalloc(&coeffs[s], (size_t)kGroups * 3 * kGroupArea * 4));
    CHECK(hipMemset(coeffs[s], 0, (size_t)kGroups * 3 * kGroupArea * 4));
  }
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int round = 0; round < 2; round++)
    for (int s = 0; s < nsets; s++) {
      const Set& S = sets[s];
      float best_a = 1e9f, best_b = 1e9f, sum_a = 0, sum_b = 0;
      const int reps = 20;
      for (int r = 0; r < reps + 3; r++) {
        float ms;
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k1_like, dim3(2048), dim3(256), 0, 0, coeffs[s], S.planes[0], S.planes[1], S.planes[2]);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 3) { best_a = ms < best_a ? ms : best_a; sum_a += ms; }
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(filter_like, dim3(4096), dim3(256), 0, 0, S.planes[0], S.planes[1], S.planes[2], S.tmp[0], S.tmp[1], S.tmp[2]);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 3) { best_b = ms < best_b ? ms : best_b; sum_b += ms; }
      }
      printf("round %d set %d: k1-like avg %.4f min %.4f ms   filter-like avg %.4f min %.4f ms   planes %p %p %p\n", round, s,
             sum_a / reps, best_a, sum_b / reps, best_b, (void*)S.planes[0], (void*)S.planes[1], (void*)S.planes[2]);
    }
  // ---- which half carries the spread, and planes carved from one allocation at chosen distances
  if (argc > 3) {
    int* sink = nullptr;
    CHECK(hipMalloc(&sink, 4));
    auto timeit = [&](auto launch) {
      float ms;
      for (int r = 0; r < 2; r++) launch();
      CHECK(hipEventRecord(e0));
      for (int r = 0; r < 10; r++) launch();
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      return ms / 10;
    };
    for (int s = 0; s < nsets; s++) {
      const Set& S = sets[s];
      const float w3 = timeit([&] { hipLaunchKernelGGL(write3, dim3(2048), dim3(256), 0, 0, S.planes[0], S.planes[1], S.planes[2]); });
      const float r3 = timeit([&] { hipLaunchKernelGGL(read3, dim3(2048), dim3(256), 0, 0, coeffs[s], sink); });
      const float w1 = timeit([&] { hipLaunchKernelGGL(write3, dim3(2048), dim3(256), 0, 0, S.planes[0], S.planes[0], S.planes[0]); });
      printf("halves set %d: write 3 planes %.4f  read 3 channels %.4f  write one plane three times %.4f\n", s, w3, r3, w1);
    }
    const size_t deltas_b[] = {0, 4194560};
    for (int a = 0; a < 2; a++) {
      float* arena = nullptr;
      CHECK(hipMalloc(&arena, 3 * (kPlane * 4 + ((size_t)65540 << 10)) + (64u << 20)));
      printf("arena %d interleaved planes: per batch (2 KB) %.4f  per 8 KB %.4f  per 256 KB %.4f  per 8 MB %.4f  per 64 MB %.4f\n", a,
             timeit([&] { hipLaunchKernelGGL(write3_interleaved<512>, dim3(2048), dim3(256), 0, 0, arena); }),
             timeit([&] { hipLaunchKernelGGL(write3_interleaved<2048>, dim3(2048), dim3(256), 0, 0, arena); }),
             timeit([&] { hipLaunchKernelGGL(write3_interleaved<65536>, dim3(2048), dim3(256), 0, 0, arena); }),
             timeit([&] { hipLaunchKernelGGL(write3_interleaved<2097152>, dim3(2048), dim3(256), 0, 0, arena); }),
             timeit([&] { hipLaunchKernelGGL(write3_interleaved<16777216>, dim3(2048), dim3(256), 0, 0, arena); }));
      for (size_t dk : deltas_b) {
        float* P[3];
        for (int c = 0; c < 3; c++) P[c] = arena + (size_t)c * (kPlane + dk / 4);
        const float w3 = timeit([&] { hipLaunchKernelGGL(write3, dim3(2048), dim3(256), 0, 0, P[0], P[1], P[2]); });
        const float k1 = timeit([&] { hipLaunchKernelGGL(k1_like, dim3(2048), dim3(256), 0, 0, coeffs[a], P[0], P[1], P[2]); });
        printf("arena %d planes %zu B + plane apart: write3 %.4f  k1-like %.4f\n", a, dk, w3, k1);
      }
    }
  }
  // ---- skews: the same buffers, every plane / tmp plane entered at its own offset (multiples of 256 KB inside 16 MB of
  // slack): what a calibration could pick from without allocating anything new
  if (argc > 2) {
    const int ntry = atoi(argv[2]);
    srand(12345);
    for (int s = 0; s < nsets && s < 3; s++) {
      float *bp[3], *bt[3];
      for (int c = 0; c < 3; c++) {
        CHECK(hipMalloc(&bp[c], kPlane * 4 + (16u << 20)));
        CHECK(hipMalloc(&bt[c], (kPlane + 8 * kSize) * 4 + (16u << 20)));
      }
      for (int t = 0; t < ntry; t++) {
        int sk[6];
        for (int k = 0; k < 6; k++) sk[k] = t == 0 ? 0 : rand() % 64;  // x 256 KB
        float* P[3];
        float* T[3];
        for (int c = 0; c < 3; c++) {
          P[c] = bp[c] + (size_t)sk[c] * 65536;
          T[c] = bt[c] + (size_t)sk[3 + c] * 65536;
        }
        float sum_a = 0, sum_b = 0;
        const int reps = 8;
        for (int r = 0; r < reps + 2; r++) {
          float ms;
          CHECK(hipEventRecord(e0));
          hipLaunchKernelGGL(k1_like, dim3(2048), dim3(256), 0, 0, coeffs[s], P[0], P[1], P[2]);
          CHECK(hipEventRecord(e1));
          CHECK(hipEventSynchronize(e1));
          CHECK(hipEventElapsedTime(&ms, e0, e1));
          if (r >= 2) sum_a += ms;
          CHECK(hipEventRecord(e0));
          hipLaunchKernelGGL(filter_like, dim3(4096), dim3(256), 0, 0, P[0], P[1], P[2], T[0], T[1], T[2]);
          CHECK(hipEventRecord(e1));
          CHECK(hipEventSynchronize(e1));
          CHECK(hipEventElapsedTime(&ms, e0, e1));
          if (r >= 2) sum_b += ms;
        }
        printf("skewed set %d try %2d: k1-like %.4f filter-like %.4f   skews x256KB %d %d %d | %d %d %d\n", s, t, sum_a / reps,
               sum_b / reps, sk[0], sk[1], sk[2], sk[3], sk[4], sk[5]);
      }
    }
  }
  return 0;
}
