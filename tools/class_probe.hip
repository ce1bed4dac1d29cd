// Do plane-sized buffers fall into CLASSES that predict how fast several of them are written together?
// N buffers of exactly 2^28 bytes, no spacers.  (1) pairwise times against every other buffer -> classes by thresholding
// against buffer 0; (2) triple times grouped by class composition.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/class_probe tools/class_probe.hip && /tmp/class_probe [buffers]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
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
  const int nb = argc > 1 ? atoi(argv[1]) : 16;
  std::vector<float*> bufs(nb);
  for (int i = 0; i < nb; i++) CHECK(hipMalloc(&bufs[i], kPlane * 4));
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
  std::vector<std::vector<float>> pair(nb, std::vector<float>(nb, 0.f));
  for (int i = 0; i < nb; i++)
    for (int j = 0; j < nb; j++)
      if (i != j) pair[i][j] = timeit(bufs[i], bufs[j], bufs[j]);
  for (int i = 0; i < nb; i++) {
    printf("%p with %2d:", (void*)bufs[i], i);
    for (int j = 0; j < nb; j++) printf(" %.3f", pair[i][j]);
    printf("\n");
  }
  // classes: j is of buffer 0's class if writing it next to buffer 0 is slow (above the middle of the observed range)
  float lo = 1e9f, hi = 0.f;
  for (int j = 1; j < nb; j++) lo = std::min(lo, pair[0][j]), hi = std::max(hi, pair[0][j]);
  const float mid = 0.5f * (lo + hi);
  std::vector<int> cls(nb, 0);
  for (int j = 1; j < nb; j++) cls[j] = pair[0][j] > mid ? 0 : 1;
  printf("classes (range %.3f .. %.3f):", lo, hi);
  for (int j = 0; j < nb; j++) printf(" %d", cls[j]);
  printf("\n");
  // ---- full clustering: a buffer joins the first class whose representative it is slow with
  {
    std::vector<int> rep;  // representative buffer of each class
    std::vector<int> c2(nb, -1);
    for (int j = 0; j < nb; j++) {
      for (size_t k = 0; k < rep.size() && c2[j] < 0; k++)
        if (pair[rep[k]][j] > mid) c2[j] = (int)k;
      if (c2[j] < 0) {
        c2[j] = (int)rep.size();
        rep.push_back(j);
      }
    }
    printf("clusters:");
    for (int j = 0; j < nb; j++) printf(" %d", c2[j]);
    printf("\n");
    int* coeffs[2];
    for (int c = 0; c < 2; c++) {
      CHECK(hipMalloc(&coeffs[c], (size_t)1024 * 3 * 65536 * 4));
      CHECK(hipMemset(coeffs[c], 0, (size_t)1024 * 3 * 65536 * 4));
    }
    auto time_k1 = [&](int* cf, int a, int b, int c) {
      float ms;
      hipLaunchKernelGGL(k1_like, dim3(2048), dim3(256), 0, 0, cf, bufs[a], bufs[b], bufs[c]);
      CHECK(hipEventRecord(e0));
      for (int r = 0; r < 6; r++) hipLaunchKernelGGL(k1_like, dim3(2048), dim3(256), 0, 0, cf, bufs[a], bufs[b], bufs[c]);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      return ms / 6;
    };
    auto time_f = [&](int a, int b, int c, int x, int y, int z) {
      float ms;
      hipLaunchKernelGGL(filter_like, dim3(4096), dim3(256), 0, 0, bufs[a], bufs[b], bufs[c], bufs[x], bufs[y], bufs[z]);
      CHECK(hipEventRecord(e0));
      for (int r = 0; r < 6; r++) hipLaunchKernelGGL(filter_like, dim3(4096), dim3(256), 0, 0, bufs[a], bufs[b], bufs[c], bufs[x], bufs[y], bufs[z]);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      return ms / 6;
    };
    // sequential sets as the library's plain path makes them: plane, tmp, plane, tmp, plane, tmp
    for (int s0 = 0; s0 + 5 < nb; s0 += 6)
      printf("sequential set at %2d (clusters %d%d%d / %d%d%d): k1-like %.4f %.4f  filter-like %.4f\n", s0, c2[s0], c2[s0 + 2], c2[s0 + 4],
             c2[s0 + 1], c2[s0 + 3], c2[s0 + 5], time_k1(coeffs[0], s0, s0 + 2, s0 + 4), time_k1(coeffs[1], s0, s0 + 2, s0 + 4),
             time_f(s0, s0 + 2, s0 + 4, s0 + 1, s0 + 3, s0 + 5));
    // class-based: planes from different clusters, tmp from clusters different from their plane's and from each other
    const int ncl = (int)rep.size();
    std::vector<std::vector<int>> members((size_t)ncl);
    for (int j = 0; j < nb; j++) members[(size_t)c2[j]].push_back(j);
    if (ncl >= 2) {
      auto take = [&](int cl) {
        int b = members[(size_t)cl].back();
        members[(size_t)cl].pop_back();
        return b;
      };
      auto pick_cl = [&](int avoid1, int avoid2) {  // the fullest cluster that is not one of the two
        int best = -1;
        for (int k = 0; k < ncl; k++)
          if (k != avoid1 && k != avoid2 && !members[(size_t)k].empty() && (best < 0 || members[(size_t)k].size() > members[(size_t)best].size())) best = k;
        if (best < 0)
          for (int k = 0; k < ncl; k++)
            if (k != avoid1 && !members[(size_t)k].empty() && (best < 0 || members[(size_t)k].size() > members[(size_t)best].size())) best = k;
        if (best < 0)
          for (int k = 0; k < ncl; k++)
            if (!members[(size_t)k].empty()) best = k;
        return best;
      };
      int pc[3], tc[3], P[3], T[3];
      pc[0] = pick_cl(-1, -1); P[0] = take(pc[0]);
      pc[1] = pick_cl(pc[0], -1); P[1] = take(pc[1]);
      pc[2] = pick_cl(pc[0], pc[1]); P[2] = take(pc[2]);
      // tmp c: not its plane's cluster; rotate the planes' clusters
      tc[0] = pick_cl(pc[0], -1) ; T[0] = take(tc[0]);
      tc[1] = pick_cl(pc[1], tc[0]); T[1] = take(tc[1]);
      tc[2] = pick_cl(pc[2], tc[1]); T[2] = take(tc[2]);
      printf("class-based set: planes %d %d %d (clusters %d%d%d), tmp %d %d %d (clusters %d%d%d): write3 %.4f  k1-like %.4f %.4f  filter-like %.4f\n",
             P[0], P[1], P[2], pc[0], pc[1], pc[2], T[0], T[1], T[2], tc[0], tc[1], tc[2], timeit(bufs[P[0]], bufs[P[1]], bufs[P[2]]),
             time_k1(coeffs[0], P[0], P[1], P[2]), time_k1(coeffs[1], P[0], P[1], P[2]), time_f(P[0], P[1], P[2], T[0], T[1], T[2]));
    }
  }
  std::map<std::string, std::vector<float>> by;
  for (int i = 0; i < nb; i++)
    for (int j = i + 1; j < nb; j++)
      for (int k = j + 1; k < nb; k++) {
        int n1 = cls[i] + cls[j] + cls[k];
        by[std::string(3 - n1, 'A') + std::string(n1, 'B')].push_back(timeit(bufs[i], bufs[j], bufs[k]));
      }
  for (auto& kv : by) {
    std::sort(kv.second.begin(), kv.second.end());
    printf("triples %s: %zu  min %.4f median %.4f max %.4f\n", kv.first.c_str(), kv.second.size(), kv.second[0],
           kv.second[kv.second.size() / 2], kv.second.back());
  }
  return 0;
}
