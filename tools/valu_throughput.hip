// Micro-probe: vector-ALU THROUGHPUT at full occupancy (8 waves per SIMD, 8 independent chains per lane): cycles a SIMD
// spends per wave64 instruction for plain and packed f32 operations -- the figure behind DESIGN.md's instruction floor.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_throughput.hip -o tools/valu_throughput && tools/valu_throughput
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
#define ITER 400
typedef float float2_t __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(256) void probe(float* out, float y) {
  float a[8];
  float2_t p[8];
  for (int i = 0; i < 8; i++) {
    a[i] = threadIdx.x + i;
    p[i] = float2_t{(float)threadIdx.x + i, (float)i};
  }
  const float2_t y2 = float2_t{y, y};
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int k = 0; k < REP / 8; k++) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(y));
        if (OP == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(y));
        if (OP == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(y2));
        if (OP == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(y2));
        if (OP == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(y2));
        if (OP == 5) asm volatile("v_sub_f32 %0, %0, %1\n v_add_f32 %0, |%0|, %1" : "+v"(a[i]) : "v"(y));
        if (OP == 6) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(y));
        if (OP == 7) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP>
void run(const char* name, float* out, int cus, double ghz) {
  const int wgs = cus * 8;  // 8 workgroups of 4 waves per CU = 8 waves per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe<OP><<<wgs, 256>>>(out, 1.0001f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<OP><<<wgs, 256>>>(out, 1.0001f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const int per = OP == 5 ? 2 : 1;
  const double instr_per_simd = (double)ITER * REP * per * 8;  // 8 waves per SIMD
  printf("%-34s %.3f ms  -> %.2f cycles per wave64 instruction and SIMD at %.1f GHz\n", name, ms,
         ms * 1e-3 * ghz * 1e9 / instr_per_simd, ghz);
}
int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const double ghz = prop.clockRate / 1e6;
  printf("%s: %d CUs, %.2f GHz\n", prop.name, cus, ghz);
  float* out;
  hipMalloc(&out, (size_t)cus * 8 * 256 * 4);
  run<0>("v_fma_f32", out, cus, ghz); run<1>("v_add_f32", out, cus, ghz);
  run<2>("v_pk_fma_f32 (2 lanes' worth)", out, cus, ghz); run<3>("v_pk_add_f32", out, cus, ghz);
  run<4>("v_pk_mul_f32", out, cus, ghz); run<5>("v_sub_f32 + v_add_f32 |x|", out, cus, ghz);
  run<6>("v_max_f32", out, cus, ghz); run<7>("v_mov_b32_dpp row_shr:1", out, cus, ghz);
  return 0;
}
