// Micro-probe: cost of dependent vs independent VALU instructions for ONE wave on a SIMD (the regime the unsqueeze
// recurrence runs in).  hipcc --offload-arch=gfx950 -O3 valu_latency.hip -o valu_latency && ./valu_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 256
#define ITER 200
template <int CHAINS, int OP>
__global__ void probe(int* out, long long* cyc, int y) {
  int x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int k = 0; k < REP / CHAINS; k++) {
      if (OP == 0) {
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(x0) : "v"(y));
        if (CHAINS > 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x1) : "v"(y));
        if (CHAINS > 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x2) : "v"(y));
        if (CHAINS > 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x3) : "v"(y));
      } else if (OP == 1) {
        asm volatile("v_mul_hi_i32 %0, %0, %1" : "+v"(x0) : "v"(y));
        if (CHAINS > 1) asm volatile("v_mul_hi_i32 %0, %0, %1" : "+v"(x1) : "v"(y));
        if (CHAINS > 2) asm volatile("v_mul_hi_i32 %0, %0, %1" : "+v"(x2) : "v"(y));
        if (CHAINS > 3) asm volatile("v_mul_hi_i32 %0, %0, %1" : "+v"(x3) : "v"(y));
      } else if (OP == 2) {
        asm volatile("v_min3_i32 %0, %0, %1, %1" : "+v"(x0) : "v"(y));
        if (CHAINS > 1) asm volatile("v_min3_i32 %0, %0, %1, %1" : "+v"(x1) : "v"(y));
        if (CHAINS > 2) asm volatile("v_min3_i32 %0, %0, %1, %1" : "+v"(x2) : "v"(y));
        if (CHAINS > 3) asm volatile("v_min3_i32 %0, %0, %1, %1" : "+v"(x3) : "v"(y));
      } else if (OP == 3) {  // compare + select (vcc round trip)
        asm volatile("v_cmp_lt_i32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x0) : "v"(y) : "vcc");
        if (CHAINS > 1) asm volatile("v_cmp_lt_i32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x1) : "v"(y) : "vcc");
      } else if (OP == 4) {
        asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(x0) : "v"(y));
        if (CHAINS > 1) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(x1) : "v"(y));
      } else if (OP == 5) {
        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x0) : "v"(y));
        if (CHAINS > 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x1) : "v"(y));
        if (CHAINS > 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x2) : "v"(y));
        if (CHAINS > 3) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x3) : "v"(y));
      }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = x0 + x1 + x2 + x3;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int CHAINS, int OP>
void run(const char* name, int* out, long long* cyc, int lanes = 64) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe<CHAINS, OP><<<1, lanes>>>(out, cyc, 3);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<CHAINS, OP><<<1, lanes>>>(out, cyc, 3);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const int per = (OP == 3) ? 2 : 1;
  const double n = (double)ITER * (REP / CHAINS) * CHAINS * per;
  if (lanes != 64) printf("[%2d active lanes] ", lanes);
  printf("%-28s chains=%d  %.2f ns/instr  (%.2f counter ticks/instr, kernel %.3f ms)\n", name, CHAINS, ms * 1e6 / n, c / n, ms);
}
int main() {
  int* out; long long* cyc;
  hipMalloc(&out, 256); hipMalloc(&cyc, 8);
  run<1, 0>("v_add_u32", out, cyc); run<2, 0>("v_add_u32", out, cyc); run<4, 0>("v_add_u32", out, cyc);
  run<1, 1>("v_mul_hi_i32", out, cyc); run<2, 1>("v_mul_hi_i32", out, cyc); run<4, 1>("v_mul_hi_i32", out, cyc);
  run<1, 2>("v_min3_i32", out, cyc); run<4, 2>("v_min3_i32", out, cyc);
  run<1, 3>("v_cmp+v_cndmask", out, cyc); run<2, 3>("v_cmp+v_cndmask", out, cyc);
  run<1, 4>("v_mul_hi_u32_u24", out, cyc); run<2, 4>("v_mul_hi_u32_u24", out, cyc);
  run<1, 5>("v_mul_f32", out, cyc); run<4, 5>("v_mul_f32", out, cyc);
  // does a wave with fewer active lanes issue faster?  (EXEC = the low 32 / 16 lanes: a block of 32 / 16 threads)
  for (int lanes : {32, 16}) {
    run<1, 0>("v_add_u32", out, cyc, lanes); run<4, 0>("v_add_u32", out, cyc, lanes);
    run<1, 1>("v_mul_hi_i32", out, cyc, lanes); run<1, 2>("v_min3_i32", out, cyc, lanes);
  }
  return 0;
}
