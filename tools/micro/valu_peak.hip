// Microbenchmark (diagnostic): chip-wide fp32 VALU throughput by wall clock (HIP events), the denominator of the VALU
// roofline in bench.py / DESIGN.md: independent v_add_f32 / v_mul_f32 / v_fma_f32 / v_pk_add_f32 / v_pk_fma_f32 streams,
// 16 accumulators per lane, 8 or 16 waves per CU, every CU busy.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ void __launch_bounds__(1024) k_valu(int iters, float* sink) {
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = (float)(threadIdx.x + i);
  const float c = 1.0001f, d = 0.5f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 pk[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) pk[i] = f2{a[2 * i], a[2 * i + 1]};
  const f2 pc = {c, c}, pd = {d, d};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(d));
        else if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
        else if (KIND == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        else if (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pk[i & 7]) : "v"(pd));
        else if (KIND == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pk[i & 7]) : "v"(pc), "v"(pd));
        else asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[i]) : "v"(d));
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += a[i];
  for (int i = 0; i < 8; ++i) s += pk[i].x + pk[i].y;
  if (s == 123.456f) sink[0] = s;
}
template <int KIND>
void run(const char* name, int threads, int lane_ops_per_instr, float* sink) {
  const int iters = 4000, grid = 256 * 4;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_valu<KIND>), dim3(grid), dim3(threads), 0, 0, 10, sink);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k_valu<KIND>), dim3(grid), dim3(threads), 0, 0, iters, sink);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double instr = (double)grid * (threads / 64) * iters * 128.0;             // wave instructions
  const double lane_ops = instr * 64.0 * lane_ops_per_instr;
  printf("%-13s %4d threads/WG: %7.2f ms  %6.2f T wave-lanes/s (instructions x 64)  %6.2f T lane-ops/s\n", name, threads, ms,
         instr * 64.0 / ms / 1e9, lane_ops / ms / 1e9);
}
int main() {
  float* sink; hipMalloc((void**)&sink, 64);
  for (int t : {512, 1024}) {
    run<0>("v_add_f32", t, 1, sink);
    run<5>("v_sub_f32", t, 1, sink);
    run<2>("v_mul_f32", t, 1, sink);
    run<1>("v_fma_f32", t, 1, sink);
    run<3>("v_pk_add_f32", t, 2, sink);
    run<4>("v_pk_fma_f32", t, 2, sink);
  }
  return 0;
}
