// Microbenchmark (diagnostic): VALU issue rate per SIMD for 1, 2, 4 waves per SIMD; independent v_add_f32 / v_fma_f32 / mixes.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ void k_valu(int iters, float* sink, unsigned long long* cyc) {
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = (float)(threadIdx.x + i);
  const float c = 1.0001f, d = 0.5f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 pk[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) pk[i] = f2{a[2 * i], a[2 * i + 1]};
  const f2 pc = {c, c}, pd = {d, d};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
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
        else if (KIND == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pk[i & 7]) : "v"(pc));
        else if (KIND == 6) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(c), "v"(d));        // one dependent chain
        else if (KIND == 7) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i & 1]) : "v"(c), "v"(d));    // two chains
        else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i & 3]) : "v"(c), "v"(d));                    // four chains
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
  float s = 0; for (int i = 0; i < 16; ++i) s += a[i];
  for (int i = 0; i < 8; ++i) s += pk[i].x + pk[i].y;
  if (s == 123.456f) sink[0] = s;
}
template <int KIND>
void run(const char* name, int waves_per_cu, float* sink, unsigned long long* cyc) {
  const int iters = 2000;
  hipLaunchKernelGGL((k_valu<KIND>), dim3(256), dim3(64 * waves_per_cu), 0, 0, iters, sink, cyc);
  hipDeviceSynchronize();
  unsigned long long h[8192];
  const int n = 256 * waves_per_cu;
  hipMemcpy(h, cyc, n * 8, hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < n; ++i) s += (double)h[i];
  const double per_wave = s / n / (iters * 128.0);
  printf("%-10s %2d waves/CU (%d per SIMD): %5.2f cycles per instruction per wave -> %5.2f SIMD cycles per wave64 instruction\n",
         name, waves_per_cu, waves_per_cu / 4, per_wave, per_wave / (waves_per_cu / 4.0));
}
int main() {
  float* sink; hipMalloc((void**)&sink, 64);
  unsigned long long* cyc; hipMalloc((void**)&cyc, 8192 * 8);
  for (int w : {4, 8, 16}) {
    run<0>("v_add_f32", w, sink, cyc);
    run<1>("v_fma_f32", w, sink, cyc);
    run<3>("v_pk_add_f32", w, sink, cyc);
    run<4>("v_pk_fma_f32", w, sink, cyc);
    run<5>("v_pk_mul_f32", w, sink, cyc);
    run<6>("fma 1 chain", w, sink, cyc);
    run<7>("fma 2 chains", w, sink, cyc);
    run<8>("fma 4 chains", w, sink, cyc);
  }
  return 0;
}
