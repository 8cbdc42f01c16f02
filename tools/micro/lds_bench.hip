// Microbenchmark (diagnostic): LDS time of K1's instruction mix per wave and block, 8 waves per CU, no VALU work:
// 48 ds_write2_b64, 16 ds_write_b128, 52 ds_read2_b64, 32 ds_read_b64, 12 ds_read_b128, 32 ds_read_u8.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0 all, 1 writes only, 2 reads only, 3 all with single b64 instead of read2/write2
__global__ void __launch_bounds__(512) k_lds(int iters, float* sink, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned char* img = smem + wave * 18496;
  f2 a = {(float)lane, 1.f}, b = {2.f, (float)wave};
  f4 c = {1.f, 2.f, 3.f, 4.f};
  float acc = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE != 2) {
#pragma unroll
      for (int i = 0; i < 48; ++i) {   // two lane-contiguous b64 writes 1152 B apart (the compiler merges them into ds_write2_b64)
        f2* p = reinterpret_cast<f2*>(img + (i % 8) * 2304 + lane * 8 + (lane >> 4) * 16);
        if (MODE == 3) { asm volatile("ds_write_b64 %0, %1" :: "v"((unsigned)(size_t)p), "v"(a) : "memory");
                         asm volatile("ds_write_b64 %0, %1 offset:1152" :: "v"((unsigned)(size_t)p), "v"(b) : "memory"); }
        else asm volatile("ds_write2_b64 %0, %1, %2 offset1:144" :: "v"((unsigned)(size_t)p), "v"(a), "v"(b) : "memory");
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        f4* p = reinterpret_cast<f4*>(img + lane * 144 + (i % 8) * 16);
        asm volatile("ds_write_b128 %0, %1" :: "v"((unsigned)(size_t)p), "v"(c) : "memory");
      }
    }
    if (MODE != 1) {
#pragma unroll
      for (int i = 0; i < 52; ++i) {
        unsigned p = (unsigned)(size_t)(img + (i % 8) * 2304 + lane * 8 + (lane >> 4) * 16);
        f4 r;
        if (MODE == 3) { f2 r0, r1; asm volatile("ds_read_b64 %0, %1" : "=v"(r0) : "v"(p) : "memory");
                         asm volatile("ds_read_b64 %0, %1 offset:1152" : "=v"(r1) : "v"(p) : "memory"); r = f4{r0.x, r0.y, r1.x, r1.y}; }
        else asm volatile("ds_read2_b64 %0, %1 offset1:144" : "=v"(r) : "v"(p) : "memory");
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        acc += r.x;
      }
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        unsigned p = (unsigned)(size_t)(img + (i % 16) * 1152 + lane * 8 + (lane >> 4) * 16);
        f2 r;
        asm volatile("ds_read_b64 %0, %1" : "=v"(r) : "v"(p) : "memory");
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        acc += r.x;
      }
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        unsigned p = (unsigned)(size_t)(img + lane * 144 + (i % 8) * 16);
        f4 r;
        asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(p) : "memory");
        asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        acc += r.x;
      }
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        unsigned p = (unsigned)(size_t)(smem + 8 * 18496 + (i * 128 + lane) * 4);
        unsigned r;
        asm volatile("ds_read_u8 %0, %1" : "=v"(r) : "v"(p) : "memory");
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        acc += (float)r;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
  if (acc == 123.456f) sink[0] = acc;
}

template <int MODE>
void run(const char* name, float* sink, unsigned long long* cyc) {
  const int iters = 2000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((k_lds<MODE>), dim3(256), dim3(512), 8 * 18496 + 8192 + 1024, 0, iters, sink, cyc);
  hipDeviceSynchronize();
  unsigned long long h[2048];
  hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
  double s = 0; for (auto v : h) s += (double)v;
  printf("%-40s %8.0f cycles per wave and block (8 waves per CU)\n", name, s / 2048 / iters);
}
int main() {
  float* sink; hipMalloc((void**)&sink, 64);
  unsigned long long* cyc; hipMalloc((void**)&cyc, 2048 * 8);
  run<0>("K1 LDS mix", sink, cyc);
  run<1>("writes only", sink, cyc);
  run<2>("reads only", sink, cyc);
  run<3>("mix, b64 pairs instead of read2/write2", sink, cyc);
  return 0;
}
