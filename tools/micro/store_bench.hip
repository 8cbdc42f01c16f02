// Microbenchmark (diagnostic, not part of the product): sustained global-store throughput of K1's spill pattern.
// Each workgroup (8 waves) owns a slab of 128 KB + pad and rewrites it `iters` times: per wave 32 (or 16) store
// instructions of W bytes per lane to 16 regions 8 KB apart, optionally with ALU work between bursts.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int W, int NT_HINT>   // W = bytes per lane: 4, 8, 16
__global__ void __launch_bounds__(512) k_store(float* base, size_t slab_floats, int nslab, int iters, int work, float* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc = (float)threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    const int slab = (blockIdx.x + it * gridDim.x) % nslab;
    char* s = reinterpret_cast<char*>(base + (size_t)slab * slab_floats);
    for (int w = 0; w < work; ++w) acc = __builtin_fmaf(acc, 1.000001f, 0.5f);
    // 128 KB per workgroup and iteration, 16 KB per wave: regions a = 0..15 of 8 KB, inside each the wave's share
    constexpr int PER_WAVE_REGION = 1024;           // bytes of a region written by one wave
    constexpr int NST = PER_WAVE_REGION / (64 * W); // store instructions per region and wave
#pragma unroll
    for (int a = 0; a < 16; ++a) {
#pragma unroll
      for (int j = 0; j < NST; ++j) {
        char* p = s + a * 8192 + wave * PER_WAVE_REGION + j * 64 * W + lane * W;
        if constexpr (W == 4) {
          if (NT_HINT) __builtin_nontemporal_store(acc, reinterpret_cast<float*>(p)); else *reinterpret_cast<float*>(p) = acc;
        } else if constexpr (W == 8) {
          typedef float f2 __attribute__((ext_vector_type(2)));
          f2 v = {acc, acc};
          if (NT_HINT) __builtin_nontemporal_store(v, reinterpret_cast<f2*>(p)); else *reinterpret_cast<f2*>(p) = v;
        } else {
          typedef float f4 __attribute__((ext_vector_type(4)));
          f4 v = {acc, acc, acc, acc};
          if (NT_HINT) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(p)); else *reinterpret_cast<f4*>(p) = v;
        }
      }
    }
  }
  if (acc == 123.456f) sink[0] = acc;
}

template <int W, int NT_HINT>
void run(const char* name, float* buf, size_t slab_floats, int nslab, int work, float* sink) {
  const int iters = 200;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_store<W, NT_HINT>), dim3(256), dim3(512), 0, 0, buf, slab_floats, nslab, 20, work, sink);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_store<W, NT_HINT>), dim3(256), dim3(512), 0, 0, buf, slab_floats, nslab, iters, work, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = 256.0 * iters * 131072.0;
  printf("%-28s work %5d nslab %6d : %7.3f ms  %7.1f GB/s  %6.2f B/clk/CU@2.05GHz  %6.2f us/iter\n", name, work, nslab, ms, bytes / ms / 1e6,
         bytes / ms / 1e6 * 1e9 / 256 / 2.05e9 / 1.0, ms * 1e3 / iters);
}

int main() {
  const size_t slab_floats = (131072 + 2048 + 128) / 4;
  const int nslab_big = 256 * 64;   // 2.1 GB: streams through HBM
  float* buf; hipMalloc((void**)&buf, slab_floats * 4 * (size_t)nslab_big);
  float* sink; hipMalloc((void**)&sink, 64);
  for (int work : {0, 2000}) {
    for (int nslab : {256, nslab_big}) {
      run<4, 0>("dword   (256 B/instr)", buf, slab_floats, nslab, work, sink);
      run<8, 0>("dwordx2 (512 B/instr)", buf, slab_floats, nslab, work, sink);
      run<16, 0>("dwordx4 (1 KB/instr)", buf, slab_floats, nslab, work, sink);
      run<8, 1>("dwordx2 nt", buf, slab_floats, nslab, work, sink);
      run<16, 1>("dwordx4 nt", buf, slab_floats, nslab, work, sink);
    }
  }
  return 0;
}
