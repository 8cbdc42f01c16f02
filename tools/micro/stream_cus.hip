// How much HBM bandwidth can a streaming kernel pull on a SUBSET of the CUs (a CU-masked stream)?  The digitiser of the first
// rescale interval is such a kernel (16 B in, 4 B out per lane and trip).  Sweeps the number of CUs, loads in flight per
// thread and waves per CU; prints GB/s in total and per CU.
//   hipcc --offload-arch=gfx950 -O3 stream_cus.hip -o stream_cus && ./stream_cus
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float nf4 __attribute__((ext_vector_type(4)));

template <int U>
__global__ void __launch_bounds__(256) stream_q(const nf4* __restrict__ in, uint32_t* __restrict__ out, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += U * stride) {
    nf4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = i0 + u * stride;
      if (i < n) v[u] = __builtin_nontemporal_load(in + i);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = i0 + u * stride;
      if (i < n) {
        const uint32_t c = (uint32_t)(v[u].x + 0.5f) | ((uint32_t)(v[u].y + 0.5f) << 8) | ((uint32_t)(v[u].z + 0.5f) << 16) | ((uint32_t)(v[u].w + 0.5f) << 24);
        __builtin_nontemporal_store(c, out + i);
      }
    }
  }
}

int main() {
  int ncu = 0;
  hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  const size_t n = (size_t)1 << 28;                 // 4 GiB in, 1 GiB out
  nf4* in; uint32_t* out;
  hipMalloc(&in, n * 16); hipMalloc(&out, n * 4);
  hipMemset(in, 0, n * 16);
  const int words = (ncu + 31) / 32;
  const int cus[] = {32, 48, 64, 96, 128, 192, 256};
  for (int nc : cus) {
    std::vector<unsigned> m(words, 0);
    for (int i = ncu - nc; i < ncu; ++i) m[i >> 5] |= 1u << (i & 31);    // the LAST nc bits (a back lane)
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, words, m.data()) != hipSuccess) { printf("mask failed\n"); return 1; }
    for (int wpc : {8, 16, 32}) {                    // waves per CU = 4 x workgroups per CU
      for (int U : {4, 8, 16}) {
        const int grid = nc * wpc / 4;
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
          hipEventRecord(a, s);
          if (U == 4) hipLaunchKernelGGL(stream_q<4>, dim3(grid), dim3(256), 0, s, in, out, n);
          else if (U == 8) hipLaunchKernelGGL(stream_q<8>, dim3(grid), dim3(256), 0, s, in, out, n);
          else hipLaunchKernelGGL(stream_q<16>, dim3(grid), dim3(256), 0, s, in, out, n);
          hipEventRecord(b, s);
          hipEventSynchronize(b);
          float ms; hipEventElapsedTime(&ms, a, b);
          if (ms < best) best = ms;
        }
        const double gb = (double)n * 20 / 1e9;
        printf("CUs %3d  waves/CU %2d  loads in flight %2d  %7.3f ms  %7.1f GB/s  %6.1f GB/s per CU\n", nc, wpc, U, best, gb / best * 1e3, gb / best * 1e3 / nc);
      }
    }
    hipStreamDestroy(s);
  }
  return 0;
}
