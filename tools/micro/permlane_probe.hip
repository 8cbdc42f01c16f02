// Probe (diagnostic): lane mapping of v_permlane32_swap / v_permlane16_swap on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  const unsigned l = threadIdx.x;
  unsigned a = 1000 + l, b = 2000 + l;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[l] = r[0]; out[64 + l] = r[1];
  auto s = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[128 + l] = s[0]; out[192 + l] = s[1];
}
int main() {
  unsigned* d; hipMalloc((void**)&d, 256 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char* nm[4] = {"swap32 first ", "swap32 second", "swap16 first ", "swap16 second"};
  for (int r = 0; r < 4; ++r) { printf("%s:", nm[r]); for (int i = 0; i < 64; i += 8) printf(" %u", h[r * 64 + i]); printf("\n"); }
  return 0;
}
