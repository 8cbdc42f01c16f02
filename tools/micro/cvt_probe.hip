// Probe (diagnostic): what v_cvt_pk_u8_f32 does with fractions, negatives, overflow and NaN, and whether an fp32 add issued
// between two s_setreg of MODE.FP_ROUND rounds toward -infinity -- the two facts the lean digitiser (frbch_quantise_fast)
// rests on: floor(t + 0.5) evaluated exactly as floor(RTN(t + 0.5)) and a one-instruction clip + convert + pack.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <cstdint>
__global__ void k_probe(const float* in, uint32_t* cvt, float* rtn, float* rne, int n) {
  const int i = threadIdx.x;
  if (i >= n) return;
  const float x = in[i];
  uint32_t acc = 0xAAAAAAAAu;
  asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(acc) : "v"(x));   // byte 1 of acc
  cvt[i] = acc;
  float y, z;
  asm volatile(
      "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 2\n\t"
      "s_nop 1\n\t"
      "v_add_f32 %0, 0.5, %2\n\t"
      "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0\n\t"
      "s_nop 1\n\t"
      "v_add_f32 %1, 0.5, %2"
      : "=&v"(y), "=&v"(z)
      : "v"(x));
  rtn[i] = y;
  rne[i] = z;
}
int main() {
  float h[64];
  int n = 0;
  const float vals[] = {-2.f, -0.5f, -0.f, 0.f, 0.4f, 0.5f, 0.6f, 0.99999994f, 1.f, 1.5f, 2.5f, 3.5f, 127.5f, 254.5f, 254.99998f,
                        255.f, 255.4f, 255.5f, 256.f, 300.f, 1e9f, INFINITY, -INFINITY, NAN};
  for (float v : vals) h[n++] = v;
  h[n++] = 0.5f - ldexpf(1.f, -25);        // t + 0.5 = 1 - 2^-25: RNE gives 1.0 (wrong floor), RTN gives 1 - 2^-24
  h[n++] = 1.5f - ldexpf(1.f, -24);
  h[n++] = 100.5f - ldexpf(1.f, -17);
  h[n++] = -0.5f - ldexpf(1.f, -25);
  float *d_in, *d_rtn, *d_rne;
  uint32_t* d_cvt;
  hipMalloc((void**)&d_in, 256); hipMalloc((void**)&d_rtn, 256); hipMalloc((void**)&d_rne, 256); hipMalloc((void**)&d_cvt, 256);
  hipMemcpy(d_in, h, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d_in, d_cvt, d_rtn, d_rne, n);
  uint32_t c[64];
  float y[64], z[64];
  hipMemcpy(c, d_cvt, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(y, d_rtn, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(z, d_rne, n * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) {
    uint32_t bx, by, bz;
    memcpy(&bx, &h[i], 4); memcpy(&by, &y[i], 4); memcpy(&bz, &z[i], 4);
    printf("x=%-14.9g (%08x)  cvt_pk_u8 -> %08x (byte %3u)   x+0.5 RTN %.9g (%08x)  RNE %.9g (%08x)\n", h[i], bx, c[i], (c[i] >> 8) & 255u,
           y[i], by, z[i], bz);
  }
  return 0;
}
