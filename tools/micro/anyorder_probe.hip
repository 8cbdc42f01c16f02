// Probe (diagnostic): can two kernels of ONE stream run at the same time on this part?  hipExtLaunchKernelGGL with
// hipExtAnyOrderLaunch drops the barrier bit of the second dispatch (hip_ext.h says "not supported on GFX9xx" for the module
// form); measured here: a 64-workgroup spinner, then a second one, (a) plain, (b) the second with the flag, (c) on two streams.
// Also prints what the start / stop events of the extended launch report, against hipEventRecord around the launch.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void __launch_bounds__(256) spin(unsigned long long ticks, unsigned long long* out) {
  const unsigned long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) out[blockIdx.x] = __builtin_readcyclecounter() - t0;
}
static double wall_ms(hipStream_t a, hipStream_t b, int mode, unsigned long long ticks, unsigned long long* buf) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipDeviceSynchronize();
  hipEventRecord(e0, a);
  hipExtLaunchKernelGGL(spin, dim3(64), dim3(256), 0, a, nullptr, nullptr, 0, ticks, buf);
  if (mode == 2) {
    hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, b, ticks, buf + 64);
    hipEvent_t eb; hipEventCreateWithFlags(&eb, hipEventDisableTiming);
    hipEventRecord(eb, b);
    hipStreamWaitEvent(a, eb, 0);
  } else {
    hipExtLaunchKernelGGL(spin, dim3(64), dim3(256), 0, a, nullptr, nullptr, mode == 1 ? hipExtAnyOrderLaunch : 0, ticks, buf + 64);
  }
  hipEventRecord(e1, a);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main() {
  unsigned long long* buf;
  hipMalloc((void**)&buf, 4096);
  hipStream_t a, b;
  hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
  const unsigned long long ticks = 100000000ull / 50;   // s_memtime / cycle counter: ~100 MHz -> 20 ms?  printed below
  for (int rep = 0; rep < 2; ++rep) {
    printf("same stream, plain:          %8.3f ms\n", wall_ms(a, b, 0, ticks, buf));
    printf("same stream, any-order flag: %8.3f ms\n", wall_ms(a, b, 1, ticks, buf));
    printf("two streams:                 %8.3f ms\n", wall_ms(a, b, 2, ticks, buf));
  }
  // start / stop events of the extended launch
  hipEvent_t s0, s1, r0, r1;
  hipEventCreate(&s0); hipEventCreate(&s1); hipEventCreate(&r0); hipEventCreate(&r1);
  hipDeviceSynchronize();
  hipEventRecord(r0, a);
  hipExtLaunchKernelGGL(spin, dim3(64), dim3(256), 0, a, s0, s1, 0, ticks, buf);
  hipEventRecord(r1, a);
  hipEventSynchronize(r1);
  float m0 = 0, m1 = 0;
  hipEventElapsedTime(&m0, s0, s1);
  hipEventElapsedTime(&m1, r0, r1);
  printf("one spinner: start/stop events of the launch %8.3f ms, hipEventRecord around it %8.3f ms\n", m0, m1);
  return 0;
}
