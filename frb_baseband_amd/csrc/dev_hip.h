// gfx950 device layer for the translation units of libfrbch (frbch_internal.h): kernel macro layer, launches, memory, events.
#ifndef FRBCH_DEV_HIP_H
#define FRBCH_DEV_HIP_H
#include <hip/hip_runtime.h>
#include <stdio.h>

#define FRBCH_BACKEND_NAME "hip-gfx950"
#define DEVFN __device__
#define KERNEL(name, PT) extern "C" __global__ void __launch_bounds__(256) name(PT p)
#define K_PROLOGUE                                                                  \
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];              \
  const int bx = blockIdx.x, by = blockIdx.y, nthr = blockDim.x
#define PHASE for (int tid = threadIdx.x, once_ = 1; once_; once_ = 0)
#define SYNC __syncthreads()
// Separately rounded fp32 product and sum for the rescale / digitiser arithmetic (the C expression
// `x * digi_scale + digi_mean` of a CPU build).  HIP's __fmul_rn / __fadd_rn are plain `*` / `+` and hipcc contracts
// them into one FMA (found by the bit-exact digitiser test: codes at exact .5 ties came out one lower); these do not
// carry the `contract` flag, so the backend cannot fuse them.
static __device__ __forceinline__ float frbch_fmul_rn(float a, float b) {
#pragma clang fp contract(off)
  const float r = a * b;
  return r;
}
static __device__ __forceinline__ float frbch_fadd_rn(float a, float b) {
#pragma clang fp contract(off)
  const float r = a + b;
  return r;
}
#define FMUL_RN(a, b) frbch_fmul_rn((a), (b))
#define FADD_RN(a, b) frbch_fadd_rn((a), (b))
// streaming (read-once / write-once) 16-byte accesses
typedef float frbch_nf4 __attribute__((ext_vector_type(4)));
#define LOAD_F4_STREAM(dst, ptr)                                                   \
  do {                                                                             \
    const frbch_nf4 t_ = __builtin_nontemporal_load((const frbch_nf4*)(ptr));      \
    (dst).v[0] = t_.x; (dst).v[1] = t_.y; (dst).v[2] = t_.z; (dst).v[3] = t_.w;    \
  } while (0)
#define STORE_U32_STREAM(ptr, val) __builtin_nontemporal_store((uint32_t)(val), (uint32_t*)(ptr))

// atomics of the post-filterbank kernels (kernels_post.inc); double-precision phase arithmetic must not be contracted
#define ATOMIC_ADD_U64(ptr, v) atomicAdd((unsigned long long*)(ptr), (unsigned long long)(v))
#define ATOMIC_ADD_U32(ptr, v) atomicAdd((unsigned int*)(ptr), (unsigned int)(v))
#define ATOMIC_ADD_F64(ptr, v) atomicAdd((double*)(ptr), (double)(v))
#define POST_NO_CONTRACT _Pragma("clang fp contract(off)")

typedef hipStream_t dev_stream_t;
typedef hipEvent_t dev_event_t;

#define DEV_LAUNCH(kern, gx, gy, nthr, lds, stream, params) \
  hipLaunchKernelGGL(kern, dim3((unsigned)(gx), (unsigned)(gy)), dim3((unsigned)(nthr)), (lds), (stream), (params))

static inline const char* dev_last_error_string() { return hipGetErrorString(hipGetLastError()); }
static inline int dev_count() {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}
static inline int dev_set(int d) { return hipSetDevice(d) == hipSuccess ? 0 : -1; }
static inline int dev_get() {
  int d = -1;
  return hipGetDevice(&d) == hipSuccess ? d : -1;
}
static inline int dev_arch_ok(int d, char* name, size_t cap, size_t* lds_limit) {
  hipDeviceProp_t pr;
  if (hipGetDeviceProperties(&pr, d) != hipSuccess) return 0;
  snprintf(name, cap, "%s", pr.gcnArchName);
  *lds_limit = pr.maxSharedMemoryPerMultiProcessor ? (size_t)pr.maxSharedMemoryPerMultiProcessor : 65536;
  if (*lds_limit > 160 * 1024) *lds_limit = 160 * 1024;
  return 1;
}
template <class K>
static inline int dev_allow_lds(K kern, size_t bytes) {
  return hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess ? 0 : -1;
}
static inline int dev_malloc(void** p, size_t n) { return hipMalloc(p, n ? n : 1) == hipSuccess ? 0 : -1; }
static inline void dev_free(void* p) { if (p) (void)hipFree(p); }
static inline int dev_h2d(void* d, const void* h, size_t n, dev_stream_t s) {
  return hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s) == hipSuccess ? 0 : -1;
}
static inline int dev_d2h(void* h, const void* d, size_t n, dev_stream_t s) {
  return hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s) == hipSuccess ? 0 : -1;
}
static inline int dev_d2d(void* d, const void* s_, size_t n, dev_stream_t s) {
  return hipMemcpyAsync(d, s_, n, hipMemcpyDeviceToDevice, s) == hipSuccess ? 0 : -1;
}
static inline int dev_copy2d(void* d, size_t dpitch, const void* s_, size_t spitch, size_t width, size_t height, dev_stream_t s) {
  return hipMemcpy2DAsync(d, dpitch, s_, spitch, width, height, hipMemcpyDeviceToDevice, s) == hipSuccess ? 0 : -1;
}
static inline int dev_memset(void* d, int v, size_t n, dev_stream_t s) {
  return hipMemsetAsync(d, v, n, s) == hipSuccess ? 0 : -1;
}
static inline int dev_memset32(void* d, uint32_t v, size_t nwords, dev_stream_t s) {
  return hipMemsetD32Async((hipDeviceptr_t)d, (int)v, nwords, s) == hipSuccess ? 0 : -1;
}
static inline int dev_sync(dev_stream_t s) { return hipStreamSynchronize(s) == hipSuccess ? 0 : -1; }
static inline int dev_stream_create(dev_stream_t* s) { return hipStreamCreateWithFlags(s, hipStreamNonBlocking) == hipSuccess ? 0 : -1; }
static inline void dev_stream_destroy(dev_stream_t s) { (void)hipStreamDestroy(s); }
// a stream whose kernels run only on the compute units whose bit is set in `mask` (bit i: XCD i % 8)
static inline int dev_stream_create_masked(dev_stream_t* s, const uint32_t* mask, uint32_t words) {
  return hipExtStreamCreateWithCUMask(s, words, mask) == hipSuccess ? 0 : -1;
}
static inline int dev_cu_count(int d) {
  int n = 0;
  return hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) == hipSuccess ? n : 0;
}
// ordering-only events (no timestamps) and cross-stream waits
static inline int dev_event_create_sync(hipEvent_t* e) { return hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess ? 0 : -1; }
static inline int dev_stream_wait(dev_stream_t s, hipEvent_t e) { return hipStreamWaitEvent(s, e, 0) == hipSuccess ? 0 : -1; }
static inline int dev_event_sync(hipEvent_t e) { return hipEventSynchronize(e) == hipSuccess ? 0 : -1; }
static inline int dev_check_launch() { return hipGetLastError() == hipSuccess ? 0 : -1; }
static inline int dev_host_alloc(void** p, size_t n) { return hipHostMalloc(p, n, hipHostMallocDefault) == hipSuccess ? 0 : -1; }
static inline void dev_host_free(void* p) { if (p) (void)hipHostFree(p); }
static inline int dev_event_create(dev_event_t* e) { return hipEventCreate(e) == hipSuccess ? 0 : -1; }
static inline void dev_event_destroy(dev_event_t e) { (void)hipEventDestroy(e); }
static inline void dev_event_record(dev_event_t e, dev_stream_t s) { (void)hipEventRecord(e, s); }
static inline float dev_event_ms(dev_event_t a, dev_event_t b) {
  float ms = 0.f;
  (void)hipEventSynchronize(b);
  (void)hipEventElapsedTime(&ms, a, b);
  return ms;
}
#endif
