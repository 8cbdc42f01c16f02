// TEST INFRASTRUCTURE ONLY: host emulation of the device layer (see csrc/dev_hip.h) so that the
// CPU unit tests can run the engine's host logic and the generic kernels' index arithmetic
// without a GPU.  Never built into, or loaded by, the frb_baseband_amd package.
#ifndef FRBCH_DEV_EMU_H
#define FRBCH_DEV_EMU_H
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#define FRBCH_BACKEND_NAME "host-emulator(test-only)"
#define FRBCH_NO_FAST 1   /* the register-level gfx950 kernels are HIP only */
#define DEVFN static
#define KERNEL(name, PT) static void name(PT p, int bx, int by, int nthr, unsigned char* smem)
#define K_PROLOGUE ((void)0)
#define PHASE for (int tid = 0; tid < nthr; ++tid)
#define SYNC ((void)0)
#define FMUL_RN(a, b) ((a) * (b))   /* built with -ffp-contract=off */
#define FADD_RN(a, b) ((a) + (b))
#define LOAD_F4_STREAM(dst, ptr) ((dst) = *(const f4*)(ptr))
#define STORE_U32_STREAM(ptr, val) (*(uint32_t*)(ptr) = (uint32_t)(val))

static inline void emu_atomic_add_u64(unsigned long long* p, unsigned long long v) {
#pragma omp atomic
  *p += v;
}
static inline void emu_atomic_add_u32(unsigned int* p, unsigned int v) {
#pragma omp atomic
  *p += v;
}
static inline void emu_atomic_add_f64(double* p, double v) {
#pragma omp atomic
  *p += v;
}
#define ATOMIC_ADD_U64(ptr, v) emu_atomic_add_u64((unsigned long long*)(ptr), (unsigned long long)(v))
#define ATOMIC_ADD_U32(ptr, v) emu_atomic_add_u32((unsigned int*)(ptr), (unsigned int)(v))
#define ATOMIC_ADD_F64(ptr, v) emu_atomic_add_f64((double*)(ptr), (double)(v))
#define POST_NO_CONTRACT   /* built with -ffp-contract=off */

typedef void* dev_stream_t;
typedef int dev_event_t;

template <class PT>
static void emu_launch(void (*k)(PT, int, int, int, unsigned char*), long gx, long gy, int nthr, size_t lds, PT p) {
#pragma omp parallel
  {
    std::vector<unsigned char> smem(lds + 64);
#pragma omp for collapse(2) schedule(dynamic)
    for (long by = 0; by < gy; ++by)
      for (long bx = 0; bx < gx; ++bx) k(p, (int)bx, (int)by, nthr, smem.data());
  }
}
#define DEV_LAUNCH(kern, gx, gy, nthr, lds, stream, params) emu_launch(kern, (long)(gx), (long)(gy), (int)(nthr), (size_t)(lds), params)

static inline const char* dev_last_error_string() { return "emulator"; }
static inline int dev_count() { return 1; }
static inline int dev_set(int) { return 0; }
static inline int dev_get() { return 0; }
static inline int dev_arch_ok(int, char* name, size_t cap, size_t* lds_limit) {
  snprintf(name, cap, "emu");
  *lds_limit = 160 * 1024;
  return 1;
}
template <class K>
static inline int dev_allow_lds(K, size_t) { return 0; }
static inline int dev_malloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? 0 : -1; }
static inline void dev_free(void* p) { free(p); }
static inline int dev_h2d(void* d, const void* h, size_t n, dev_stream_t) { memcpy(d, h, n); return 0; }
static inline int dev_d2h(void* h, const void* d, size_t n, dev_stream_t) { memcpy(h, d, n); return 0; }
static inline int dev_d2d(void* d, const void* s, size_t n, dev_stream_t) { memmove(d, s, n); return 0; }
static inline int dev_copy2d(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, dev_stream_t) {
  for (size_t r = 0; r < height; ++r) memmove((char*)d + r * dpitch, (const char*)s + r * spitch, width);
  return 0;
}
static inline int dev_memset(void* d, int v, size_t n, dev_stream_t) { memset(d, v, n); return 0; }
static inline int dev_memset32(void* d, uint32_t v, size_t nwords, dev_stream_t) {
  for (size_t i = 0; i < nwords; ++i) ((uint32_t*)d)[i] = v;
  return 0;
}
static inline int dev_sync(dev_stream_t) { return 0; }
static inline int dev_stream_create(dev_stream_t* s) { *s = (void*)1; return 0; }
static inline void dev_stream_destroy(dev_stream_t) {}
// (the emulator runs everything in call order: lanes and events only exercise the engine's bookkeeping)
static inline int dev_stream_create_masked(dev_stream_t* s, const uint32_t*, uint32_t) { *s = (void*)2; return 0; }
static inline int dev_cu_count(int) { return 256; }
static inline int dev_event_create_sync(int* e) { *e = 0; return 0; }
static inline int dev_stream_wait(dev_stream_t, int) { return 0; }
static inline int dev_event_sync(int) { return 0; }
static inline int dev_check_launch() { return 0; }
static inline int dev_host_alloc(void** p, size_t n) { *p = malloc(n); return *p ? 0 : -1; }
static inline void dev_host_free(void* p) { free(p); }
static inline int dev_event_create(dev_event_t* e) { *e = 0; return 0; }
static inline void dev_event_destroy(dev_event_t) {}
static inline void dev_event_record(dev_event_t, dev_stream_t) {}
static inline float dev_event_ms(dev_event_t, dev_event_t) { return 0.f; }
#endif
