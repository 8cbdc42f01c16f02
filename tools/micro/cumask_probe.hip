// Which compute units does a CU-masked stream use?  Bit i of the mask handed to hipExtStreamCreateWithCUMask is dealt
// round-robin over the XCDs by the driver (bit i -> XCD i % 8); this probe launches more workgroups than fit at once on a
// stream masked to the first F bits and on one masked to the rest, and prints the (XCC, SE, CU) sets each one ran on.
//   hipcc --offload-arch=gfx950 -O2 cumask_probe.hip -o cumask_probe && ./cumask_probe 160
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <set>
#include <vector>

__global__ void probe(unsigned* out, int spin) {
  if (threadIdx.x == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));    // HW_REG_XCC_ID[3:0]
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));     // HW_REG_HW_ID
    out[blockIdx.x] = (xcc << 24) | (hw & 0xFFFFFFu);
  }
  // stay resident long enough that every CU of the lane gets a workgroup
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) {}
}

int main(int argc, char** argv) {
  const int nf = argc > 1 ? atoi(argv[1]) : 160;
  int ncu = 0;
  hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  const int words = (ncu + 31) / 32;
  std::vector<unsigned> mf(words, 0), mb(words, 0);
  for (int i = 0; i < ncu; ++i) (i < nf ? mf : mb)[i >> 5] |= 1u << (i & 31);
  hipStream_t sf, sb;
  if (hipExtStreamCreateWithCUMask(&sf, words, mf.data()) != hipSuccess || hipExtStreamCreateWithCUMask(&sb, words, mb.data()) != hipSuccess) {
    printf("hipExtStreamCreateWithCUMask failed: %s\n", hipGetErrorString(hipGetLastError()));
    return 1;
  }
  const int nwg = 4096;
  unsigned *df, *db;
  hipMalloc(&df, nwg * 4);
  hipMalloc(&db, nwg * 4);
  hipEvent_t e0, e1, e2, e3;
  hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2); hipEventCreate(&e3);
  hipEventRecord(e0, sf);
  hipLaunchKernelGGL(probe, dim3(nwg), dim3(1024), 65536, sf, df, 20000);
  hipEventRecord(e1, sf);
  hipEventRecord(e2, sb);
  hipLaunchKernelGGL(probe, dim3(nwg), dim3(1024), 65536, sb, db, 20000);
  hipEventRecord(e3, sb);
  hipDeviceSynchronize();
  float tf = 0, tb = 0;
  hipEventElapsedTime(&tf, e0, e1);
  hipEventElapsedTime(&tb, e2, e3);
  std::vector<unsigned> hf(nwg), hb(nwg);
  hipMemcpy(hf.data(), df, nwg * 4, hipMemcpyDeviceToHost);
  hipMemcpy(hb.data(), db, nwg * 4, hipMemcpyDeviceToHost);
  auto report = [&](const char* name, const std::vector<unsigned>& h, float ms) {
    std::set<unsigned> cus;
    int per_xcc[16] = {0};
    for (unsigned v : h) {
      const unsigned xcc = v >> 24, cu = (v >> 8) & 15, sh = (v >> 12) & 1, se = (v >> 13) & 7;
      if (cus.insert((xcc << 16) | (se << 8) | (sh << 4) | cu).second) per_xcc[xcc & 15]++;
    }
    printf("%s: %zu distinct CUs, per XCC:", name, cus.size());
    for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
    printf("   (%.3f ms)\n", ms);
    // XCC of consecutive workgroups: round-robin?
    printf("  XCC of workgroups 0..15:");
    for (int i = 0; i < 16; ++i) printf(" %u", h[i] >> 24);
    printf("\n");
    return cus;
  };
  auto a = report("front lane", hf, tf);
  auto b = report("back lane ", hb, tb);
  int common = 0;
  for (unsigned c : a) common += b.count(c);
  printf("CUs used by both lanes: %d\n", common);
  return 0;
}
