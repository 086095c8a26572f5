// How many single-wave workgroups a CU really holds as a function of their LDS size (and how many 256-thread ones): every
// wave stamps its start, spins for ~40 us of wall clock and stamps its end; the host sweeps the stamps for the largest
// number of waves alive at the same time.   hipcc --offload-arch=gfx950 -O3 -o /tmp/occ_probe tools/occ_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void k_spin(uint64_t* stamps, int spin_ticks) {
  extern __shared__ int lds[];
  const uint64_t t0 = wall_clock64();
  if (threadIdx.x == 0) lds[0] = (int)t0;  // (the allocation is used)
  while (wall_clock64() - t0 < (uint64_t)spin_ticks) __builtin_amdgcn_s_sleep(8);
  if ((threadIdx.x & 63) == 0) {
    const size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    stamps[2 * w] = t0;
    stamps[2 * w + 1] = wall_clock64();
  }
}

static int max_alive(const std::vector<uint64_t>& h, size_t n) {
  std::vector<std::pair<uint64_t, int>> ev;
  for (size_t i = 0; i < n; ++i) { ev.push_back({h[2 * i], 1}); ev.push_back({h[2 * i + 1], -1}); }
  std::sort(ev.begin(), ev.end());
  int cur = 0, best = 0;
  for (auto& e : ev) { cur += e.second; best = std::max(best, cur); }
  return best;
}

int main(int argc, char**) {
  const int n_wg = 16384;
  uint64_t* d;
  CK(hipMalloc(&d, (size_t)n_wg * 4 * 2 * sizeof(uint64_t)));
  printf("%10s %8s %16s %14s %12s\n", "threads/WG", "LDS B", "waves alive max", "per CU (/256)", "ms");
  // (argv: "fine" = the sizes around the emission's workgroups, to find the allocation granule)
  const bool fine = argc > 1;
  const std::vector<int> coarse = {512, 2048, 4096, 6144, 8200, 12288, 16384, 32768, 40000};
  const std::vector<int> fine64 = {6400, 6656, 6724, 6912, 7168, 7424, 7492, 7680, 7752, 8192};
  const std::vector<int> fine256 = {25600, 26112, 26624, 26880, 26896, 27136, 27306, 27648, 28160, 29968, 30720, 31744, 32016, 32768};
  for (int threads : {64, 256}) {
    for (int lds : (fine ? (threads == 64 ? fine64 : fine256) : coarse)) {
      CK(hipFuncSetAttribute((const void*)k_spin, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
      const size_t nw = (size_t)n_wg * (threads / 64);
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_spin, dim3(n_wg), dim3(threads), lds, 0, d, 2000);  // 20 us
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<uint64_t> h(2 * nw);
      CK(hipMemcpy(h.data(), d, 2 * nw * sizeof(uint64_t), hipMemcpyDeviceToHost));
      const int m = max_alive(h, nw);
      printf("%10d %8d %16d %14.1f %12.3f\n", threads, lds, m, m / 256.0, ms);
    }
  }
  return 0;
}
