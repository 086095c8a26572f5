// tools/sort_probe.hip -- MEASUREMENT ONLY (not part of liblidarhip.so): what the ROCm library's device radix sort
// (rocPRIM, onesweep) needs for the LBVH build's sort -- 1 M (30-bit Morton key, face index) pairs -- on this GPU, as the
// yardstick for lt_build.hip's own three-pass LSD sort (k_hist / k_scan / k_scatter, 9 launches, ms_sort in the bench line).
//   hipcc -O3 --offload-arch=gfx950 -o tools/sort_probe.bin tools/sort_probe.hip && tools/sort_probe.bin [n]
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 999788;
  std::vector<uint32_t> hk(n), hv(n);
  uint32_t s = 12345u;
  for (int i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; hk[i] = (s >> 2) & 0x3FFFFFFFu; hv[i] = (uint32_t)i; }
  uint32_t *k0, *k1, *v0, *v1;
  CK(hipMalloc(&k0, 4 * n)); CK(hipMalloc(&k1, 4 * n)); CK(hipMalloc(&v0, 4 * n)); CK(hipMalloc(&v1, 4 * n));
  size_t tmp_bytes = 0;
  CK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, k0, k1, v0, v1, (size_t)n, 0, 30));
  void* tmp; CK(hipMalloc(&tmp, tmp_bytes));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9f, sum = 0.f;
  const int reps = 30;
  for (int r = 0; r < reps + 3; ++r) {
    CK(hipMemcpyAsync(k0, hk.data(), 4 * n, hipMemcpyHostToDevice, st));
    CK(hipMemcpyAsync(v0, hv.data(), 4 * n, hipMemcpyHostToDevice, st));
    CK(hipEventRecord(a, st));
    CK(rocprim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, v0, v1, (size_t)n, 0, 30, st));
    CK(hipEventRecord(b, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (r >= 3) { best = std::min(best, ms); sum += ms; }
  }
  std::vector<uint32_t> ok(n);
  CK(hipMemcpy(ok.data(), k1, 4 * n, hipMemcpyDeviceToHost));
  bool sorted = std::is_sorted(ok.begin(), ok.end());
  printf("rocprim::radix_sort_pairs n=%d bits 0..30: avg %.1f us, best %.1f us, temp %zu B, sorted=%d\n", n, 1e3f * sum / reps, 1e3f * best, tmp_bytes, (int)sorted);
  return sorted ? 0 : 2;
}
