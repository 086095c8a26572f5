// What the chip does with the z-min merge of the scatter strategy: non-returning 64-bit atomicMin on random cells of an
// image-sized region, issued from every CU at once (the cells of a scan are 1 MB; eight scans in a batch: 8 MB).  The eight
// XCDs have private L2s, so a device-scope atomic is executed at the memory side (TCC_EA0_ATOMIC counts every one of
// k_sc_tris's) -- this measures how many per second that is.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/atomic_probe tools/atomic_probe.hip && /tmp/atomic_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE>  // 0: 64-bit atomicMin, 1: 32-bit atomicMin, 2: plain 64-bit store (what the same addresses cost without the RMW)
__global__ void k_atomics(unsigned long long* cells, size_t n_cells, int per_lane, unsigned seed) {
  uint64_t x = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + seed;
  for (int i = 0; i < per_lane; ++i) {
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    const size_t c = (size_t)(x % n_cells);
    const unsigned long long v = (x >> 8) | 1ull;
    if (MODE == 0) atomicMin(cells + c, v);
    else if (MODE == 1) atomicMin((unsigned*)cells + 2 * c, (unsigned)v);
    else cells[c] = v;
  }
}

template <int MODE>
static double rate(unsigned long long* cells, size_t n_cells, int waves) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int per_lane = 64;
  hipLaunchKernelGGL(k_atomics<MODE>, dim3(waves), dim3(64), 0, 0, cells, n_cells, 4, 1u);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_atomics<MODE>, dim3(waves), dim3(64), 0, 0, cells, n_cells, per_lane, 2u);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return (double)waves * 64 * per_lane / (ms * 1e-3) / 1e9;
}

// ---- the pattern of the z-min PROJECTION (k_pb_project): the points of a spinning LiDAR arrive in firing order, so the lanes
// of a wave hit neighbouring cells -- 8-byte keys, `run` consecutive lanes inside one image row's run of consecutive cells
// (run = 64: a whole wave in 512 contiguous bytes = 8 lines; run = 8: one 64-byte line per 8 lanes; run = 1: the random pattern
// above), runs placed at random in the region.  Rate in LANE atomics per second, the unit bench.py's projection record uses.
__global__ void k_atomics_runs(unsigned long long* cells, size_t n_cells, int per_lane, unsigned seed, int run) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t x = (tid / run) * 0x9E3779B97F4A7C15ull + seed;
  for (int i = 0; i < per_lane; ++i) {
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    const size_t c = (size_t)((x % (n_cells / run)) * run + tid % run);
    atomicMin(cells + c, ((x >> 8) | 1ull) + tid);
  }
}
static double rate_runs(unsigned long long* cells, size_t n_cells, int waves, int run) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int per_lane = 64;
  hipLaunchKernelGGL(k_atomics_runs, dim3(waves), dim3(64), 0, 0, cells, n_cells, 4, 1u, run);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_atomics_runs, dim3(waves), dim3(64), 0, 0, cells, n_cells, per_lane, 2u, run);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return (double)waves * 64 * per_lane / (ms * 1e-3) / 1e9;
}

int main() {
  {
    printf("# lane atomics/s (64-bit atomicMin) by run length of neighbouring lanes -> neighbouring cells\n");
    printf("%12s %8s %12s %12s %12s %12s %12s\n", "region", "waves", "run 1", "run 4", "run 8", "run 16", "run 64");
    for (size_t bytes : {(size_t)1 << 20, (size_t)8 << 20, (size_t)64 << 20}) {
      unsigned long long* cells;
      CK(hipMalloc(&cells, bytes));
      CK(hipMemset(cells, 0xFF, bytes));
      for (int waves : {8192, 32768})
        printf("%9zu MB %8d %12.2f %12.2f %12.2f %12.2f %12.2f\n", bytes >> 20, waves, rate_runs(cells, bytes / 8, waves, 1),
               rate_runs(cells, bytes / 8, waves, 4), rate_runs(cells, bytes / 8, waves, 8), rate_runs(cells, bytes / 8, waves, 16),
               rate_runs(cells, bytes / 8, waves, 64));
      CK(hipFree(cells));
    }
  }

  printf("%12s %8s %16s %16s %16s\n", "region", "waves", "min64 G/s", "min32 G/s", "store64 G/s");
  const size_t sizes[] = {(size_t)1 << 20, (size_t)8 << 20, (size_t)64 << 20, (size_t)1 << 30};
  for (size_t bytes : sizes) {
    unsigned long long* cells;
    CK(hipMalloc(&cells, bytes));
    CK(hipMemset(cells, 0xFF, bytes));
    for (int waves : {2048, 8192, 32768}) {
      const size_t n = bytes / 8;
      printf("%9zu MB %8d %16.2f %16.2f %16.2f\n", bytes >> 20, waves, rate<0>(cells, n, waves), rate<1>(cells, n, waves), rate<2>(cells, n, waves));
    }
    CK(hipFree(cells));
  }
  return 0;
}
