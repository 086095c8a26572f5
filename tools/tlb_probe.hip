// What a dependent load costs on this chip as a function of the ADDRESS RANGE it wanders over: a single lane chases a random
// permutation of N nodes spaced `stride` bytes apart (every hop a cold cache line: the nodes' lines exceed L2 + Infinity
// Cache), so the only thing that changes between the cases is how many 2 MB pages the chain visits -- the reach of the
// address translation caches.  The fusion chain's kernels wander over 4 x 3.2 GB volumes.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/tlb_probe tools/tlb_probe.hip && /tmp/tlb_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void k_init(uint64_t* buf, size_t n, size_t stride_q, uint64_t a, uint64_t c) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) buf[i * stride_q] = (i * a + c) & (n - 1);  // full-period LCG on a power of two: one cycle through all nodes
}

__global__ void k_chase(const uint64_t* __restrict__ buf, size_t stride_q, uint64_t start, int hops, uint64_t* out) {
  uint64_t idx = start;
  const uint64_t t0 = wall_clock64();
  for (int h = 0; h < hops; ++h) idx = __builtin_nontemporal_load(buf + idx * stride_q);
  const uint64_t t1 = wall_clock64();
  out[0] = t1 - t0;  // 100 MHz
  out[1] = idx;
}

// many independent chains at once: `waves` waves of 64 lanes, a chain per lane
__global__ void k_chase_many(const uint64_t* __restrict__ buf, size_t stride_q, size_t n, int hops, uint64_t* out) {
  uint64_t idx = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2654435761ull & (n - 1);
  for (int h = 0; h < hops; ++h) idx = buf[idx * stride_q];
  if (idx == 0xdeadbeefull) out[0] = idx;
}

// ... with M independent chains per lane (memory-level parallelism per wave), and optionally BOTH 64-byte halves of a node's
// 128-byte line per hop: what the memory side delivers in random REQUESTS per second
template <int M, bool PAIR>
__global__ void k_chase_mlp(const uint64_t* __restrict__ buf, size_t stride_q, size_t n, int hops, uint64_t* out) {
  uint64_t idx[M];
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int m = 0; m < M; ++m) idx[m] = ((t * M + m) * 2654435761ull + 12345u) & (n - 1);
  uint64_t acc = 0;
  for (int h = 0; h < hops; ++h) {
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const uint64_t* p = buf + idx[m] * stride_q;
      if (PAIR) acc += p[8];  // the other 64-byte half of the 128-byte line (stride >= 128 B)
      idx[m] = p[0];
    }
  }
  if (acc == 0x1234567ull) out[0] = acc;
#pragma unroll
  for (int m = 0; m < M; ++m)
    if (idx[m] == 0xdeadbeefull) out[0] = idx[m];
}

template <int M, bool PAIR>
static double rate_mlp(const uint64_t* buf, size_t st, size_t n, uint64_t* out) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int hops = 64 / M;
  hipLaunchKernelGGL((k_chase_mlp<M, PAIR>), dim3(8192), dim3(64), 0, 0, buf, st / 8, n, 4, out);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_chase_mlp<M, PAIR>), dim3(8192), dim3(64), 0, 0, buf, st / 8, n, hops, out);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return 8192.0 * 64 * M * hops / (ms * 1e-3) / 1e9;  // G hops / s
}

int main() {
  const size_t n = (size_t)1 << 24;  // 16 M nodes: 1 GB of distinct cache lines, far beyond L2 (32 MB) + Infinity Cache (256 MB)
  const size_t strides[] = {64, 256, 1024, 2048, 3072};
  uint64_t* out;
  CK(hipMalloc(&out, 16));
  printf("%10s %10s %12s %14s %16s\n", "stride B", "range GB", "2MB pages", "ns per hop", "many: Ghops/s");
  for (size_t st : strides) {
    const size_t bytes = n * st;
    uint64_t* buf;
    if (hipMalloc(&buf, bytes) != hipSuccess) { printf("%10zu: allocation of %.1f GB failed\n", st, bytes / 1e9); continue; }
    hipLaunchKernelGGL(k_init, dim3((unsigned)(n / 256)), dim3(256), 0, 0, buf, n, st / 8, 1664525ull * 4 + 1, 1013904223ull | 1);
    CK(hipDeviceSynchronize());
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, 0, buf, st / 8, (uint64_t)(12345 + 7919 * rep), 20000, out);
      uint64_t h[2];
      CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
      const double ns = h[0] * 10.0 / 20000;
      if (ns < best) best = ns;
    }
    // throughput with the chip full of independent chains (8192 waves x 64 lanes, 64 hops each)
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_chase_many, dim3(8192), dim3(64), 0, 0, buf, st / 8, n, 8, out);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_chase_many, dim3(8192), dim3(64), 0, 0, buf, st / 8, n, 64, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%10zu %10.2f %12zu %14.0f %16.2f\n", st, bytes / 1e9, bytes >> 21, best, 8192.0 * 64 * 64 / (ms * 1e-3) / 1e9);
    if (st == 256 || st == 2048) {
      printf("    chains per lane 1 / 2 / 4 / 8, 64 B per hop:  %.1f / %.1f / %.1f / %.1f G hops/s\n", rate_mlp<1, false>(buf, st, n, out),
             rate_mlp<2, false>(buf, st, n, out), rate_mlp<4, false>(buf, st, n, out), rate_mlp<8, false>(buf, st, n, out));
      printf("    ... both halves of the 128-byte line per hop:  %.1f / %.1f / %.1f / %.1f G hops/s (x 128 B)\n", rate_mlp<1, true>(buf, st, n, out),
             rate_mlp<2, true>(buf, st, n, out), rate_mlp<4, true>(buf, st, n, out), rate_mlp<8, true>(buf, st, n, out));
    }
    CK(hipFree(buf));
  }
  return 0;
}
