// FETCH_SIZE calibration for the access patterns of k_sc_tris (MI355X_MICROARCH.md, HBM section: "FETCH_SIZE reports exactly
// 1/2 of the bytes of a wide coalesced streaming read (16 B/lane) ... other access widths are uncalibrated: calibrate on a known
// byte count in your own access pattern").  Every kernel below moves a KNOWN number of bytes out of a 1 GiB table (4 x the
// Infinity Cache, fresh per launch: the launches walk four tables in turn); run under
//     rocprofv3 --kernel-trace --pmc FETCH_SIZE -- /tmp/fetch_calib        (tools/r05/fetch_calib.sh)
// and divide.  Patterns:
//   k_stream16   16 B per lane, coalesced (the guide's own case: expect known / counted = 2.0)
//   k_stream12   one 12-byte index triple per lane, coalesced (global_load_dwordx3: a wave reads 768 contiguous bytes) --
//                k_sc_tris's face loads
//   k_stream4    4 B per lane, coalesced
//   k_window12   one 12-byte vertex per lane at a mesh-like index: face f reads vertex f/2 + (hash(f) % 64) -- every vertex is
//                read ~6 times by lanes of neighbouring waves, the table is swept once: unique bytes = table bytes --
//                k_sc_tris's vertex gathers on a marching-cubes mesh
//   k_random12   one 12-byte record per lane at a uniformly random index: every gather is its own line(s)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_stream16(const float4* __restrict__ t, size_t n, float* sink) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float4 v = t[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 12345.678f) *sink = acc;
}
struct f3 { float x, y, z; };
__global__ __launch_bounds__(256) void k_stream12(const f3* __restrict__ t, size_t n, float* sink) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const f3 v = t[i];
    acc += v.x + v.y + v.z;
  }
  if (acc == 12345.678f) *sink = acc;
}
__global__ __launch_bounds__(256) void k_stream4(const float* __restrict__ t, size_t n, float* sink) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += t[i];
  if (acc == 12345.678f) *sink = acc;
}
__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32; x *= 0x94D049BB133111EBull; x ^= x >> 29;
  return x;
}
__global__ __launch_bounds__(256) void k_window12(const f3* __restrict__ t, size_t n_verts, size_t n_faces, float* sink) {
  float acc = 0.f;
  for (size_t f = (size_t)blockIdx.x * 256 + threadIdx.x; f < n_faces; f += (size_t)gridDim.x * 256) {
    size_t v = f / 2 + (size_t)(mix(f) % 64);
    if (v >= n_verts) v = n_verts - 1;
    const f3 q = t[v];
    acc += q.x + q.y + q.z;
  }
  if (acc == 12345.678f) *sink = acc;
}
__global__ __launch_bounds__(256) void k_random12(const f3* __restrict__ t, size_t n_verts, size_t n_gathers, float* sink) {
  float acc = 0.f;
  for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < n_gathers; g += (size_t)gridDim.x * 256) {
    const f3 q = t[mix(g + 77) % n_verts];
    acc += q.x + q.y + q.z;
  }
  if (acc == 12345.678f) *sink = acc;
}

int main() {
  const size_t bytes = (size_t)1 << 30;
  const int n_tab = 4, reps = 8, blocks = 256 * 16;
  char* tab[n_tab];
  float* sink;
  CK(hipMalloc(&sink, 4));
  for (int k = 0; k < n_tab; ++k) { CK(hipMalloc(&tab[k], bytes)); CK(hipMemset(tab[k], k + 1, bytes)); }
  CK(hipDeviceSynchronize());
  const size_t n16 = bytes / 16, n12 = bytes / 12, n4 = bytes / 4;
  const size_t n_faces = 2 * (n12 - 64), n_rand = (size_t)16 << 20;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms[5] = {0, 0, 0, 0, 0};
  for (int r = 0; r < reps; ++r) {
    const char* t = tab[r % n_tab];
    float m;
#define TIMED(idx, call) CK(hipEventRecord(e0)); call; CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&m, e0, e1)); if (r) ms[idx] += m;
    TIMED(0, hipLaunchKernelGGL(k_stream16, dim3(blocks), dim3(256), 0, 0, (const float4*)t, n16, sink));
    TIMED(1, hipLaunchKernelGGL(k_stream12, dim3(blocks), dim3(256), 0, 0, (const f3*)t, n12, sink));
    TIMED(2, hipLaunchKernelGGL(k_stream4, dim3(blocks), dim3(256), 0, 0, (const float*)t, n4, sink));
    TIMED(3, hipLaunchKernelGGL(k_window12, dim3(blocks), dim3(256), 0, 0, (const f3*)t, n12, n_faces, sink));
    TIMED(4, hipLaunchKernelGGL(k_random12, dim3(blocks), dim3(256), 0, 0, (const f3*)t, n12, n_rand, sink));
  }
  const char* names[5] = {"k_stream16", "k_stream12", "k_stream4", "k_window12", "k_random12"};
  // known bytes per launch: the table once for the four sweeps; for k_random12 every gather is (at least) one line of its own:
  // 12-byte records at a 12-byte pitch straddle a 64-byte boundary in 8 of 64 positions (a 128-byte boundary in 8 of 128)
  const double known[5] = {(double)n16 * 16, (double)n12 * 12, (double)n4 * 4, (double)n12 * 12, 0.0};
  printf("# reps %d (first untimed), table %zu bytes x %d, %d workgroups\n", reps, bytes, n_tab, blocks);
  for (int k = 0; k < 5; ++k) {
    const double avg = ms[k] / (reps - 1);
    if (k < 4) printf("%-12s known_bytes %.0f  avg_ms %.4f  GB/s %.1f\n", names[k], known[k], avg, known[k] / (avg * 1e-3) / 1e9);
    else printf("%-12s gathers %zu  avg_ms %.4f  G gathers/s %.2f  (64-B lines: %.0f bytes, 128-B lines: %.0f bytes incl. straddles)\n", names[k],
                n_rand, avg, n_rand / (avg * 1e-3) / 1e9, n_rand * 64.0 * (1 + 8.0 / 64), n_rand * 128.0 * (1 + 8.0 / 128));
  }
  return 0;
}
