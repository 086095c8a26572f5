// tools/valu_calib.hip -- how many shader cycles does ONE wave64 VALU instruction cost on this chip, and in which unit do
// SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES count?  ("VALU-issue bound" was argued
// with 4 cycles per instruction in one place and 2 in another.)
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/valu_calib.bin tools/valu_calib.hip
//   ./tools/valu_calib.bin                       -> cycles per instruction from s_memtime, per regime
//   rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES ... -- ./tools/valu_calib.bin
//                                                -> the same launches seen by the counters (tools/valu_calib.sh)
//
// Every kernel executes exactly N_INST v_fma_f32 per lane in straight-line inline assembly (the compiler cannot fuse, pair
// or reorder them), between two s_memtime reads (shader-clock ticks, MI355X_MICROARCH.md) and two wall-clock reads
// (100 MHz constant clock -> the effective shader clock).  Regimes:
//   dep1   one accumulator, every fma depends on the previous one     (latency of the VALU pipeline)
//   ind8   eight independent accumulators, round-robin                (issue rate of ONE wave)
//   each at 1, 2, 4 and 8 waves per SIMD (256 CUs x 4 SIMDs x w waves, launched as 256-thread workgroups)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define N_REP 64       // loop trips
#define N_UNROLL 64    // fmas per trip
#define N_INST (N_REP * N_UNROLL)

#define FMA_DEP "v_fma_f32 %0, %0, %8, %9\n\t"
#define FMA8                                                                                                  \
  "v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t" \
  "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9\n\t"
#define R8(x) x x x x x x x x

template <bool DEP>
__global__ __launch_bounds__(256) void k_valu(float* __restrict__ sink, unsigned long long* __restrict__ stamps, float m,
                                              float c) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const unsigned long long w0 = wall_clock64();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < N_REP; ++r) {
    if (DEP) {
      asm volatile(R8(R8(FMA_DEP))
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                   : "v"(m), "v"(c));
    } else {
      asm volatile(R8(FMA8)
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                   : "v"(m), "v"(c));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  const unsigned long long w1 = wall_clock64();
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
  if ((threadIdx.x & 63) == 0) {
    stamps[2 * wave] = t1 - t0;
    stamps[2 * wave + 1] = w1 - w0;
  }
  sink[blockIdx.x * 256 + threadIdx.x] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
}

int main() {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
  const int cus = prop.multiProcessorCount;
  printf("# %s, %d CUs, clockRate %.0f MHz; %d v_fma_f32 per lane per launch\n", prop.gcnArchName, cus, prop.clockRate / 1e3,
         N_INST);
  printf("# regime waves/SIMD  cycles/inst(s_memtime,median wave)  wall_us(median wave)  eff_GHz  launch_us(events)  "
         "wave-insts  => SIMD-cycles per wave-instruction at that occupancy\n");
  float* sink;
  unsigned long long* stamps;
  const int max_waves = cus * 4 * 8;
  hipMalloc(&sink, (size_t)max_waves * 64 * sizeof(float));
  hipMalloc(&stamps, (size_t)max_waves * 2 * sizeof(unsigned long long));
  std::vector<unsigned long long> h(max_waves * 2);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int dep = 1; dep >= 0; --dep)
    for (int wps : {1, 2, 4, 8}) {
      const int blocks = cus * wps;  // a 256-thread workgroup = one wave on each of a CU's four SIMDs
      float ms = 0;
      for (int rep = 0; rep < 3; ++rep) {  // the last repetition counts (clocks ramped up)
        hipEventRecord(e0);
        if (dep) hipLaunchKernelGGL(k_valu<true>, dim3(blocks), dim3(256), 0, 0, sink, stamps, 1.0000001f, 1e-9f);
        else hipLaunchKernelGGL(k_valu<false>, dim3(blocks), dim3(256), 0, 0, sink, stamps, 1.0000001f, 1e-9f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      const int waves = blocks * 4;
      hipMemcpy(h.data(), stamps, (size_t)waves * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      std::vector<double> cyc(waves), wall(waves);
      for (int w = 0; w < waves; ++w) { cyc[w] = (double)h[2 * w]; wall[w] = (double)h[2 * w + 1]; }
      std::sort(cyc.begin(), cyc.end());
      std::sort(wall.begin(), wall.end());
      const double c_med = cyc[waves / 2], w_med_us = wall[waves / 2] / 100.0;  // wall_clock64: 100 MHz
      const double ghz = c_med / (w_med_us * 1e3);
      // wps waves share a SIMD: the SIMD issued wps * N_INST wave-instructions in c_med cycles
      printf("%s %d  %.3f  %.2f  %.3f  %.1f  %lld  => %.3f\n", dep ? "dep1" : "ind8", wps, c_med / N_INST, w_med_us, ghz,
             ms * 1e3, (long long)waves * N_INST, c_med / ((double)N_INST * wps));
    }
  hipFree(sink);
  hipFree(stamps);
  return 0;
}
