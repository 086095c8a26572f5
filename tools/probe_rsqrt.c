/* probe_rsqrt.c -- how many leading mantissa bits does RSQRTSS depend on on THIS CPU, and is it
 * exponent-invariant (result for 2^(2k+p) * 1.m == result for 2^p * 1.m with k subtracted from the exponent)?
 *   gcc -O2 -fopenmp -o probe_rsqrt probe_rsqrt.c && ./probe_rsqrt */
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <xmmintrin.h>
static inline float hw(float x) { return _mm_cvtss_f32(_mm_rsqrt_ss(_mm_set_ss(x))); }
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
int main(void) {
  /* 1. exponent invariance */
  long long bad_e = 0;
#pragma omp parallel for reduction(+ : bad_e)
  for (long long b = 0x00800000LL; b < 0x7f800000LL; ++b) {
    const uint32_t u = (uint32_t)b;
    const int e = (int)(u >> 23), p = (e - 127) & 1, k = (e - 127 - p) / 2;
    const uint32_t base = ((127u + p) << 23) | (u & 0x7fffffu);
    if (f2u(hw(u2f(u))) != f2u(hw(u2f(base))) - ((uint32_t)k << 23)) ++bad_e;
  }
  printf("exponent-invariance mismatches: %lld\n", bad_e);
  /* 2. mantissa bits */
  for (int m = 8; m <= 23; ++m) {
    long long bad = 0;
    const int sh = 23 - m;
    for (int p = 0; p < 2; ++p) {
#pragma omp parallel for reduction(+ : bad)
      for (long long i = 0; i < (1LL << 23); ++i) {
        const uint32_t u = ((127u + p) << 23) | (uint32_t)i;
        const uint32_t rep = ((127u + p) << 23) | (((uint32_t)i >> sh) << sh);
        if (f2u(hw(u2f(u))) != f2u(hw(u2f(rep)))) ++bad;
      }
    }
    printf("top %2d mantissa bits: %lld of 16777216 inputs differ from their bucket's first\n", m, bad);
  }
  /* 3. output granularity: how many low result bits are always zero */
  uint32_t orv = 0;
  for (int p = 0; p < 2; ++p)
    for (uint32_t i = 0; i < (1u << 23); i += 1) orv |= f2u(hw(u2f(((127u + p) << 23) | i)));
  printf("OR of result mantissas: 0x%08x\n", orv & 0x7fffffu);
  /* samples */
  for (uint32_t i = 0; i < 8; ++i) printf("%08x -> %08x\n", (127u << 23) | (i << 9), f2u(hw(u2f((127u << 23) | (i << 9)))));
  return 0;
}
