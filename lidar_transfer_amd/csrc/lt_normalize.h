// lt_normalize.h -- the 1/sqrt seed of the reference's normalize() (Vector3.h:73-89) on the device.
//
// The reference seeds its Newton-Raphson step with _mm_rsqrt_ps (Vector3.h:83), whose result bits are CPU-vendor
// specific.  RSQRTSS is exponent-invariant and looks only at the exponent parity and the leading mantissa bits of
// its input: 10 on Intel, 12 on AMD Zen 5 (tools/probe_rsqrt.c).  oracle/gen_rsqrt_table.c measures the 2 x 2^M
// entry table on a host and PROVES exhaustively (all 2 130 706 432 positive normal floats) that it reproduces the
// instruction.  Two measured tables are shipped:
//   LT_RSQRT_SSE_TABLE  GenuineIntel (the build container's Xeon; the golden vectors were made there) -- default
//   LT_RSQRT_AMD_TABLE  AuthenticAMD (the MI355X box's EPYC 9575F host)                                -- LT_TRACE_NORM_AMD
// Replaying the seed makes the ray directions -- and with them every output bit -- those the reference produces
// on that CPU family; LT_TRACE_NORM_EXACT selects a correctly rounded seed instead (vendor independent).
#pragma once
#include "lt_internal.h"
#define LT_TABLE_ATTR __device__
#include "lt_rsqrt_sse_table.h"
#include "lt_rsqrt_amd_table.h"

template <int BITS>
__device__ __forceinline__ float lt_rsqrt_x86(const unsigned* __restrict__ table, float x) {
  const unsigned b = __float_as_uint(x);
  const int e = (int)((b >> 23) & 255u);
  if (e == 0) return INFINITY;                      // zero / denormal source is treated as zero
  if (e == 255) return (b & 0x7fffffu) ? x : 0.0f;  // NaN -> NaN, +inf -> 0
  const int p = (e - 127) & 1;
  const int k = (e - 127 - p) / 2;
  return __uint_as_float(table[(p << BITS) + ((b >> (23 - BITS)) & ((1u << BITS) - 1u))] - ((unsigned)k << 23));
}

// r0 ~ 1 / sqrt(D) as the selected host would have produced it
__device__ __forceinline__ float lt_rsqrt_seed(float D, unsigned flags) {
  if (flags & LT_TRACE_NORM_EXACT) return 1.0f / sqrtf(D);
  if (flags & LT_TRACE_NORM_AMD) return lt_rsqrt_x86<LT_RSQRT_AMD_TABLE_BITS>(LT_RSQRT_AMD_TABLE, D);
  return lt_rsqrt_x86<LT_RSQRT_SSE_TABLE_BITS>(LT_RSQRT_SSE_TABLE, D);
}
