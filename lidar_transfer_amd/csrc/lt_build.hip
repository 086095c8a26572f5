// lt_build.hip -- linear BVH construction on gfx950 (replaces BVH::build, reference
// auxiliary/raytracer/BVH.cpp:143-243, and the triangle de-indexing loop RayTracer.cpp:32-51).
//
// Pipeline (every kernel is free of inter-workgroup communication, so the build is deterministic
// and needs no agent-scope fences):
//   k_bounds     vertex AABB partials per workgroup                      (HBM stream of verts)
//   k_morton     reduce partials, centroid -> 30-bit Morton key, val = face id
//   3 x { k_hist, k_scan, k_scatter }   stable LSD radix sort, 10-bit digits, LDS ranking
//   k_gather     sorted triangle records (v0,e1,e2,face) + padded leaf boxes
//   k_seg_sub / k_seg_top   min/max segment tree over the leaf boxes (coalesced AABB reduction)
//   k_hierarchy  Karras radix-tree topology by binary search on the sorted keys; both child
//                boxes of every node come from O(log) segment-tree range queries
#include "lt_internal.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define WAVE 64

// ---- small device helpers -----------------------------------------------------------------------
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, WAVE));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, WAVE));
  return v;
}

// block-wide reduction of 3 mins + 3 maxes; result valid in every thread
__device__ __forceinline__ void block_bounds(float lo[3], float hi[3], float (*red)[6]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    lo[k] = wave_min(lo[k]);
    hi[k] = wave_max(hi[k]);
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      red[wave][k] = lo[k];
      red[wave][3 + k] = hi[k];
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float l = red[0][k], h = red[0][3 + k];
    for (int w = 1; w < nw; ++w) {
      l = fminf(l, red[w][k]);
      h = fmaxf(h, red[w][3 + k]);
    }
    lo[k] = l;
    hi[k] = h;
  }
}

// ---- scene bounds ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bounds(const float* __restrict__ verts, int n_verts,
                                                float* __restrict__ partial) {
  __shared__ float red[4][6];
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int v = blockIdx.x * 256 + threadIdx.x; v < n_verts; v += gridDim.x * 256) {
    const float x = verts[3 * (size_t)v], y = verts[3 * (size_t)v + 1], z = verts[3 * (size_t)v + 2];
    lo[0] = fminf(lo[0], x); lo[1] = fminf(lo[1], y); lo[2] = fminf(lo[2], z);
    hi[0] = fmaxf(hi[0], x); hi[1] = fmaxf(hi[1], y); hi[2] = fmaxf(hi[2], z);
  }
  block_bounds(lo, hi, red);
  if (threadIdx.x < 3) partial[blockIdx.x * 6 + threadIdx.x] = lo[threadIdx.x];
  else if (threadIdx.x < 6) partial[blockIdx.x * 6 + threadIdx.x] = hi[threadIdx.x - 3];
}

__device__ __forceinline__ uint32_t expand10(uint32_t v) {
  v &= 0x3ffu;
  v = (v | (v << 16)) & 0x030000FFu;
  v = (v | (v << 8)) & 0x0300F00Fu;
  v = (v | (v << 4)) & 0x030C30C3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}

// ---- Morton keys ----------------------------------------------------------------------------------
// Every workgroup re-reduces the LT_BOUNDS_BLOCKS partials (6 KB from L2) instead of waiting for a
// grid-wide result: no atomics, no second launch.  params = {lo.x, lo.y, lo.z, scale, pad}.
__global__ __launch_bounds__(256) void k_morton(const float* __restrict__ verts, const int* __restrict__ faces,
                                                int n_verts, int n_faces, const float* __restrict__ partial,
                                                float* __restrict__ params, uint32_t* __restrict__ keys,
                                                uint32_t* __restrict__ vals, unsigned* __restrict__ flags) {
  __shared__ float red[4][6];
  float lo[3], hi[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    lo[k] = partial[threadIdx.x * 6 + k];
    hi[k] = partial[threadIdx.x * 6 + 3 + k];
  }
  block_bounds(lo, hi, red);
  const float ext = fmaxf(fmaxf(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
  const float scale = ext > 0.0f ? 1024.0f / ext : 0.0f;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const float maxabs = fmaxf(fmaxf(fmaxf(fabsf(lo[0]), fabsf(hi[0])), fmaxf(fabsf(lo[1]), fabsf(hi[1]))),
                               fmaxf(fabsf(lo[2]), fabsf(hi[2])));
    params[0] = lo[0]; params[1] = lo[1]; params[2] = lo[2];
    params[3] = scale;
    params[4] = 1e-4f + 2e-6f * maxabs;  // box padding, see DESIGN.md "conservative boxes"
  }
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_faces) return;
  const int a = faces[3 * (size_t)i], b = faces[3 * (size_t)i + 1], c = faces[3 * (size_t)i + 2];
  uint32_t key = 0x3FFFFFFFu;
  if ((unsigned)a < (unsigned)n_verts && (unsigned)b < (unsigned)n_verts && (unsigned)c < (unsigned)n_verts) {
    const float* pa = verts + 3 * (size_t)a;
    const float* pb = verts + 3 * (size_t)b;
    const float* pc = verts + 3 * (size_t)c;
    // centroid as the reference: (v0 + v1 + v2) / 3 (Triangle.h:78-80)
    const float cx = ((pa[0] + pb[0]) + pc[0]) / 3.0f;
    const float cy = ((pa[1] + pb[1]) + pc[1]) / 3.0f;
    const float cz = ((pa[2] + pb[2]) + pc[2]) / 3.0f;
    const uint32_t qx = (uint32_t)fminf(fmaxf((cx - lo[0]) * scale, 0.0f), 1023.0f);
    const uint32_t qy = (uint32_t)fminf(fmaxf((cy - lo[1]) * scale, 0.0f), 1023.0f);
    const uint32_t qz = (uint32_t)fminf(fmaxf((cz - lo[2]) * scale, 0.0f), 1023.0f);
    key = (expand10(qx) << 2) | (expand10(qy) << 1) | expand10(qz);
  } else {
    atomicOr(flags, LT_FLAG_BAD_INDEX);
  }
  keys[i] = key;
  vals[i] = (uint32_t)i;
}

// ---- LSD radix sort: 3 passes of 10-bit digits over the 30-bit Morton key --------------------------------
// One workgroup owns LT_SORT_TILE consecutive keys per pass.  hist is digit-major: hist[d * nb + b],
// followed by the LT_RD digit totals.
#ifndef LT_RB
#define LT_RB 10   // digit width of the LSD radix sort (A/B: -DLT_RB=8 -> four passes of 256 digits)
#endif
#define LT_RD (1 << LT_RB)
#define LT_DPT (LT_RD / LT_SORT_THREADS)  // digits owned per thread (4)

__global__ __launch_bounds__(LT_SORT_THREADS) void k_hist(const uint32_t* __restrict__ keys, int n, int shift,
                                                         uint32_t* __restrict__ hist, int nb) {
  __shared__ uint32_t h[LT_RD];
#pragma unroll
  for (int k = 0; k < LT_DPT; ++k) h[k * LT_SORT_THREADS + threadIdx.x] = 0;
  __syncthreads();
  const int base = blockIdx.x * LT_SORT_TILE;
#pragma unroll
  for (int k = 0; k < LT_SORT_TILE / LT_SORT_THREADS; ++k) {
    const int e = base + k * LT_SORT_THREADS + threadIdx.x;
    if (e < n) atomicAdd(&h[(keys[e] >> shift) & (LT_RD - 1u)], 1u);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < LT_DPT; ++k) {
    const int d = k * LT_SORT_THREADS + threadIdx.x;
    hist[(size_t)d * nb + blockIdx.x] = h[d];
  }
}

// Row scan: workgroup d turns hist[d*nb .. d*nb+nb) (the counts of digit d per tile) into its
// exclusive prefix sum and stores the digit total in hist[LT_RD*nb + d].  The cross-digit prefix
// (LT_RD values) is redone by every k_scatter workgroup in LDS -- cheaper than another launch.
__global__ __launch_bounds__(256) void k_scan(uint32_t* __restrict__ hist, int nb) {
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t carry_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t* row = hist + (size_t)blockIdx.x * nb;
  uint32_t carry = 0;
  for (int base = 0; base < nb; base += 256) {
    const int i = base + threadIdx.x;
    const uint32_t v = i < nb ? row[i] : 0u;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(inc, o, WAVE);
      if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    if (i < nb) row[i] = carry + woff + inc - v;
    if (threadIdx.x == 255) carry_s = carry + woff + inc;
    __syncthreads();
    carry = carry_s;
  }
  if (threadIdx.x == 0) hist[(size_t)LT_RD * nb + blockIdx.x] = carry;
}

// Stable scatter.  Wave w of the workgroup ranks the contiguous quarter [w*1024, (w+1)*1024) of the
// tile in 16 rounds of 64 keys: lanes holding the same digit find each other with LT_RB ballots, the
// running per-(wave, digit) count lives in LDS.  Order inside a tile is (wave, round, lane) = memory
// order, so the sort is stable.
__global__ __launch_bounds__(LT_SORT_THREADS) void k_scatter(const uint32_t* __restrict__ keys_in,
                                                            const uint32_t* __restrict__ vals_in,
                                                            uint32_t* __restrict__ keys_out,
                                                            uint32_t* __restrict__ vals_out, int n, int shift,
                                                            const uint32_t* __restrict__ hist, int nb) {
  constexpr int ROUNDS = LT_SORT_TILE / LT_SORT_THREADS;  // 16
  constexpr int NW = LT_SORT_THREADS / 64;                 // 4
  __shared__ uint32_t cnt[NW][LT_RD];
  __shared__ uint32_t wtot[NW];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int w = 0; w < NW; ++w)
#pragma unroll
    for (int k = 0; k < LT_DPT; ++k) cnt[w][k * LT_SORT_THREADS + threadIdx.x] = 0;
  __syncthreads();
  volatile uint32_t* mycnt = cnt[wave];
  const int base = blockIdx.x * LT_SORT_TILE + wave * (ROUNDS * 64);
  uint32_t key[ROUNDS], val[ROUNDS], rank[ROUNDS];
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int e = base + r * 64 + lane;
    const bool valid = e < n;
    key[r] = valid ? keys_in[e] : 0xFFFFFFFFu;
    val[r] = valid ? vals_in[e] : 0u;
  }
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int e = base + r * 64 + lane;
    const bool valid = e < n;
    const uint32_t digit = (key[r] >> shift) & (LT_RD - 1u);
    unsigned long long m = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < LT_RB; ++bit) {
      const bool b = (digit >> bit) & 1u;
      const unsigned long long bal = __ballot(b);
      m &= b ? bal : ~bal;
    }
    if (valid) {
      const uint32_t prev = mycnt[digit];
      rank[r] = prev + (uint32_t)__popcll(m & lt_mask);
      if ((m & lt_mask) == 0ull) mycnt[digit] = prev + (uint32_t)__popcll(m);  // group leader
    } else {
      rank[r] = 0;
    }
  }
  __syncthreads();
  {  // thread t owns digits 4t .. 4t+3: digit base (exclusive scan of the totals) + tile base + prefix over waves
    const int d0 = threadIdx.x * LT_DPT;
    uint32_t tot[LT_DPT], sum = 0;
#pragma unroll
    for (int k = 0; k < LT_DPT; ++k) {
      tot[k] = hist[(size_t)LT_RD * nb + d0 + k];
      sum += tot[k];
    }
    uint32_t inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(inc, o, WAVE);
      if (lane >= o) inc += t;
    }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    uint32_t dbase = inc - sum;
    for (int w = 0; w < wave; ++w) dbase += wtot[w];
#pragma unroll
    for (int k = 0; k < LT_DPT; ++k) {
      uint32_t run = dbase + hist[(size_t)(d0 + k) * nb + blockIdx.x];
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const uint32_t c = cnt[w][d0 + k];
        cnt[w][d0 + k] = run;
        run += c;
      }
      dbase += tot[k];
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int e = base + r * 64 + lane;
    if (e < n) {
      const uint32_t pos = cnt[wave][(key[r] >> shift) & (LT_RD - 1u)] + rank[r];
      keys_out[pos] = key[r];
      vals_out[pos] = val[r];
    }
  }
}

// ---- sorted triangle records + padded leaf boxes -----------------------------------------------------
__global__ __launch_bounds__(256) void k_gather(const float* __restrict__ verts, const int* __restrict__ faces,
                                                int n_verts, int n_faces, int np,
                                                const uint32_t* __restrict__ vals, const float* __restrict__ params,
                                                float4* __restrict__ tris, float4* __restrict__ seg) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= np) return;
  float4 lo = make_float4(INFINITY, INFINITY, INFINITY, 0.f), hi = make_float4(-INFINITY, -INFINITY, -INFINITY, 0.f);
  if (p < n_faces) {
    const int f = (int)vals[p];
    const int a = faces[3 * (size_t)f], b = faces[3 * (size_t)f + 1], c = faces[3 * (size_t)f + 2];
    float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0, q2 = q0;
    q2.y = __int_as_float(f);
    if ((unsigned)a < (unsigned)n_verts && (unsigned)b < (unsigned)n_verts && (unsigned)c < (unsigned)n_verts) {
      const float pad = params[4];
      const float* pa = verts + 3 * (size_t)a;
      const float* pb = verts + 3 * (size_t)b;
      const float* pc = verts + 3 * (size_t)c;
      const float v0x = pa[0], v0y = pa[1], v0z = pa[2];
      const float v1x = pb[0], v1y = pb[1], v1z = pb[2];
      const float v2x = pc[0], v2y = pc[1], v2z = pc[2];
      q0 = make_float4(v0x, v0y, v0z, v1x - v0x);
      q1 = make_float4(v1y - v0y, v1z - v0z, v2x - v0x, v2y - v0y);
      q2.x = v2z - v0z;
      lo.x = fminf(v0x, fminf(v1x, v2x)) - pad; lo.y = fminf(v0y, fminf(v1y, v2y)) - pad;
      lo.z = fminf(v0z, fminf(v1z, v2z)) - pad;
      hi.x = fmaxf(v0x, fmaxf(v1x, v2x)) + pad; hi.y = fmaxf(v0y, fmaxf(v1y, v2y)) + pad;
      hi.z = fmaxf(v0z, fmaxf(v1z, v2z)) + pad;
    }
    tris[3 * (size_t)p] = q0;
    tris[3 * (size_t)p + 1] = q1;
    tris[3 * (size_t)p + 2] = q2;
  }
  seg[2 * ((size_t)np + p)] = lo;
  seg[2 * ((size_t)np + p) + 1] = hi;
}

__device__ __forceinline__ void box_union(float4& lo, float4& hi, const float4 l2, const float4 h2) {
  lo.x = fminf(lo.x, l2.x); lo.y = fminf(lo.y, l2.y); lo.z = fminf(lo.z, l2.z);
  hi.x = fmaxf(hi.x, h2.x); hi.y = fmaxf(hi.y, h2.y); hi.z = fmaxf(hi.z, h2.z);
}

// ---- segment tree: each workgroup reduces `sub` consecutive leaves up to their common ancestor ----------
__global__ __launch_bounds__(256) void k_seg_sub(float4* __restrict__ seg, int np, int sub) {
  __shared__ float4 slo[LT_SEG_SUB / 2], shi[LT_SEG_SUB / 2];
  const int t = threadIdx.x;
  int cnt = sub >> 1;                                  // nodes at the level above the leaves
  size_t first = ((size_t)np + (size_t)blockIdx.x * sub) >> 1;  // heap index of the first of them
  float4 lo, hi;
  if (t < cnt) {
    const size_t k = first + t;
    lo = seg[2 * (2 * k)]; hi = seg[2 * (2 * k) + 1];
    box_union(lo, hi, seg[2 * (2 * k + 1)], seg[2 * (2 * k + 1) + 1]);
    seg[2 * k] = lo; seg[2 * k + 1] = hi;
    slo[t] = lo; shi[t] = hi;
  }
  __syncthreads();
  while (cnt > 1) {
    cnt >>= 1;
    first >>= 1;
    if (t < cnt) {
      lo = slo[2 * t]; hi = shi[2 * t];
      box_union(lo, hi, slo[2 * t + 1], shi[2 * t + 1]);
    }
    __syncthreads();
    if (t < cnt) {
      slo[t] = lo; shi[t] = hi;
      seg[2 * (first + t)] = lo; seg[2 * (first + t) + 1] = hi;
    }
    __syncthreads();
  }
}

// top of the segment tree: one workgroup, level by level through global memory
__global__ __launch_bounds__(1024) void k_seg_top(float4* __restrict__ seg, int m) {
  for (int cnt = m >> 1; cnt >= 1; cnt >>= 1) {
    for (int j = threadIdx.x; j < cnt; j += 1024) {
      const size_t k = (size_t)cnt + j;
      float4 lo = seg[2 * (2 * k)], hi = seg[2 * (2 * k) + 1];
      box_union(lo, hi, seg[2 * (2 * k + 1)], seg[2 * (2 * k + 1) + 1]);
      seg[2 * k] = lo; seg[2 * k + 1] = hi;
    }
    __threadfence_block();
    __syncthreads();
  }
}

// ---- Karras topology + child boxes --------------------------------------------------------------------
__device__ __forceinline__ int delta(const uint32_t* __restrict__ keys, int n, int i, uint32_t ki, int j) {
  if (j < 0 || j >= n) return -1;
  const uint32_t kj = keys[j];
  return ki != kj ? __clz((int)(ki ^ kj)) : 32 + __clz(i ^ j);
}

// Box of the sorted leaves [a, b]: the standard bottom-up walk over the heap-ordered segment tree.  (Issuing the loads
// of several levels together -- which nodes the walk takes depends on a and b only -- was measured: four levels per
// batch cut the mean wave life from 31 to 24 us but cost 86 VGPRs and a third round of workgroups, 122 us instead of
// 110; two levels per batch, 56 VGPRs, changed nothing.)
__device__ __forceinline__ void range_box(const float4* __restrict__ seg, int np, int a, int b, float4& lo,
                                          float4& hi) {
  lo = make_float4(INFINITY, INFINITY, INFINITY, 0.f);
  hi = make_float4(-INFINITY, -INFINITY, -INFINITY, 0.f);
  int l = a + np, r = b + np + 1;
  while (l < r) {
    if (l & 1) { box_union(lo, hi, seg[2 * (size_t)l], seg[2 * (size_t)l + 1]); ++l; }
    if (r & 1) { --r; box_union(lo, hi, seg[2 * (size_t)r], seg[2 * (size_t)r + 1]); }
    l >>= 1;
    r >>= 1;
  }
}

__device__ __forceinline__ int leaf_ref(int start, int cnt) { return ~(start | ((cnt - 1) << 28)); }

__global__ __launch_bounds__(256) void k_hierarchy(const uint32_t* __restrict__ keys, int n, int np,
                                                   const float4* __restrict__ seg, float4* __restrict__ nodes) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (n == 1) {
    if (i == 0) {
      const float4 lo = seg[2 * (size_t)np], hi = seg[2 * (size_t)np + 1];
      const int c = leaf_ref(0, 1);
      nodes[0] = make_float4(lo.x, lo.y, lo.z, hi.x);
      nodes[1] = make_float4(hi.y, hi.z, INFINITY, INFINITY);   // second child: mn = mx = +inf, never hit
      nodes[2] = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
      nodes[3] = make_float4(__int_as_float(c), __int_as_float(c), 0.f, 0.f);
    }
    return;
  }
  if (i >= n - 1) return;
  const uint32_t ki = keys[i];
  const int d = (delta(keys, n, i, ki, i + 1) - delta(keys, n, i, ki, i - 1)) >= 0 ? 1 : -1;
  const int dmin = delta(keys, n, i, ki, i - d);
  int lmax = 2;
  while (delta(keys, n, i, ki, i + lmax * d) > dmin) lmax <<= 1;
  int l = 0;
  for (int t = lmax >> 1; t >= 1; t >>= 1)
    if (delta(keys, n, i, ki, i + (l + t) * d) > dmin) l += t;
  const int j = i + l * d;
  const int dn = delta(keys, n, i, ki, j);
  int s = 0;
  for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
    if (delta(keys, n, i, ki, i + (s + t) * d) > dn) s += t;
    if (t == 1) break;
  }
  const int g = i + s * d + (d < 0 ? -1 : 0);
  const int first = min(i, j), last = max(i, j);
  if (last - first + 1 <= LT_LEAF_MAX && i != 0) {  // swallowed by a leaf of an ancestor: mark dead
    nodes[4 * (size_t)i + 3] = make_float4(__int_as_float(0x7fffffff), __int_as_float(0x7fffffff), 0.f, 0.f);
    return;
  }
  const int lc = g - first + 1, rc = last - g;
  const int c0 = lc <= LT_LEAF_MAX ? leaf_ref(first, lc) : g;
  const int c1 = rc <= LT_LEAF_MAX ? leaf_ref(g + 1, rc) : g + 1;
  float4 l0, h0, l1, h1;
  range_box(seg, np, first, g, l0, h0);
  range_box(seg, np, g + 1, last, l1, h1);
  float4* N = nodes + 4 * (size_t)i;
  N[0] = make_float4(l0.x, l0.y, l0.z, h0.x);
  N[1] = make_float4(h0.y, h0.z, l1.x, l1.y);
  N[2] = make_float4(l1.z, h1.x, h1.y, h1.z);
  N[3] = make_float4(__int_as_float(c0), __int_as_float(c1), 0.f, 0.f);
}

// ---- Karras topology straight into 4-wide nodes (default path) -------------------------------------------
// Node i finds its own key range and split as k_hierarchy does; because the ranges of its two children
// are then known, their splits need no range search -- one binary search each -- and the up to four
// grandchild ranges become the entries of the 128-B node.  The sorted keys around the workgroup are
// staged in LDS and every search of k_hierarchy4 stays inside that window.
#define LT_HWIN 512  // keys staged on each side of the workgroup's nodes (>= 2 LT_HBIG)

struct key_window {
  const uint32_t* lds;  // window copy
  int lo, hi, n;        // window = [lo, hi) of the n sorted keys
};

__device__ __forceinline__ void put_entry(float4* __restrict__ O, int k, const float4 lo, const float4 hi, int ref) {
  O[2 * k] = make_float4(lo.x, lo.y, lo.z, hi.x);
  O[2 * k + 1] = make_float4(hi.y, hi.z, __int_as_float(ref), 0.f);
}

// What bounds this kernel is the LONGEST chain in a wave, not the work: splits are binary searches and range boxes are
// level-by-level walks, both ~log2(range) steps, and every 64 consecutive Karras nodes hold ranges of all sizes.  The
// one-thread-per-node kernel (round 1) ran 15 624 waves of ~30 us in two rounds, the slowest 90 us, 110 us in all
// (per-wave wall-clock stamps: LIDARHIP_DEBUG_HIER=1 python tools/wave_times.py --hier).  Now, on the 1 M-triangle mesh:
//   * a workgroup takes LT_HNODES consecutive nodes.  Phase 1a (3 probes a node) finds the survivors: three quarters of
//     all Karras nodes span <= LT_LEAF_MAX leaves, are swallowed by a leaf of an ancestor and end there.  Phase 1b finds
//     the survivors' key ranges with every lane busy; nodes spanning more than LT_HBIG leaves are QUEUED for
//     k_hierarchy4_big -- one device-scope atomic per WORKGROUP (one per wave and pass, 12 000 on one hot word served at
//     ~130 per us, was a quarter of the kernel).  Phase 2 -- splits, grandchild ranges, range boxes -- walks the four
//     boxes of a node TOGETHER (range_box4): 128 VGPRs, which is exactly what keeps all 977 workgroups resident at once.
//     Stamps: phase 1 ends at 14 us, the splits at 22 us, the boxes (<= 8 levels x ~1.5 us of memory-side latency: the
//     64 MB segment tree is cold) at 34 us mean / 49 us max: 49 us.
//   * k_hierarchy4_big gives each of the ~8 k queued nodes a whole wave: 64-ary range and split searches (64 keys per
//     round trip, both children's splits side by side in the two half-waves) and range boxes whose 2 x 32 levels are
//     loaded at once (level l of the walk is closed-form: left ceil(l0 / 2^l), right floor(r0 / 2^l)) and reduced with
//     shuffles: range 2.7 us, splits 2.2 us, boxes 6.8 us per node, 18 us in all.
// Measured and rejected: LDS copies of the window's segment-tree levels (occupancy: a second round of workgroups);
// warming the XCD's L2 with the window's tree lines during phase 1 (85 us instead of 78); LT_HNODES = 256 with four
// times the workgroups.
#define LT_HNODES 1024
#define LT_HBIG 256
__device__ __forceinline__ int lw_delta(const key_window& kw, int i, uint32_t ki, int j) {
  if (j < 0 || j >= kw.n) return -1;
  const uint32_t kj = kw.lds[j - kw.lo];
  return ki != kj ? __clz((int)(ki ^ kj)) : 32 + __clz(i ^ j);
}
__device__ __forceinline__ int lw_split(const key_window& kw, int a, int b) {
  const uint32_t ka = kw.lds[a - kw.lo];
  const int dn = lw_delta(kw, a, ka, b);
  const int l = b - a;
  int s = 0;
  for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
    if (s + t < l && lw_delta(kw, a, ka, a + s + t) > dn) s += t;
    if (t == 1) break;
  }
  return a + s;
}
struct lt_f3 { float x, y, z; };
// The boxes of up to four sorted-leaf ranges, walked level by level TOGETHER: which heap nodes a walk takes depends on
// its end points only, so the (up to 8) loads of a level are independent and in flight at once -- the chain is the
// deepest walk, not the sum of the four.
__device__ __forceinline__ void range_box4(const float4* __restrict__ seg, int np, const int* a, const int* b,
                                           float4* lo, float4* hi) {
  int l[4], r[4];
  bool any = false;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    lo[k] = make_float4(INFINITY, INFINITY, INFINITY, 0.f);
    hi[k] = make_float4(-INFINITY, -INFINITY, -INFINITY, 0.f);
    l[k] = a[k] + np;  // (an empty slot has b = a - 1: l == r, no step)
    r[k] = b[k] + np + 1;
    any |= l[k] < r[k];
  }
  while (any) {
    lt_f3 tl[8], th[8];  // (12-byte loads: the w lanes of the tree's float4s carry nothing)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool live = l[k] < r[k];
      const bool tk_l = live && (l[k] & 1), tk_r = live && (r[k] & 1);
      tl[2 * k] = lt_f3{INFINITY, INFINITY, INFINITY}; th[2 * k] = lt_f3{-INFINITY, -INFINITY, -INFINITY};
      tl[2 * k + 1] = tl[2 * k]; th[2 * k + 1] = th[2 * k];
      if (tk_l) { tl[2 * k] = *(const lt_f3*)&seg[2 * (size_t)l[k]]; th[2 * k] = *(const lt_f3*)&seg[2 * (size_t)l[k] + 1]; }
      if (tk_r) { tl[2 * k + 1] = *(const lt_f3*)&seg[2 * (size_t)(r[k] - 1)]; th[2 * k + 1] = *(const lt_f3*)&seg[2 * (size_t)(r[k] - 1) + 1]; }
      l[k] = (l[k] + (tk_l ? 1 : 0)) >> 1;
      r[k] = (r[k] - (tk_r ? 1 : 0)) >> 1;
    }
    any = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        lo[k].x = fminf(lo[k].x, tl[2 * k + f].x); lo[k].y = fminf(lo[k].y, tl[2 * k + f].y); lo[k].z = fminf(lo[k].z, tl[2 * k + f].z);
        hi[k].x = fmaxf(hi[k].x, th[2 * k + f].x); hi[k].y = fmaxf(hi[k].y, th[2 * k + f].y); hi[k].z = fmaxf(hi[k].z, th[2 * k + f].z);
      }
      any |= l[k] < r[k];
    }
  }
}

__device__ __forceinline__ void hier_emit(const key_window& kw, const float4* __restrict__ seg, int np,
                                          float4* __restrict__ nodes4, int i, int first, int last, bool stamp,
                                          unsigned long long& t_split) {
  const float inf = INFINITY;
  const float4 e_lo = make_float4(inf, inf, inf, 0.f), e_hi = make_float4(inf, inf, inf, 0.f);  // mn = mx = +inf
  const int g = lw_split(kw, first, last);
  float4* O = nodes4 + 8 * (size_t)i;
  // slots 0, 1 belong to the left child, 2, 3 to the right one; a child that is a leaf uses its first slot only (the
  // other stays an empty range, which the walk skips); entries are written compacted, in slot order
  int ra[4], rb[4], ref[4];
#pragma unroll
  for (int side = 0; side < 2; ++side) {
    const int a = side == 0 ? first : g + 1, b = side == 0 ? g : last;
    if (b - a + 1 <= LT_LEAF_MAX) {
      ra[2 * side] = a; rb[2 * side] = b; ref[2 * side] = leaf_ref(a, b - a + 1);
      ra[2 * side + 1] = 0; rb[2 * side + 1] = -1; ref[2 * side + 1] = 0;
    } else {
      const int gc = lw_split(kw, a, b);
      const int lc = gc - a + 1, rc = b - gc;
      ra[2 * side] = a; rb[2 * side] = gc; ref[2 * side] = lc <= LT_LEAF_MAX ? leaf_ref(a, lc) : gc;
      ra[2 * side + 1] = gc + 1; rb[2 * side + 1] = b; ref[2 * side + 1] = rc <= LT_LEAF_MAX ? leaf_ref(gc + 1, rc) : gc + 1;
    }
  }
  if (stamp) t_split = (unsigned long long)wall_clock64();
  float4 lo[4], hi[4];
  range_box4(seg, np, ra, rb, lo, hi);
  int ne = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (rb[k] >= ra[k]) put_entry(O, ne++, lo[k], hi[k], ref[k]);
  for (; ne < 4; ++ne) put_entry(O, ne, e_lo, e_hi, 0x7fffffff);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_hierarchy4(const uint32_t* __restrict__ keys, int n, int np,
                                                    const float4* __restrict__ seg, float4* __restrict__ nodes4,
                                                    int* __restrict__ big_queue, int big_cap, unsigned* __restrict__ big_count,
                                                    unsigned long long* __restrict__ dbg) {
  __shared__ uint32_t wkeys[LT_HNODES + 2 * LT_HWIN];
  __shared__ int sv_i[LT_HNODES], sv_first[LT_HNODES], sv_last[LT_HNODES];
  __shared__ int n_surv, n_big, n_big_out, big_base;
  const unsigned long long t_dbg = dbg ? (unsigned long long)wall_clock64() : 0ull;
  const int i0 = blockIdx.x * LT_HNODES;
  key_window kw;
  kw.lds = wkeys; kw.n = n;
  kw.lo = max(i0 - LT_HWIN, 0);
  kw.hi = min(i0 + LT_HNODES + LT_HWIN, n);
  for (int k = threadIdx.x; k < kw.hi - kw.lo; k += 256) wkeys[k] = keys[kw.lo + k];
  if (threadIdx.x == 0) { n_surv = 0; n_big = 0; n_big_out = 0; }
  __syncthreads();
  if (n == 1) {
    if (i0 == 0 && threadIdx.x == 0) {
      const float inf = INFINITY;
      const float4 e_lo = make_float4(inf, inf, inf, 0.f), e_hi = make_float4(inf, inf, inf, 0.f);
      put_entry(nodes4, 0, seg[2 * (size_t)np], seg[2 * (size_t)np + 1], leaf_ref(0, 1));
      for (int k = 1; k < 4; ++k) put_entry(nodes4, k, e_lo, e_hi, 0x7fffffff);
    }
    return;
  }
  // ---- phase 1a: which nodes survive?  Node i spans more than LT_LEAF_MAX leaves exactly when delta(i, i + LT_LEAF_MAX d)
  // > dmin (delta is monotone along d) -- three probes; the other three quarters of the nodes are swallowed by a leaf of
  // an ancestor, never referenced, and end here.  Survivors (i, d, dmin) are compacted in LDS.
  for (int k = 0; k < LT_HNODES / 256; ++k) {
    const int i = i0 + k * 256 + threadIdx.x;
    bool keep = false;
    int d = 0, dmin = 0;
    if (i < n - 1) {
      const uint32_t ki = wkeys[i - kw.lo];
      const int dp = lw_delta(kw, i, ki, i + 1), dm = lw_delta(kw, i, ki, i - 1);
      d = dp - dm >= 0 ? 1 : -1;
      dmin = d > 0 ? dm : dp;  // = delta(i, i - d)
      keep = i == 0 || lw_delta(kw, i, ki, i + LT_LEAF_MAX * d) > dmin;
    }
    const unsigned long long ms = __ballot(keep);
    int base_s = 0;
    if ((threadIdx.x & 63) == 0 && ms) base_s = atomicAdd(&n_surv, __popcll(ms));
    base_s = __shfl(base_s, 0, 64);
    if (keep) {
      const int slot = base_s + __popcll(ms & ((1ull << (threadIdx.x & 63)) - 1ull));
      sv_i[slot] = i; sv_first[slot] = d; sv_last[slot] = dmin;
    }
  }
  __syncthreads();
  // ---- phase 1b: the survivors' key ranges, every lane busy.  The doubling search stops at LT_HBIG: up to there every
  // probe is inside the LDS key window; a node whose range reaches further is handed to k_hierarchy4_big with (d, dmin)
  // -- ITS range search is the long chain of dependent global loads that every wave used to wait for.
  for (int sidx = threadIdx.x; sidx < n_surv; sidx += 256) {
    const int i = sv_i[sidx], d = sv_first[sidx], dmin = sv_last[sidx];
    const uint32_t ki = wkeys[i - kw.lo];
    // (known: delta(i, i + d) > dmin by the choice of d, and delta(i, i + LT_LEAF_MAX d) > dmin for every survivor but a
    // root with fewer leaves)
    int lmax = i == 0 ? 2 : 2 * LT_LEAF_MAX;
    while (lmax <= LT_HBIG && lw_delta(kw, i, ki, i + lmax * d) > dmin) lmax <<= 1;
    if (lmax > LT_HBIG) {
      atomicAdd(&n_big, 1);
      sv_i[sidx] = ~i;  // queued below with (d, dmin), which stay in the slot; not for phase 2
    } else {
      int l = lmax >> 1;  // the last probe that held
      for (int t = lmax >> 2; t >= 1; t >>= 1)
        if (lw_delta(kw, i, ki, i + (l + t) * d) > dmin) l += t;
      const int j = i + l * d;
      sv_first[sidx] = min(i, j);
      sv_last[sidx] = max(i, j);
    }
  }
  __syncthreads();
  // the workgroup's big nodes go to the global queue with ONE device-scope atomic (see above).  The queue cannot
  // overflow: the ranges of one tree level are disjoint, so a level holds fewer than n / LT_HBIG big nodes, and the tree
  // is at most 30 key bits + 28 index bits deep -- under n / 4 in all, the queue holds n / 3.
  if (threadIdx.x == 0 && n_big > 0) big_base = (int)atomicAdd(big_count, (unsigned)n_big);
  __syncthreads();
  if (n_big > 0)
    for (int sidx = threadIdx.x; sidx < n_surv; sidx += 256)
      if (sv_i[sidx] < 0) {
        const int slot = big_base + atomicAdd(&n_big_out, 1);
        if (slot < big_cap) { big_queue[3 * slot] = ~sv_i[sidx]; big_queue[3 * slot + 1] = sv_first[sidx]; big_queue[3 * slot + 2] = sv_last[sidx]; }
      }
  // ---- phase 2: the small surviving nodes, one per thread
  unsigned long long t_p1 = dbg ? (unsigned long long)wall_clock64() : 0ull, t_split = 0ull;
  for (int sidx = threadIdx.x; sidx < n_surv; sidx += 256)
    if (sv_i[sidx] >= 0) hier_emit(kw, seg, np, nodes4, sv_i[sidx], sv_first[sidx], sv_last[sidx], dbg != nullptr, t_split);
  if (dbg && (threadIdx.x & 63) == 0) {
    const int wg = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wg < LT_DBG_WAVES) { dbg[8 + 2 * wg] = t_dbg; dbg[8 + 2 * wg + 1] = (unsigned long long)wall_clock64() - t_dbg; }
    if (wg < LT_DBG_WAVES) { dbg[8 + 2 * LT_DBG_WAVES + 2 * wg] = t_p1 - t_dbg; dbg[8 + 2 * LT_DBG_WAVES + 2 * wg + 1] = t_split ? t_split - t_dbg : 0ull; }
  }
}

// ---- one WAVE per big node ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int g_delta(const uint32_t* __restrict__ keys, int i, uint32_t ki, int j) {  // 0 <= j < n
  const uint32_t kj = keys[j];
  return ki != kj ? __clz((int)(ki ^ kj)) : 32 + __clz(i ^ j);
}
// split of [a, b] (b > a), all lanes take part: the predicate delta(a, k) > delta(a, b) holds for k <= g and fails for
// k > g; 63 positions are probed per round
__device__ __forceinline__ int wave_split(const uint32_t* __restrict__ keys, int a, int b) {
  const int lane = threadIdx.x & 63;
  const uint32_t ka = keys[a];
  const int dn = g_delta(keys, a, ka, b);
  int lo = a, hi = b;  // predicate true at lo, false at hi
  while (hi - lo > 1) {
    const int step = (hi - lo + 62) / 63;
    const int k = lo + (lane + 1) * step;  // lanes 0 .. 62 probe lo + step .. ; lane 63 idles
    const bool ok = lane < 63 && k < hi && g_delta(keys, a, ka, k) > dn;
    const unsigned long long m = __ballot(ok);
    const int t = m ? 64 - __clzll((long long)m) : 0;  // number of leading true probes (the predicate is monotone)
    const int nlo = lo + t * step;
    hi = min(hi, nlo + step);
    lo = nlo;
  }
  return lo;
}
__global__ __launch_bounds__(256) void k_hierarchy4_big(const uint32_t* __restrict__ keys, int n, int np,
                                                        const float4* __restrict__ seg, float4* __restrict__ nodes4,
                                                        const int* __restrict__ big_queue, int big_cap,
                                                        const unsigned* __restrict__ big_count,
                                                        unsigned long long* __restrict__ dbg) {
  const int lane = threadIdx.x & 63;
  const int n_big = min((int)*big_count, big_cap);
  const float inf = INFINITY;
  const float4 e_lo = make_float4(inf, inf, inf, 0.f), e_hi = make_float4(inf, inf, inf, 0.f);
  for (int q = blockIdx.x * 4 + (threadIdx.x >> 6); q < n_big; q += gridDim.x * 4) {
    const unsigned long long t0 = dbg ? (unsigned long long)wall_clock64() : 0ull;
    const int i = big_queue[3 * q], d = big_queue[3 * q + 1], dmin = big_queue[3 * q + 2];
    // the node's range: largest L with delta(i, i + L d) > dmin (monotone in L).  Lanes probe L = 2^lane at once, then a
    // 64-ary search inside [2^T, 2^(T+1)).
    const uint32_t ki = keys[i];
    int L;
    {
      const long long j0 = (long long)i + ((long long)d << min(lane, 31));
      const bool ok0 = lane < 31 && j0 >= 0 && j0 < n && g_delta(keys, i, ki, (int)j0) > dmin;
      const unsigned long long m0 = __ballot(ok0);
      const int T = 63 - __clzll((long long)m0);  // >= 8: delta(i, i + LT_HBIG d) > dmin brought the node here
      long long lo = 1ll << T, hi = 2ll << T;     // predicate true at lo, false (or out of range) at hi
      while (hi - lo > 1) {
        const long long step = (hi - lo + 62) / 63;
        const long long Lk = lo + (long long)(lane + 1) * step;
        const long long jk = (long long)i + Lk * d;
        const bool ok = lane < 63 && Lk < hi && jk >= 0 && jk < n && g_delta(keys, i, ki, (int)jk) > dmin;
        const unsigned long long m = __ballot(ok);
        const int t = m ? 64 - __clzll((long long)m) : 0;
        const long long nlo = lo + t * step;
        hi = hi < nlo + step ? hi : nlo + step;
        lo = nlo;
      }
      L = (int)lo;
    }
    const int j = i + L * d;
    const int first = min(i, j), last = max(i, j);
    const unsigned long long t1 = dbg ? (unsigned long long)wall_clock64() : 0ull;
    const int g = wave_split(keys, first, last);
    const unsigned long long t2 = dbg ? (unsigned long long)wall_clock64() : 0ull;
    // both children at once: lanes 0 .. 31 search the split of the left child, lanes 32 .. 63 that of the right one (31
    // probes per round trip; a child of <= LT_LEAF_MAX leaves is a leaf and idles)
    const int half = lane >> 5, hl = lane & 31;
    const int ca = half ? g + 1 : first, cb = half ? last : g;
    const bool need = cb - ca + 1 > LT_LEAF_MAX;
    int gc;
    {
      const uint32_t ka = keys[ca], kb = keys[cb];
      const int dn = ka != kb ? __clz((int)(ka ^ kb)) : 32 + __clz(ca ^ cb);
      int lo = ca, hi = need ? cb : ca;  // predicate true at lo, false at hi
      while (__any(hi - lo > 1)) {
        const int step = (hi - lo + 30) / 31;
        const int k = lo + (hl + 1) * step;
        const bool ok = hl < 31 && k < hi && g_delta(keys, ca, ka, k) > dn;
        const unsigned long long m = __ballot(ok);
        const unsigned mh = (unsigned)(m >> (32 * half));
        const int t = mh ? 32 - __clz((int)mh) : 0;
        if (hi - lo > 1) {
          const int nlo = lo + t * step;
          hi = min(hi, nlo + step);
          lo = nlo;
        }
      }
      gc = lo;
    }
    const unsigned long long t3 = dbg ? (unsigned long long)wall_clock64() : 0ull;
    const int gcl = __builtin_amdgcn_readlane(gc, 0), gcr = __builtin_amdgcn_readlane(gc, 32);  // (scalars from here on)
    const bool needl = first != g && g - first + 1 > LT_LEAF_MAX, needr = last - g > LT_LEAF_MAX;
    // slots 0, 1: left child (or its two halves); 2, 3: right child; an unused slot is the empty range [0, -1]
    int ra[4], rb[4], ref[4];
    ra[0] = first; rb[0] = needl ? gcl : g;
    ref[0] = needl ? (gcl - first + 1 <= LT_LEAF_MAX ? leaf_ref(first, gcl - first + 1) : gcl) : leaf_ref(first, g - first + 1);
    ra[1] = needl ? gcl + 1 : 0; rb[1] = needl ? g : -1;
    ref[1] = g - gcl <= LT_LEAF_MAX ? leaf_ref(gcl + 1, max(g - gcl, 1)) : gcl + 1;
    ra[2] = g + 1; rb[2] = needr ? gcr : last;
    ref[2] = needr ? (gcr - g <= LT_LEAF_MAX ? leaf_ref(g + 1, gcr - g) : gcr) : leaf_ref(g + 1, last - g);
    ra[3] = needr ? gcr + 1 : 0; rb[3] = needr ? last : -1;
    ref[3] = last - gcr <= LT_LEAF_MAX ? leaf_ref(gcr + 1, max(last - gcr, 1)) : gcr + 1;
    // the range boxes two at a time (four would cost the registers of a resident wave per SIMD): lane (flank, level)
    // loads its heap node of both ranges -- 4 x 12 bytes in flight per lane -- then one round of shuffles
    const int lev = lane & 31;
    float4* O = nodes4 + 8 * (size_t)i;
    int ne = 0;
#pragma unroll 1
    for (int pair = 0; pair < 2; ++pair) {
      lt_f3 lo[2], hi[2];
      int pa[2], pb[2], pref[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        pa[k] = pair ? ra[2 + k] : ra[k]; pb[k] = pair ? rb[2 + k] : rb[k]; pref[k] = pair ? ref[2 + k] : ref[k];
        lo[k] = lt_f3{INFINITY, INFINITY, INFINITY};
        hi[k] = lt_f3{-INFINITY, -INFINITY, -INFINITY};
        const long long l0 = (long long)pa[k] + np, r0 = (long long)pb[k] + np + 1;
        const long long ll = (l0 + ((1ll << lev) - 1)) >> lev, rl = r0 >> lev;
        if (ll < rl) {
          if (lane < 32 && (ll & 1)) { lo[k] = *(const lt_f3*)&seg[2 * (size_t)ll]; hi[k] = *(const lt_f3*)&seg[2 * (size_t)ll + 1]; }
          if (lane >= 32 && (rl & 1)) { lo[k] = *(const lt_f3*)&seg[2 * (size_t)(rl - 1)]; hi[k] = *(const lt_f3*)&seg[2 * (size_t)(rl - 1) + 1]; }
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          lo[k].x = fminf(lo[k].x, __shfl_xor(lo[k].x, o, 64)); lo[k].y = fminf(lo[k].y, __shfl_xor(lo[k].y, o, 64)); lo[k].z = fminf(lo[k].z, __shfl_xor(lo[k].z, o, 64));
          hi[k].x = fmaxf(hi[k].x, __shfl_xor(hi[k].x, o, 64)); hi[k].y = fmaxf(hi[k].y, __shfl_xor(hi[k].y, o, 64)); hi[k].z = fmaxf(hi[k].z, __shfl_xor(hi[k].z, o, 64));
        }
      }
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if (pb[k] >= pa[k]) {  // (wave-uniform)
          if (lane == 0) put_entry(O, ne, make_float4(lo[k].x, lo[k].y, lo[k].z, 0.f), make_float4(hi[k].x, hi[k].y, hi[k].z, 0.f), pref[k]);
          ++ne;
        }
    }
    if (lane == 0)
      for (; ne < 4; ++ne) put_entry(O, ne, e_lo, e_hi, 0x7fffffff);
    if (dbg && lane == 0 && q < 4096) {  // debug (LIDARHIP_DEBUG_HIER=1): start and the ends of range / split / child splits / boxes
      unsigned long long* o = dbg + 8 + 8192 + 6 * (size_t)q;
      o[0] = t0; o[1] = t1 - t0; o[2] = t2 - t0; o[3] = t3 - t0; o[4] = (unsigned long long)wall_clock64() - t0; o[5] = (unsigned long long)(last - first + 1);
    }
  }
}

// ---- host orchestration -----------------------------------------------------------------------------
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Stable LSD radix sort of n (key, value) pairs on the low `key_bits` bits; keys[0]/vals[0] hold the input,
// *out_buffer tells which of the two ping-pong buffers holds the result.  hist: LT_RD * tiles + LT_RD words.
void lt_sort_pairs(uint32_t* keys[2], uint32_t* vals[2], uint32_t* hist, int n, int key_bits, hipStream_t stream,
                   int* out_buffer) {
  const int nb = cdiv(n, LT_SORT_TILE);
  int cur = 0;
  for (int pass = 0; pass * LT_RB < key_bits; ++pass) {
    const int shift = LT_RB * pass;
    hipLaunchKernelGGL(k_hist, dim3(nb), dim3(LT_SORT_THREADS), 0, stream, keys[cur], n, shift, hist, nb);
    hipLaunchKernelGGL(k_scan, dim3(LT_RD), dim3(256), 0, stream, hist, nb);
    hipLaunchKernelGGL(k_scatter, dim3(nb), dim3(LT_SORT_THREADS), 0, stream, keys[cur], vals[cur], keys[cur ^ 1],
                       vals[cur ^ 1], n, shift, hist, nb);
    cur ^= 1;
  }
  *out_buffer = cur;
}

int lt_build_launch(lt_scene* s, hipStream_t stream, lt_stats* stats) {
  const int n = s->n_faces;
  s->built = 0;
  s->last_stream = stream;
  if (n > LT_MAX_FACES - 1) {
    lt_set_error("lt_scene_build: %d faces exceed LT_MAX_FACES", n);
    return LT_ERR_TOO_LARGE;
  }
  LT_CHECK(lt_scene_reserve(s, n));
  const bool timed = stats != nullptr;
  int ev = 0;
#define LT_MARK()                                              \
  do {                                                         \
    if (timed) LT_HIP(hipEventRecord(s->ev[ev++], stream));    \
  } while (0)
  LT_MARK();  // 0
  if (n > 0) {
    int np = 1;
    while (np < n) np <<= 1;
    s->np = np;
    hipLaunchKernelGGL(k_bounds, dim3(LT_BOUNDS_BLOCKS), dim3(256), 0, stream, s->verts, s->n_verts, s->partial);
    LT_MARK();  // 1
    hipLaunchKernelGGL(k_morton, dim3(cdiv(n, 256)), dim3(256), 0, stream, s->verts, s->faces, s->n_verts, n,
                       s->partial, s->params, s->keys[0], s->vals[0], s->flags);
    LT_MARK();  // 2
    int cur = 0;
    lt_sort_pairs(s->keys, s->vals, s->hist, n, 30, stream, &cur);
    // sorted data is in buffer `cur`
    LT_MARK();  // 3
    hipLaunchKernelGGL(k_gather, dim3(cdiv(np, 256)), dim3(256), 0, stream, s->verts, s->faces, s->n_verts, n, np,
                       s->vals[cur], s->params, s->tris, s->seg);
    LT_MARK();  // 4
    if (np >= 2) {
      const int sub = np < LT_SEG_SUB ? np : LT_SEG_SUB;
      hipLaunchKernelGGL(k_seg_sub, dim3(np / sub), dim3(256), 0, stream, s->seg, np, sub);
      if (np / sub > 1) hipLaunchKernelGGL(k_seg_top, dim3(1), dim3(1024), 0, stream, s->seg, np / sub);
    }
    LT_MARK();  // 5
    if (lt_binary_path()) {  // A/B: binary nodes for k_trace (one ray per lane)
      hipLaunchKernelGGL(k_hierarchy, dim3(cdiv(n > 1 ? n - 1 : 1, 256)), dim3(256), 0, stream, s->keys[cur], n, np,
                         s->seg, s->nodes);
    } else {
      static const bool dbg_hier = getenv("LIDARHIP_DEBUG_HIER") != nullptr;
      // queue of the nodes whose range search leaves the LDS key window: (i, d, dmin) triples in the sort's spare key buffer,
      // counter in flags[2]
      int* big_queue = (int*)s->keys[cur ^ 1];
      const int big_cap = s->cap_faces / 3;
      LT_HIP(hipMemsetAsync(s->flags + 2, 0, sizeof(unsigned), stream));
      hipLaunchKernelGGL(k_hierarchy4, dim3(cdiv(n > 1 ? n - 1 : 1, LT_HNODES)), dim3(256), 0, stream, s->keys[cur], n,
                         np, s->seg, s->nodes4, big_queue, big_cap, s->flags + 2, dbg_hier ? s->counters : nullptr);
      if (n > LT_HBIG)
        hipLaunchKernelGGL(k_hierarchy4_big, dim3(min(cdiv(2 * n / LT_HBIG + 4, 4), 2048)), dim3(256), 0, stream, s->keys[cur],
                           n, np, s->seg, s->nodes4, big_queue, big_cap, s->flags + 2, dbg_hier ? s->counters : nullptr);
    }
    LT_MARK();  // 6
    LT_HIP(hipGetLastError());
  }
  s->built = 1;
  s->stats.n_faces = n;
  s->stats.n_nodes = n > 1 ? n - 1 : (n == 1 ? 1 : 0);
  if (timed) {
    LT_HIP(hipStreamSynchronize(stream));
    float ms[6] = {0, 0, 0, 0, 0, 0};
    if (n > 0)
      for (int k = 0; k < 6; ++k) LT_HIP(hipEventElapsedTime(&ms[k], s->ev[k], s->ev[k + 1]));
    s->stats.ms_bounds = ms[0]; s->stats.ms_morton = ms[1]; s->stats.ms_sort = ms[2];
    s->stats.ms_gather = ms[3]; s->stats.ms_segtree = ms[4]; s->stats.ms_hierarchy = ms[5];
    float tot = 0.f;
    if (n > 0) LT_HIP(hipEventElapsedTime(&tot, s->ev[0], s->ev[6]));
    s->stats.ms_build = tot;
    *stats = s->stats;
  }
#undef LT_MARK
  return LT_OK;
}

// debug helpers (not part of the documented ABI): fill / fetch the 4-wide node array of a scene (tests pin the node array of
// seeded meshes by SHA-256: a change of the sort or of the hierarchy kernels must leave it bit-identical)
extern "C" int lt_debug_nodes4_fill(lt_scene* s, int byte) {
  if (!s || !s->nodes4) return LT_ERR_INVALID_ARG;
  LT_HIP(hipSetDevice(s->device));
  LT_HIP(hipDeviceSynchronize());
  LT_HIP(hipMemset(s->nodes4, byte, (size_t)s->cap_faces * 128));
  return LT_OK;
}
extern "C" int lt_debug_nodes4_get(lt_scene* s, void* out, int n_nodes) {
  if (!s || !s->nodes4 || !out || n_nodes < 0 || n_nodes > s->cap_faces) return LT_ERR_INVALID_ARG;
  LT_HIP(hipSetDevice(s->device));
  LT_HIP(hipDeviceSynchronize());
  LT_HIP(hipMemcpy(out, s->nodes4, (size_t)n_nodes * 128, hipMemcpyDeviceToHost));
  return LT_OK;
}
