// lt_trace.hip -- closest-hit ray cast on gfx950 (replaces the OpenMP ray loop of the reference,
// auxiliary/raytracer/RayTracer.cpp:62-92, with BVH::getIntersection BVH.cpp:19-110,
// Triangle::getIntersection Triangle.h:27-50, normalize Vector3.h:73-89, Ray Ray.h:11-12).
//
// One ray per lane.  A wave owns a 4 x 16 (beam x azimuth) tile of the range image so its 64 rays
// walk almost the same nodes; workgroups are handed out so that each XCD (private 4 MiB L2) owns one
// contiguous azimuth sector.  Traversal is "while-while": lanes descend internal nodes until every
// lane of the wave holds a leaf, then the leaves are intersected.  The per-ray stack of deferred
// children lives in LDS ([depth][lane] layout: bank = lane, conflict-free); entries beyond
// LT_STACK_LDS spill to HBM (never observed on real scenes, kept for adversarial inputs).
//
// Float arithmetic of the triangle test, the normalisation and the write-back is the reference's,
// operation by operation; the file is compiled with -ffp-contract=off and IEEE division / sqrt.
// The closest hit is the minimum over (t, face index), independent of traversal order.
#include "lt_internal.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "lt_normalize.h"  // the reference's RSQRTSS seed, replayed (Intel / AMD table) or exact

#define LT_DONE ((int)0x80000000)
#define LT_TILE_H 4
#define LT_TILE_W 16

template <bool COUNT>
__global__ __launch_bounds__(256) void k_trace(
    const float4* __restrict__ nodes, const float4* __restrict__ tris, const float* __restrict__ rays, float ox,
    float oy, float oz, int H, int W, int n_faces, const int* __restrict__ faces, const int* __restrict__ colors,
    const float* __restrict__ rem, float* __restrict__ endpoints, int* __restrict__ endcolors,
    float* __restrict__ range, float* __restrict__ endrem, int* __restrict__ tri_out, unsigned flags,
    int* __restrict__ overflow, unsigned long long* __restrict__ counters) {
  __shared__ int stack[4][LT_STACK_LDS][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long t_start = COUNT ? (unsigned long long)clock64() : 0ull;
  // XCD-aware tile assignment: workgroup b runs on XCD b % 8; give XCD x the x-th eighth of the
  // (column-major) tile list, i.e. one azimuth sector.  gridDim.x is a multiple of 8.
  const int nb = gridDim.x;
  const int lb = (blockIdx.x & 7) * (nb >> 3) + (blockIdx.x >> 3);
  const int tiles_h = (H + LT_TILE_H - 1) / LT_TILE_H, tiles_w = (W + LT_TILE_W - 1) / LT_TILE_W;
  const int wt = lb * 4 + wave;
  const int tile_x = wt / tiles_h, tile_y = wt - tile_x * tiles_h;
  const int h = tile_y * LT_TILE_H + (lane >> 4), w = tile_x * LT_TILE_W + (lane & 15);
  const bool active = tile_x < tiles_w && h < H && w < W;
  const size_t ray = (size_t)h * W + w;

  float dx = 0.f, dy = 0.f, dz = 1.f;
  if (active) {
    const float rx = rays[3 * ray], ry = rays[3 * ray + 1], rz = rays[3 * ray + 2];
    // normalize (Vector3.h:73-89): D = (x^2 + y^2) + z^2, r0 ~ 1/sqrt(D), one Newton-Raphson step
    const float D = (rx * rx + ry * ry) + rz * rz;
    const float r0 = lt_rsqrt_seed(D, flags);
    const float r = (1.5f * r0) + (((D * -0.5f) * r0) * (r0 * r0));
    dx = rx * r; dy = ry * r; dz = rz * r;
  }
  const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;  // Ray.h:12, +-inf allowed

  float best_t = 999999999.f;  // BVH.cpp:20
  int best_face = 0x7fffffff;
  int sp = 0;
  // NaN in the direction or the origin: never a hit (see k_trace4), no traversal
  const bool finite_ray = (dx == dx) && (dy == dy) && (dz == dz) && (ox == ox) && (oy == oy) && (oz == oz);
  int cur = (active && n_faces > 0 && finite_ray) ? 0 : LT_DONE;
  unsigned n_nodes = 0, n_tris = 0, n_ovf = 0;
  const float eps = 0.000001f;  // Triangle.h:32

  while (true) {
    // ---- descend ------------------------------------------------------------------------------
    while (cur >= 0) {
      const float4* N = nodes + 4 * (size_t)cur;
      const float4 q0 = N[0], q1 = N[1], q2 = N[2], q3 = N[3];
      if (COUNT) ++n_nodes;
      const float a0x = (q0.x - ox) * ix, b0x = (q0.w - ox) * ix;
      const float a0y = (q0.y - oy) * iy, b0y = (q1.x - oy) * iy;
      const float a0z = (q0.z - oz) * iz, b0z = (q1.y - oz) * iz;
      const float a1x = (q1.z - ox) * ix, b1x = (q2.y - ox) * ix;
      const float a1y = (q1.w - oy) * iy, b1y = (q2.z - oy) * iy;
      const float a1z = (q2.x - oz) * iz, b1z = (q2.w - oz) * iz;
      const float tn0 = fmaxf(fmaxf(fminf(a0x, b0x), fminf(a0y, b0y)), fmaxf(fminf(a0z, b0z), 0.0f));
      const float tf0 = fminf(fminf(fmaxf(a0x, b0x), fmaxf(a0y, b0y)), fminf(fmaxf(a0z, b0z), best_t));
      const float tn1 = fmaxf(fmaxf(fminf(a1x, b1x), fminf(a1y, b1y)), fmaxf(fminf(a1z, b1z), 0.0f));
      const float tf1 = fminf(fminf(fmaxf(a1x, b1x), fmaxf(a1y, b1y)), fminf(fmaxf(a1z, b1z), best_t));
      const bool h0 = tn0 <= tf0, h1 = tn1 <= tf1;
      const int c0 = __float_as_int(q3.x), c1 = __float_as_int(q3.y);
      if (h0 && h1) {
        const bool swap = tn1 < tn0;
        const int farc = swap ? c0 : c1;
        cur = swap ? c1 : c0;
        if (sp < LT_STACK_LDS) {
          stack[wave][sp][lane] = farc;
        } else {
          overflow[ray * (LT_STACK_MAX - LT_STACK_LDS) + (sp - LT_STACK_LDS)] = farc;
          if (COUNT) ++n_ovf;
        }
        ++sp;
      } else if (h0) {
        cur = c0;
      } else if (h1) {
        cur = c1;
      } else if (sp > 0) {
        --sp;
        cur = sp < LT_STACK_LDS ? stack[wave][sp][lane]
                                : overflow[ray * (LT_STACK_MAX - LT_STACK_LDS) + (sp - LT_STACK_LDS)];
      } else {
        cur = LT_DONE;
      }
    }
    if (cur == LT_DONE) break;
    // ---- leaf: Moller-Trumbore, operation order of Triangle.h:27-50 -------------------------------
    {
      const int ref = ~cur;
      const int start = ref & 0x0fffffff, cnt = (ref >> 28) + 1;
      for (int k = 0; k < cnt; ++k) {
        const float4* T = tris + 3 * (size_t)(start + k);
        const float4 t0 = T[0], t1 = T[1], t2 = T[2];
        if (COUNT) ++n_tris;
        const float e1x = t0.w, e1y = t1.x, e1z = t1.y, e2x = t1.z, e2y = t1.w, e2z = t2.x;
        const float hx = dy * e2z - dz * e2y, hy = dz * e2x - dx * e2z, hz = dx * e2y - dy * e2x;
        const float a = (e1x * hx + e1y * hy) + e1z * hz;
        if (a < eps && a > -eps) continue;
        const float inv_a = lt_rcp_ieee(a);  // = 1.0f / a, bit for bit (lt_internal.h)
        const float sx = ox - t0.x, sy = oy - t0.y, sz = oz - t0.z;
        const float u = ((sx * hx + sy * hy) + sz * hz) * inv_a;
        if (u < 0 || u > 1) continue;
        const float qx = sy * e1z - sz * e1y, qy = sz * e1x - sx * e1z, qz = sx * e1y - sy * e1x;
        const float v = ((dx * qx + dy * qy) + dz * qz) * inv_a;
        if (v < 0 || u + v > 1) continue;
        const float t = ((e2x * qx + e2y * qy) + e2z * qz) * inv_a;
        if (t < eps) continue;
        const int f = __float_as_int(t2.y);
        if (t < best_t || (t == best_t && f < best_face)) {
          best_t = t;
          best_face = f;
        }
      }
    }
    if (sp > 0) {
      --sp;
      cur = sp < LT_STACK_LDS ? stack[wave][sp][lane]
                              : overflow[ray * (LT_STACK_MAX - LT_STACK_LDS) + (sp - LT_STACK_LDS)];
    } else {
      cur = LT_DONE;
    }
  }

  // ---- write-back (RayTracer.cpp:73-90) ------------------------------------------------------------
  const bool hit = best_face != 0x7fffffff;
  if (active) {
    if (hit) {
      const int i0 = faces[3 * (size_t)best_face], i1 = faces[3 * (size_t)best_face + 1],
                i2 = faces[3 * (size_t)best_face + 2];
      if (endpoints) {  // hit = o + d * t (BVH.cpp:107)
        endpoints[3 * ray] = ox + dx * best_t;
        endpoints[3 * ray + 1] = oy + dy * best_t;
        endpoints[3 * ray + 2] = oz + dz * best_t;
      }
      if (endcolors) {  // colour of vertex 0, int -> float -> int (RayTracer.cpp:36, :80-82)
        if (flags & LT_TRACE_LABEL_IMAGE) {  // deform's unpack label_image = ray_colors[..., 2], laserscan.py:912
          endcolors[ray] = (int)(float)colors[3 * (size_t)i0 + 2];
        } else {
          endcolors[3 * ray] = (int)(float)colors[3 * (size_t)i0];
          endcolors[3 * ray + 1] = (int)(float)colors[3 * (size_t)i0 + 1];
          endcolors[3 * ray + 2] = (int)(float)colors[3 * (size_t)i0 + 2];
        }
      }
      if (endrem) endrem[ray] = ((rem[i0] + rem[i1]) + rem[i2]) / 3.0f;  // Triangle.h:69
      if (range) range[ray] = best_t;
      if (tri_out) tri_out[ray] = best_face;
    } else if (flags & LT_TRACE_WRITE_MISSES) {
      if (endpoints) { endpoints[3 * ray] = 0.f; endpoints[3 * ray + 1] = 0.f; endpoints[3 * ray + 2] = 0.f; }
      if (endcolors) {
        if (flags & LT_TRACE_LABEL_IMAGE) endcolors[ray] = 0;
        else { endcolors[3 * ray] = 0; endcolors[3 * ray + 1] = 0; endcolors[3 * ray + 2] = 0; }
      }
      if (endrem) endrem[ray] = 0.f;
      if (range) range[ray] = 0.f;
      if (tri_out) tri_out[ray] = -1;
    }
  }
  if (COUNT) {
    unsigned long long vn = n_nodes, vt = n_tris, vh = (active && hit) ? 1u : 0u, vo = n_ovf;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      vn += __shfl_xor(vn, o, 64);
      vt += __shfl_xor(vt, o, 64);
      vh += __shfl_xor(vh, o, 64);
      vo += __shfl_xor(vo, o, 64);
    }
    if (lane == 0) {
      atomicAdd(&counters[0], vn);
      atomicAdd(&counters[1], vt);
      atomicAdd(&counters[2], vh);
      atomicAdd(&counters[3], vo);
      const int wg = blockIdx.x * 4 + wave;  // debug: wave start / end clock (tools/wave_times.py)
      if (wg < LT_DBG_WAVES) {
        counters[8 + 2 * wg] = t_start;
        counters[8 + 2 * wg + 1] = (unsigned long long)clock64();
      }
    }
  }
}

// =====================================================================================================
// Quad traversal over the 4-wide nodes: FOUR LANES PER RAY.
//
// A 64x2048 scan is only 131 072 rays = 2 waves per SIMD with one ray per lane: far too few to hide the
// ~1 us dependent node fetch of a pointer-chasing traversal on 256 CUs.  Spreading a ray over a quad of
// lanes gives 8 waves per SIMD (full occupancy), halves the number of dependent steps (4-wide nodes), and
// turns the leaf loop into one step:
//   * node step: the quad reads one 128-B node as a single cache line, lane j tests child j; hits are
//     ranked by entry distance with DPP quad broadcasts; the nearest becomes the next node, the others
//     are pushed so that the nearer one is popped first;
//   * leaf step: lane j runs Moller-Trumbore on triangle j of the leaf (<= 4), then a quad min over
//     (t, face index).
// The per-ray stack is shared by the quad: LDS [depth][16 rays], spill to HBM beyond LT_STACK4_LDS.
//
// What bounds the kernel is the LONGEST walk, not the sum of the walks: a step is a dependent chain of ~0.9 us
// (node line from L2 / Infinity Cache + ~100 dependent VALU / DPP / LDS instructions), the mean ray needs 30 steps
// and the worst one 148 -- per-wave wall-clock stamps (tools/wave_times.py --quad) showed 99 % of the waves gone
// after 71 us of a 123 us launch, the chip 25 % busy over the span.  Two measures (C2: 125 -> 68 + 19 us):
//   * ONE loop whose trips are node OR leaf steps, so that a wave needs max-over-quads trips (the nested form made
//     a quad at a leaf wait for every other quad's run of node steps: 110 trips for the slowest wave although no
//     quad took more than 48 steps);
//   * hand-over: a ray that is not done after `step_cap` steps (LT_TRACE4_STEP_CAP = 40: 6 % of the rays on C2) parks
//     its state -- its stack plus the reference it was about to visit, best (t, face) -- in a queue, compacted
//     across the wave with one ballot-counted atomic per wave, and k_trace4_tail walks each such ray with a WHOLE
//     WAVE: the 16 quads pop 16 stack entries at a time and share best (t, face) through an LDS atomic, so the
//     remaining ~100 dependent steps of the worst ray become ~10.
// Results do not depend on the traversal order (min over (t, face), children are culled only when their entry
// distance exceeds the best t), so both kernels are bit-identical to k_trace and to the brute-force oracle.
// =====================================================================================================
#define LT_Q_BCAST(k) ((k) | ((k) << 2) | ((k) << 4) | ((k) << 6))
#define LT_Q_XOR1 0xB1  // quad_perm [1,0,3,2]
#define LT_Q_XOR2 0x4E  // quad_perm [2,3,0,1]

// (mov_dpp = update_dpp with an UNDEFINED old value: every source lane of a quad permutation exists, so nothing is kept
// from it -- with old = 0 the compiler set the destination to 0 before each of the ten permutations of a node step,
// 10 % of the vector instructions on the dependent chain of every node step)
template <int CTRL>
__device__ __forceinline__ int qperm_i(int v) {
  return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true);
}
template <int CTRL>
__device__ __forceinline__ float qperm_f(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}

// rays of a wave: a LT_TILE4_H x LT_TILE4_W tile of the image (16 rays).  k_trace4 on C2 (tools/tile_ab.sh): 2 x 8 60.3 us,
// 1 x 16 61.9, 4 x 4 57.0, 8 x 2 56.9 -- a beam step is 2.5 azimuth steps there, 4 x 4 is the most compact in angle
#ifndef LT_TILE4_H
#define LT_TILE4_H 4
#define LT_TILE4_W 4
#endif

// hit write-back of one ray (RayTracer.cpp:73-90); misses are written only with LT_TRACE_WRITE_MISSES
__device__ __forceinline__ void trace_writeback(size_t ray, float best_t, int best_face, float ox, float oy, float oz,
                                                float dx, float dy, float dz, const int* __restrict__ faces,
                                                const int* __restrict__ colors, const float* __restrict__ rem,
                                                float* __restrict__ endpoints, int* __restrict__ endcolors,
                                                float* __restrict__ range, float* __restrict__ endrem,
                                                int* __restrict__ tri_out, unsigned flags) {
  if (best_face != 0x7fffffff) {
    const int i0 = faces[3 * (size_t)best_face], i1 = faces[3 * (size_t)best_face + 1],
              i2 = faces[3 * (size_t)best_face + 2];
    if (endpoints) {
      endpoints[3 * ray] = ox + dx * best_t;
      endpoints[3 * ray + 1] = oy + dy * best_t;
      endpoints[3 * ray + 2] = oz + dz * best_t;
    }
    if (endcolors) {
      if (flags & LT_TRACE_LABEL_IMAGE) {  // deform's unpack label_image = ray_colors[..., 2], laserscan.py:912
        endcolors[ray] = (int)(float)colors[3 * (size_t)i0 + 2];
      } else {
        endcolors[3 * ray] = (int)(float)colors[3 * (size_t)i0];
        endcolors[3 * ray + 1] = (int)(float)colors[3 * (size_t)i0 + 1];
        endcolors[3 * ray + 2] = (int)(float)colors[3 * (size_t)i0 + 2];
      }
    }
    if (endrem) endrem[ray] = ((rem[i0] + rem[i1]) + rem[i2]) / 3.0f;
    if (range) range[ray] = best_t;
    if (tri_out) tri_out[ray] = best_face;
  } else if (flags & LT_TRACE_WRITE_MISSES) {
    if (endpoints) { endpoints[3 * ray] = 0.f; endpoints[3 * ray + 1] = 0.f; endpoints[3 * ray + 2] = 0.f; }
    if (endcolors) {
      if (flags & LT_TRACE_LABEL_IMAGE) endcolors[ray] = 0;
      else { endcolors[3 * ray] = 0; endcolors[3 * ray + 1] = 0; endcolors[3 * ray + 2] = 0; }
    }
    if (endrem) endrem[ray] = 0.f;
    if (range) range[ray] = 0.f;
    if (tri_out) tri_out[ray] = -1;
  }
}

// Moller-Trumbore with the reference's operation order (Triangle.h:27-50) on a sorted triangle record; returns t
// (INFINITY = no hit) and the face index
__device__ __forceinline__ float trace_tri(const float4* __restrict__ T, float ox, float oy, float oz, float dx, float dy,
                                           float dz, int& face) {
  const float eps = 0.000001f;
  const float4 t0 = T[0], t1 = T[1], t2 = T[2];
  const float e1x = t0.w, e1y = t1.x, e1z = t1.y, e2x = t1.z, e2y = t1.w, e2z = t2.x;
  const float hx = dy * e2z - dz * e2y, hy = dz * e2x - dx * e2z, hz = dx * e2y - dy * e2x;
  const float aa = (e1x * hx + e1y * hy) + e1z * hz;
  face = 0x7fffffff;
  if (aa < eps && aa > -eps) return INFINITY;
  const float inv_a = lt_rcp_ieee(aa);  // = 1.0f / aa, bit for bit (lt_internal.h)
  const float sx = ox - t0.x, sy = oy - t0.y, sz = oz - t0.z;
  const float u = ((sx * hx + sy * hy) + sz * hz) * inv_a;
  if (u < 0 || u > 1) return INFINITY;
  const float qx = sy * e1z - sz * e1y, qy = sz * e1x - sx * e1z, qz = sx * e1y - sy * e1x;
  const float v = ((dx * qx + dy * qy) + dz * qz) * inv_a;
  if (v < 0 || u + v > 1) return INFINITY;
  const float tt = ((e2x * qx + e2y * qy) + e2z * qz) * inv_a;
  if (tt < eps || !(tt == tt)) return INFINITY;  // NaN t is never recorded by the reference either (BVH.cpp:59)
  face = __float_as_int(t2.y);
  return tt;
}

// slab test of child j of a 4-wide node against the ray; hit iff the child exists and tn <= min(tfar, best_t)
__device__ __forceinline__ int trace_slab(const float4 a, const float4 b, float ox, float oy, float oz, float ix,
                                          float iy, float iz, float best_t, float& tn, int& ref) {
  const float l1x = (a.x - ox) * ix, l2x = (a.w - ox) * ix;
  const float l1y = (a.y - oy) * iy, l2y = (b.x - oy) * iy;
  const float l1z = (a.z - oz) * iz, l2z = (b.y - oz) * iz;
  tn = fmaxf(fmaxf(fminf(l1x, l2x), fminf(l1y, l2y)), fmaxf(fminf(l1z, l2z), 0.0f));
  const float tf = fminf(fminf(fmaxf(l1x, l2x), fmaxf(l1y, l2y)), fminf(fmaxf(l1z, l2z), best_t));
  ref = __float_as_int(b.z);
  // an unused child slot (box +inf, ref 0x7fffffff) fails the slab test of every finite ray, but fminf /
  // fmaxf drop NaN operands, so a ray or origin with a NaN would "hit" it: exclude it explicitly
  return (tn <= tf && ref != 0x7fffffff) ? 1 : 0;
}

// ray of cell (h, w): direction normalised like Vector3.h:73-89
__device__ __forceinline__ void trace_ray_dir(const float* __restrict__ rays, size_t ray, unsigned flags, float& dx,
                                              float& dy, float& dz) {
  const float rx = rays[3 * ray], ry = rays[3 * ray + 1], rz = rays[3 * ray + 2];
  const float D = (rx * rx + ry * ry) + rz * rz;
  const float r0 = lt_rsqrt_seed(D, flags);
  const float r = (1.5f * r0) + (((D * -0.5f) * r0) * (r0 * r0));
  dx = rx * r; dy = ry * r; dz = rz * r;
}

struct tail_args {  // hand-over area of a launch (lt_internal.h: lt_scene::tail_*)
  int* stack; float4* meta; int* queue;
  int* count;       // rays parked by this launch (starts at 0)
  int* count_next;  // the counter of the scene's NEXT launch: k_trace4_tail zeroes it (no memset between launches)
};

template <bool COUNT>
__global__ __launch_bounds__(256) void k_trace4(
    const float4* __restrict__ nodes4, const float4* __restrict__ tris, const float* __restrict__ rays, float ox,
    float oy, float oz, int H, int W, int n_faces, const int* __restrict__ faces, const int* __restrict__ colors,
    const float* __restrict__ rem, float* __restrict__ endpoints, int* __restrict__ endcolors,
    float* __restrict__ range, float* __restrict__ endrem, int* __restrict__ tri_out, unsigned flags,
    int* __restrict__ overflow, unsigned long long* __restrict__ counters, int step_cap, tail_args tail) {
  __shared__ int stack[4][LT_STACK4_LDS][16];
#ifdef LT_TRACE_CULL
  // A/B (LIDARHIP_EXTRA_FLAGS=-DLT_TRACE_CULL), measured and NOT the default: the entry distance of every deferred child, upper
  // 16 bits of the float (tn >= 0: truncation rounds DOWN) -- a popped entry whose box begins behind the best hit found since
  // it was pushed is dropped without fetching its node, BVH.cpp:41 ("if (near > intersection->t) continue").  C2: 27.2 -> 25.3
  // node visits per ray (7 % culled), and k_trace4 57 -> 73 us: the 16-bit LDS store per push and load per pop sit on the
  // dependent chain of EVERY step of a kernel that is bound by exactly that chain.
  __shared__ unsigned short stack_tn[4][LT_STACK4_LDS][16];
#endif
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 3, q = lane >> 2;
  const bool dbg_times = COUNT || (flags & LT_TRACE_DEBUG_TIMES);
  const unsigned long long t_start = dbg_times ? (unsigned long long)wall_clock64() : 0ull;  // 100 MHz
  const int nb = gridDim.x;
  const int lb = (blockIdx.x & 7) * (nb >> 3) + (blockIdx.x >> 3);  // XCD x owns the x-th azimuth sector
  const int tiles_h = (H + LT_TILE4_H - 1) / LT_TILE4_H, tiles_w = (W + LT_TILE4_W - 1) / LT_TILE4_W;
  const int wt = lb * 4 + wave;
  const int tile_x = wt / tiles_h, tile_y = wt - tile_x * tiles_h;
  const int h = tile_y * LT_TILE4_H + q / LT_TILE4_W, w = tile_x * LT_TILE4_W + q % LT_TILE4_W;
  const bool active = tile_x < tiles_w && h < H && w < W;
  const size_t ray = (size_t)h * W + w;

  float dx = 0.f, dy = 0.f, dz = 1.f;
  if (active) trace_ray_dir(rays, ray, flags, dx, dy, dz);
  const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;

  float best_t = 999999999.f;
  int best_face = 0x7fffffff;
  int sp = 0;
  // A direction or origin with a NaN can never be accepted by the triangle test (every product chain reaches
  // t as NaN, and BVH.cpp:59 keeps a hit only if t < best): such rays are misses without a traversal.
  const bool finite_ray = (dx == dx) && (dy == dy) && (dz == dz) && (ox == ox) && (oy == oy) && (oz == oz);
  int cur = (active && n_faces > 0 && finite_ray) ? 0 : LT_DONE;
  unsigned n_nodes = 0, n_tris = 0, n_ovf = 0;
  int n_steps = 0;          // node + leaf steps of this quad's ray
  bool handed_over = false;  // the ray goes on in k_trace4_tail
  int* spill = overflow + ray * (LT_STACK4_MAX - LT_STACK4_LDS);
  // entry s of this quad's stack: LDS below LT_STACK4_LDS, the spill area above.  (Written as a select of the two, the
  // compiler loads through a FLAT pointer chosen per lane: an LDS read always, the global one in a branch almost never taken.)
  auto pop = [&](int s) {
    typedef __attribute__((address_space(3))) int lds_int;  // (volatile LDS access: not to be merged with the other load)
    int v = *(volatile lds_int*)(lds_int*)&stack[wave][min(s, LT_STACK4_LDS - 1)][q];
    if (s >= LT_STACK4_LDS) v = spill[s - LT_STACK4_LDS];
    return v;
  };
  unsigned n_culled = 0;
  // next reference to visit: the top of the stack (with LT_TRACE_CULL: skipping entries that lie behind the best hit by now;
  // spilled entries carry no distance and are never culled).  sp, best_t are uniform within a quad.
  auto pop_next = [&]() {
    int r = LT_DONE;
    while (sp > 0) {
      --sp;
#ifdef LT_TRACE_CULL
      const float tnp = sp < LT_STACK4_LDS ? __uint_as_float((unsigned)stack_tn[wave][sp][q] << 16) : 0.f;
      if (tnp > best_t) {
        if (COUNT && j == 0) ++n_culled;
        continue;
      }
#endif
      r = pop(sp);
      break;
    }
    return r;
  };

  // ONE loop whose trips are steps of either kind ("if-if" traversal): a wave needs max-over-its-quads trips.  The
  // nested form (inner loop over nodes, leaf step outside) makes a quad that has reached a leaf wait for every other
  // quad's run of node steps and the other way round: measured 110 trips for the slowest wave although no quad took
  // more than 48 steps.  (cur, sp, best_*, n_steps are uniform within a quad.)
  while (cur != LT_DONE && n_steps < step_cap) {
    ++n_steps;
    if (cur >= 0) {
      // ---- node step: lane j tests child j ---------------------------------------------------------------
      const float4* N = nodes4 + 8 * (size_t)cur + 2 * j;
      const float4 a = N[0], b = N[1];
      if (COUNT && j == 0) ++n_nodes;
      float tn;
      int ref;
      const int hit = trace_slab(a, b, ox, oy, oz, ix, iy, iz, best_t, tn, ref);
      // rank of this child among the hit children of the quad: by (tn, lane) packed into one unsigned key -- tn >= 0, so
      // its bits order like its value; the two mantissa bits given to the lane only perturb the ORDER of near-equal
      // children, which the result does not depend on; a child that is not hit has the largest key and is never counted
      const unsigned key = hit ? ((__float_as_uint(tn) & ~3u) | (unsigned)j) : 0xFFFFFFFFu;
      const unsigned k0 = (unsigned)qperm_i<LT_Q_BCAST(0)>((int)key), k1 = (unsigned)qperm_i<LT_Q_BCAST(1)>((int)key);
      const unsigned k2 = (unsigned)qperm_i<LT_Q_BCAST(2)>((int)key), k3 = (unsigned)qperm_i<LT_Q_BCAST(3)>((int)key);
      const int rank = (int)(k0 < key) + (int)(k1 < key) + (int)(k2 < key) + (int)(k3 < key);
      const unsigned long long hm = __ballot(hit != 0);
      const int nh = __popc((unsigned)(hm >> (lane & ~3)) & 15u);
      if (nh == 0) {
        cur = pop_next();
      } else {
        // nearest child -> next node (OR-reduce the single rank-0 reference over the quad)
        int nxt = (hit && rank == 0) ? ref : 0;
        nxt |= qperm_i<LT_Q_XOR1>(nxt);
        nxt |= qperm_i<LT_Q_XOR2>(nxt);
        if (hit && rank > 0) {  // rank 1 ends on top of the stack
          const int slot = sp + (nh - 1 - rank);
          if (slot < LT_STACK4_LDS) {
            stack[wave][slot][q] = ref;
#ifdef LT_TRACE_CULL
            stack_tn[wave][slot][q] = (unsigned short)(__float_as_uint(tn) >> 16);
#endif
          } else {
            spill[slot - LT_STACK4_LDS] = ref;
            if (COUNT) ++n_ovf;
          }
        }
        sp += nh - 1;
        cur = nxt;
      }
    } else {
      // ---- leaf step: one triangle per lane (Triangle.h:27-50 operation order) ----------------------------------
      const int lref = ~cur;
      const int start = lref & 0x0fffffff, cnt = (lref >> 28) + 1;
      float t = INFINITY;
      int f = 0x7fffffff;
      if (j < cnt) {
        t = trace_tri(tris + 3 * (size_t)(start + j), ox, oy, oz, dx, dy, dz, f);
        if (COUNT) ++n_tris;
      }
      // quad min over (t, face)
      {
        const float ot = qperm_f<LT_Q_XOR1>(t);
        const int of = qperm_i<LT_Q_XOR1>(f);
        if (ot < t || (ot == t && of < f)) { t = ot; f = of; }
      }
      {
        const float ot = qperm_f<LT_Q_XOR2>(t);
        const int of = qperm_i<LT_Q_XOR2>(f);
        if (ot < t || (ot == t && of < f)) { t = ot; f = of; }
      }
      if (t < best_t || (t == best_t && f < best_face)) {
        best_t = t;
        best_face = f;
      }
      cur = pop_next();
    }
  }
  handed_over = cur != LT_DONE;  // left by the step cap: cur is the node / leaf reference still to be visited

  // ---- hand-over of the rays that hit the step cap: compaction across the wave (one atomic per wave) ---------
  {
    const unsigned long long m = __ballot(handed_over && j == 0);
    if (m) {  // wave-uniform
      int base = 0;
      if (lane == 0) base = atomicAdd(tail.count, __popcll(m));
      base = __builtin_amdgcn_readfirstlane(base);
      if (handed_over) {
        const int slot = base + __popcll(m & ((1ull << (lane & ~3)) - 1ull));  // rank of this quad among the parked ones
        int* dst = tail.stack + (size_t)slot * LT_TAIL_SAVE;
        for (int i = j; i < sp; i += 4) dst[i] = i < LT_STACK4_LDS ? stack[wave][i][q] : spill[i - LT_STACK4_LDS];
        if (j == 0) {
          dst[sp] = cur;  // the reference it was about to visit goes on top
          tail.meta[slot] = make_float4(best_t, __int_as_float(best_face), __int_as_float(sp + 1), 0.f);
          tail.queue[slot] = (int)ray;
        }
      }
    }
  }

  // ---- write-back by lane 0 of the quad (RayTracer.cpp:73-90) ----------------------------------------------
  const bool hit = best_face != 0x7fffffff;
  if ((flags & LT_TRACE_DEBUG_STEPS) && tri_out) {  // debug: steps per ray instead of the face (cap off)
    if (active && j == 0) tri_out[ray] = n_steps;
  } else if (active && j == 0 && !handed_over) {
    trace_writeback(ray, best_t, best_face, ox, oy, oz, dx, dy, dz, faces, colors, rem, endpoints, endcolors, range,
                    endrem, tri_out, flags);
  }
  if (COUNT) {
    unsigned long long vn = n_nodes, vt = n_tris, vh = (active && hit && j == 0) ? 1u : 0u, vo = n_ovf, vc = n_culled;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      vn += __shfl_xor(vn, o, 64);
      vt += __shfl_xor(vt, o, 64);
      vh += __shfl_xor(vh, o, 64);
      vo += __shfl_xor(vo, o, 64);
      vc += __shfl_xor(vc, o, 64);
    }
    if (lane == 0) {
      atomicAdd(&counters[0], vn);
      atomicAdd(&counters[1], vt);
      atomicAdd(&counters[2], vh);
      atomicAdd(&counters[3], vo);
      atomicAdd(&counters[4], vc);  // stack entries dropped at pop (behind the best hit)
    }
  }
  if (dbg_times && lane == 0) {  // debug (tools/wave_times.py --quad): start / duration at 100 MHz, steps of quad 0
    const int wg = blockIdx.x * 4 + wave;
    if (wg < LT_DBG_WAVES) {
      counters[8 + 2 * wg] = t_start;
      counters[8 + 2 * wg + 1] = ((unsigned long long)wall_clock64() - t_start) | ((unsigned long long)n_steps << 32);
    }
  }
}

// One WAVE per handed-over ray: the 16 quads pop up to 16 entries of the ray's stack at a time (LT_TAIL_DFS: one at a
// time while the stack is very full), every quad visits its node (lane j tests child j against the shared best t) or
// its leaf (lane j tests triangle j; quad minimum into the shared best (t, face) with one 64-bit LDS atomic), the hit
// children of all quads are pushed with a wave prefix sum.  Workgroup = one wave, so __syncthreads() is a wave barrier.
__global__ __launch_bounds__(64) void k_trace4_tail(
    const float4* __restrict__ nodes4, const float4* __restrict__ tris, const float* __restrict__ rays, float ox,
    float oy, float oz, const int* __restrict__ faces, const int* __restrict__ colors, const float* __restrict__ rem,
    float* __restrict__ endpoints, int* __restrict__ endcolors, float* __restrict__ range, float* __restrict__ endrem,
    int* __restrict__ tri_out, unsigned flags, tail_args tail) {
  __shared__ int stk[LT_TAIL_STACK];
  __shared__ unsigned long long best;  // (t bits) << 32 | face: for t > 0 the integer order is the order of (t, face)
  const int lane = threadIdx.x, j = lane & 3, q = lane >> 2;
  const int n_tail = *tail.count;
  if (blockIdx.x == 0 && lane == 0) *tail.count_next = 0;
  for (int it = blockIdx.x; it < n_tail; it += gridDim.x) {
    const size_t ray = (size_t)tail.queue[it];
    const float4 meta = tail.meta[it];
    int sp = __float_as_int(meta.z);
    const int* src = tail.stack + (size_t)it * LT_TAIL_SAVE;
    for (int i = lane; i < sp; i += 64) stk[i] = src[i];
    if (lane == 0) best = ((unsigned long long)__float_as_uint(meta.x) << 32) | (unsigned)__float_as_int(meta.y);
    float dx, dy, dz;
    trace_ray_dir(rays, ray, flags, dx, dy, dz);
    const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;
    __syncthreads();
    while (sp > 0) {  // sp is wave-uniform
      const int take = sp > LT_TAIL_DFS ? 1 : min(sp, 16);
      const int ref_in = q < take ? stk[sp - 1 - q] : LT_DONE;  // quad 0 takes the top = the nearest deferred child
      const float best_t = __uint_as_float((unsigned)(best >> 32));
      __syncthreads();  // every quad has read its entry and the best t before anything is pushed / improved
      sp -= take;
      int hit = 0, ref = 0;
      float tn = 0.f;
      if (ref_in >= 0) {  // node: lane j tests child j
        const float4* N = nodes4 + 8 * (size_t)ref_in + 2 * j;
        hit = trace_slab(N[0], N[1], ox, oy, oz, ix, iy, iz, best_t, tn, ref);
      } else if (ref_in != LT_DONE) {  // leaf: lane j tests triangle j
        const int lref = ~ref_in;
        const int start = lref & 0x0fffffff, cnt = (lref >> 28) + 1;
        float t = INFINITY;
        int f = 0x7fffffff;
        if (j < cnt) t = trace_tri(tris + 3 * (size_t)(start + j), ox, oy, oz, dx, dy, dz, f);
        {
          const float ot = qperm_f<LT_Q_XOR1>(t);
          const int of = qperm_i<LT_Q_XOR1>(f);
          if (ot < t || (ot == t && of < f)) { t = ot; f = of; }
        }
        {
          const float ot = qperm_f<LT_Q_XOR2>(t);
          const int of = qperm_i<LT_Q_XOR2>(f);
          if (ot < t || (ot == t && of < f)) { t = ot; f = of; }
        }
        if (j == 0 && f != 0x7fffffff && t < 999999999.f)
          atomicMin(&best, ((unsigned long long)__float_as_uint(t) << 32) | (unsigned)f);
      }
      // push the hit children of all quads: far ones first inside a quad (the nearer child is popped first); the
      // quads are stacked in reverse order so that quad 0's children -- the subtree of the old top -- end on top
      // rank inside the quad by the packed (tn, lane) key, as in k_trace4
      const unsigned key = hit ? ((__float_as_uint(tn) & ~3u) | (unsigned)j) : 0xFFFFFFFFu;
      const unsigned k0 = (unsigned)qperm_i<LT_Q_BCAST(0)>((int)key), k1 = (unsigned)qperm_i<LT_Q_BCAST(1)>((int)key);
      const unsigned k2 = (unsigned)qperm_i<LT_Q_BCAST(2)>((int)key), k3 = (unsigned)qperm_i<LT_Q_BCAST(3)>((int)key);
      const int rank = (int)(k0 < key) + (int)(k1 < key) + (int)(k2 < key) + (int)(k3 < key);
      // exclusive prefix of nh over the quads ABOVE this one (quads q+1 .. 15), via the hit ballot
      const unsigned long long hm = __ballot(hit != 0);
      const int nh = __popc((unsigned)(hm >> (lane & ~3)) & 15u);
      const int above = q == 15 ? 0 : __popcll(hm >> ((q + 1) * 4));
      const int total = __popcll(hm);
      if (hit) stk[sp + above + (nh - 1 - rank)] = ref;
      sp += total;
      __syncthreads();
    }
    if (lane == 0) {
      const unsigned long long k = best;
      trace_writeback(ray, __uint_as_float((unsigned)(k >> 32)), (int)(unsigned)(k & 0xFFFFFFFFull), ox, oy, oz, dx, dy,
                      dz, faces, colors, rem, endpoints, endcolors, range, endrem, tri_out, flags);
    }
    __syncthreads();
  }
}

bool lt_binary_path() {
  static const bool v = []() {
    const char* e = getenv("LIDARHIP_TRACE");
    return e && strcmp(e, "binary") == 0;
  }();
  return v;
}

int lt_trace_launch(lt_scene* s, const float* rays, const float* origin, int n_rays, int height, float* endpoints,
                    int* endcolors, float* range, float* endrem, int* tri, unsigned flags, hipStream_t stream,
                    lt_stats* stats) {
  if (!s->built) {
    lt_set_error("lt_scene_trace_dev: scene has no BVH (call lt_scene_build first)");
    return LT_ERR_NOT_BUILT;
  }
  if (height <= 0 || n_rays < 0 || !origin || (n_rays > 0 && !rays)) {
    lt_set_error("lt_scene_trace_dev: invalid argument (height=%d n_rays=%d)", height, n_rays);
    return LT_ERR_INVALID_ARG;
  }
  s->last_stream = stream;
  const int W = n_rays / height;  // RayTracer.cpp:56
  const int H = height;
  const bool timed = stats != nullptr;
  const bool count = (flags & LT_TRACE_COUNT) != 0;
  s->stats.n_rays = W * H;
  const bool binary_path_used = lt_binary_path();
  if (W > 0) {
    LT_CHECK(lt_scene_reserve_rays(s, W * H));
    if (count) LT_HIP(hipMemsetAsync(s->counters, 0, 5 * sizeof(unsigned long long), stream));
    const bool binary_path = lt_binary_path();
    if (timed) LT_HIP(hipEventRecord(s->ev[7], stream));
    if (binary_path) {
      const int tiles = ((H + LT_TILE_H - 1) / LT_TILE_H) * ((W + LT_TILE_W - 1) / LT_TILE_W);
      int nblocks = (tiles + 3) / 4;
      nblocks = (nblocks + 7) & ~7;
      if (count)
        hipLaunchKernelGGL(k_trace<true>, dim3(nblocks), dim3(256), 0, stream, s->nodes, s->tris, rays, origin[0],
                           origin[1], origin[2], H, W, s->n_faces, s->faces, s->colors, s->rem, endpoints,
                           endcolors, range, endrem, tri, flags, s->overflow, s->counters);
      else
        hipLaunchKernelGGL(k_trace<false>, dim3(nblocks), dim3(256), 0, stream, s->nodes, s->tris, rays, origin[0],
                           origin[1], origin[2], H, W, s->n_faces, s->faces, s->colors, s->rem, endpoints,
                           endcolors, range, endrem, tri, flags, s->overflow, s->counters);
    } else {
      const int tiles = ((H + LT_TILE4_H - 1) / LT_TILE4_H) * ((W + LT_TILE4_W - 1) / LT_TILE4_W);
      int nblocks = (tiles + 3) / 4;
      nblocks = (nblocks + 7) & ~7;
      // two hand-over counters used alternately: the tail kernel of a launch re-arms the other one
      const bool with_tail = !count && !(flags & LT_TRACE_DEBUG_STEPS);
      const tail_args tail = {s->tail_stack, s->tail_meta, s->tail_queue, s->tail_count + (s->tail_parity & 1),
                              s->tail_count + ((s->tail_parity & 1) ^ 1)};
      if (with_tail) s->tail_parity ^= 1;
      static const int env_cap = []() {
        const char* e = getenv("LIDARHIP_STEP_CAP");
        return e ? atoi(e) : LT_TRACE4_STEP_CAP;
      }();
      // counting and the per-ray step image measure the undisturbed walk: no hand-over there
      const int step_cap = (!with_tail || env_cap <= 0) ? 0x7fffffff : env_cap;
      if (count) {
        hipLaunchKernelGGL(k_trace4<true>, dim3(nblocks), dim3(256), 0, stream, s->nodes4, s->tris, rays, origin[0],
                           origin[1], origin[2], H, W, s->n_faces, s->faces, s->colors, s->rem, endpoints,
                           endcolors, range, endrem, tri, flags, s->overflow, s->counters, step_cap, tail);
      } else {
        if (s->probe[0]) LT_HIP(hipEventRecord(s->probe[0], stream));
        hipLaunchKernelGGL(k_trace4<false>, dim3(nblocks), dim3(256), 0, stream, s->nodes4, s->tris, rays,
                           origin[0], origin[1], origin[2], H, W, s->n_faces, s->faces, s->colors, s->rem,
                           endpoints, endcolors, range, endrem, tri, flags, s->overflow, s->counters, step_cap, tail);
        // (launched even when the cap is off: it re-arms the counter and finds an empty queue)
        hipLaunchKernelGGL(k_trace4_tail, dim3(4096), dim3(64), 0, stream, s->nodes4, s->tris, rays, origin[0],
                             origin[1], origin[2], s->faces, s->colors, s->rem, endpoints, endcolors, range, endrem,
                             tri, flags, tail);
        if (s->probe[1]) LT_HIP(hipEventRecord(s->probe[1], stream));
        s->probe[0] = s->probe[1] = nullptr;
      }
    }
    if (timed) LT_HIP(hipEventRecord(s->ev[8], stream));
    LT_HIP(hipGetLastError());
  }
  if (timed || count) {
    LT_HIP(hipStreamSynchronize(stream));
    if (timed && W > 0) LT_HIP(hipEventElapsedTime(&s->stats.ms_trace, s->ev[7], s->ev[8]));
    if (count && W > 0) {
      unsigned long long c[5];
      LT_HIP(hipMemcpy(c, s->counters, sizeof(c), hipMemcpyDeviceToHost));
      s->stats.nodes_visited = c[0];
      s->stats.tris_tested = c[1];
      s->stats.n_hits = (int)c[2];
      s->stats.stack_overflows = c[3];
      s->stats.entries_culled = binary_path_used ? 0 : c[4];
    }
    if (stats) *stats = s->stats;
  }
  return LT_OK;
}
