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
#define LT_TABLE_ATTR __device__
#include "lt_rsqrt_sse_table.h"

// x86 RSQRTSS replayed from the measured 2 x 1024 table (generator + exhaustive proof:
// oracle/gen_rsqrt_table.c).  The reference seeds its normalize() with _mm_rsqrt_ps
// (Vector3.h:83); replaying the seed makes the ray directions -- and with them every output
// bit -- those the reference produces on the CPU family the table was measured on.
__device__ __forceinline__ float rsqrt_sse(float x) {
  const unsigned b = __float_as_uint(x);
  const int e = (int)((b >> 23) & 255u);
  if (e == 0) return INFINITY;                       // zero / denormal source is treated as zero
  if (e == 255) return (b & 0x7fffffu) ? x : 0.0f;   // NaN -> NaN, +inf -> 0
  const int p = (e - 127) & 1;
  const int k = (e - 127 - p) / 2;
  return __uint_as_float(LT_RSQRT_SSE_TABLE[p * 1024 + ((b >> 13) & 1023u)] - ((unsigned)k << 23));
}

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
    const float r0 = (flags & LT_TRACE_NORM_EXACT) ? 1.0f / sqrtf(D) : rsqrt_sse(D);
    const float r = (1.5f * r0) + (((D * -0.5f) * r0) * (r0 * r0));
    dx = rx * r; dy = ry * r; dz = rz * r;
  }
  const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;  // Ray.h:12, +-inf allowed

  float best_t = 999999999.f;  // BVH.cpp:20
  int best_face = 0x7fffffff;
  int sp = 0;
  int cur = (active && n_faces > 0) ? 0 : LT_DONE;
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
        const float inv_a = 1.0f / a;
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
        endcolors[3 * ray] = (int)(float)colors[3 * (size_t)i0];
        endcolors[3 * ray + 1] = (int)(float)colors[3 * (size_t)i0 + 1];
        endcolors[3 * ray + 2] = (int)(float)colors[3 * (size_t)i0 + 2];
      }
      if (endrem) endrem[ray] = ((rem[i0] + rem[i1]) + rem[i2]) / 3.0f;  // Triangle.h:69
      if (range) range[ray] = best_t;
      if (tri_out) tri_out[ray] = best_face;
    } else if (flags & LT_TRACE_WRITE_MISSES) {
      if (endpoints) { endpoints[3 * ray] = 0.f; endpoints[3 * ray + 1] = 0.f; endpoints[3 * ray + 2] = 0.f; }
      if (endcolors) { endcolors[3 * ray] = 0; endcolors[3 * ray + 1] = 0; endcolors[3 * ray + 2] = 0; }
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
    }
  }
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
  if (W > 0) {
    LT_CHECK(lt_scene_reserve_rays(s, W * H));
    if (count) LT_HIP(hipMemsetAsync(s->counters, 0, 4 * sizeof(unsigned long long), stream));
    const int tiles = ((H + LT_TILE_H - 1) / LT_TILE_H) * ((W + LT_TILE_W - 1) / LT_TILE_W);
    int nblocks = (tiles + 3) / 4;
    nblocks = (nblocks + 7) & ~7;
    if (timed) LT_HIP(hipEventRecord(s->ev[7], stream));
    if (count)
      hipLaunchKernelGGL(k_trace<true>, dim3(nblocks), dim3(256), 0, stream, s->nodes, s->tris, rays, origin[0],
                         origin[1], origin[2], H, W, s->n_faces, s->faces, s->colors, s->rem, endpoints, endcolors,
                         range, endrem, tri, flags, s->overflow, s->counters);
    else
      hipLaunchKernelGGL(k_trace<false>, dim3(nblocks), dim3(256), 0, stream, s->nodes, s->tris, rays, origin[0],
                         origin[1], origin[2], H, W, s->n_faces, s->faces, s->colors, s->rem, endpoints, endcolors,
                         range, endrem, tri, flags, s->overflow, s->counters);
    if (timed) LT_HIP(hipEventRecord(s->ev[8], stream));
    LT_HIP(hipGetLastError());
  }
  if (timed || count) {
    LT_HIP(hipStreamSynchronize(stream));
    if (timed && W > 0) LT_HIP(hipEventElapsedTime(&s->stats.ms_trace, s->ev[7], s->ev[8]));
    if (count && W > 0) {
      unsigned long long c[4];
      LT_HIP(hipMemcpy(c, s->counters, sizeof(c), hipMemcpyDeviceToHost));
      s->stats.nodes_visited = c[0];
      s->stats.tris_tested = c[1];
      s->stats.n_hits = (int)c[2];
      s->stats.stack_overflows = c[3];
    }
    if (stats) *stats = s->stats;
  }
  return LT_OK;
}
