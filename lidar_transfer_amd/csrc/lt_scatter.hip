// lt_scatter.hip -- single-origin fast path: stream the TRIANGLES, scatter hits into the range image.
//
// Every call of the reference's `ctrace` casts all rays from ONE origin (RayTracer.cpp:58, :68).  For
// that case the closest hit per ray can be computed without any hierarchy over the mesh: bin the rays
// once by direction (azimuth x elevation, rays at the bin centres), then stream the triangles -- each
// triangle computes conservative angular bounds as seen from the origin, visits the bins whose ray can lie
// inside them (for a sensor grid: the rays inside the bounds, none for most sub-pixel triangles), runs the
// reference's Moller-Trumbore test (Triangle.h:27-50, identical float operations to lt_trace.hip) against
// those rays only, and merges accepted hits with a 64-bit atomicMin on (t bits << 32 | face).
// The result is, by construction, the minimum over (t, face index) of all accepted triangles -- the same
// tree-independent definition the LBVH path implements, hence bit-identical images -- but the work is one
// coalesced pass over the mesh with no sort, no tree and no dependent pointer chase.  What bounds it (the latency
// chain of a workgroup at full occupancy, at ~45 % of the HBM roof and ~45 % of the VALU roof) and which
// restructurings were measured and rejected: DESIGN.md section 5d.
//
//   rayset (built once per sensor model, read-only afterwards: shared by every scene and stream):
//     k_rs_dirs    normalise (Vector3.h:73-89), azimuth/elevation, elevation range partials
//     k_rs_fit     fit the azimuth grid (W or W - 1 columns) to the rays
//     k_rs_keys    bin id per ray  ->  radix sort (k_hist/k_scan/k_scatter of lt_build.hip)
//     k_rs_starts  first sorted slot of every bin
//     k_rs_sortdirs / k_rs_grid  directions in bin order; one 16-B entry per bin (its ray, empty, or a slot range)
//   per scan:
//     k_sc_tris    256 triangles per workgroup: bounds, candidate bins, balanced MT rounds, atomicMin;
//                  the excess of heavy workgroups and big triangles -> queues
//     k_sc_rest    queued slices of heavy workgroups; up to 8 waves per queued big triangle
//     k_sc_resolve one thread per ray: unpack (t, face), write-back (RayTracer.cpp:73-90), reset the cell
//   (the z-min cells and the two queues are the state of a render in flight: they belong to the scene)
#include "lt_internal.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "lt_normalize.h"

#define LT_PI_F 3.14159265358979f
#define LT_BIN_SLACK 4e-3f  // bins; float rounding of a grid coordinate is < 8192 * 2^-22
#define LT_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull

// Bin grid of a rayset.  A ray with azimuth phi and elevation th has the continuous grid coordinates
//   x = (phi + pi) * az_scale - az_off   (nb_az columns, periodic),   y = (th - el_lo) * el_scale   (nb_el rows)
// and lives in the bin (round(x) mod nb_az, clamp(round(y))): the grid is laid out so that the rays of a
// regular sensor model sit at the bin CENTRES (integer x, y).  dev_az / dev_el = the largest distance of any
// ray from the centre of its bin, measured when the rayset is built.  A triangle whose padded angular bounds
// are [x_lo, x_hi] can only be hit by rays of the columns ceil(x_lo - dev) .. floor(x_hi + dev): for a sensor
// grid (dev ~ 1e-3) that is "the rays inside the bounds" -- none at all for most sub-pixel triangles --
// instead of "every bin the bounds touch"; for an irregular ray set (dev ~ 0.5) it degrades to the latter.
// The azimuth grid is fitted to the rays (k_rs_fit): W columns over [-pi, pi) or W - 1 columns when the model
// spans [-pi, pi] inclusively (the reference's create_rays, laserscan.py, does: first and last column
// coincide), with the phase of ray 0.
struct rs_params {
  int nb_az, nb_el;
  float az_scale;        // nb_az / 2pi
  float az_off;          // phase: ray 0 sits at an integer x
  float el_lo, el_scale;
  float dev_az, dev_el;  // written with atomicMax on the float bits (non-negative)
  float dev_fit[2];      // k_rs_fit: dev_az the grid would have with W / W - 1 columns
  // derived on the host once the deviations are known (rs_derive): tri_bins' bin coordinates as ONE fma each,
  //   row bounds    ceil(th * el_scale + el_c0), floor(th * el_scale + el_c1)
  //   column bounds ceil(a * az_scale + az_c0),  floor(a * az_scale + az_c1)
  float el_c0, el_c1, az_c0, az_c1;
};

__device__ __forceinline__ float rs_az_off(float phi0, float az_scale) {
  if (!(phi0 == phi0)) return 0.f;
  const float x0 = (phi0 + LT_PI_F) * az_scale;
  return x0 - floorf(x0 + 0.5f);
}

// ---- rayset -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rs_dirs(const float* __restrict__ rays, int n, unsigned flags,
                                                 float4* __restrict__ dirs, float2* __restrict__ ang,
                                                 float* __restrict__ partial) {
  __shared__ float red[4][2];
  float lo = INFINITY, hi = -INFINITY;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float rx = rays[3 * (size_t)i], ry = rays[3 * (size_t)i + 1], rz = rays[3 * (size_t)i + 2];
    const float D = (rx * rx + ry * ry) + rz * rz;
    const float r0 = lt_rsqrt_seed(D, flags);
    const float r = (1.5f * r0) + (((D * -0.5f) * r0) * (r0 * r0));
    const float dx = rx * r, dy = ry * r, dz = rz * r;
    dirs[i] = make_float4(dx, dy, dz, 0.f);
    const float phi = atan2f(dy, dx), th = atan2f(dz, sqrtf(dx * dx + dy * dy));
    ang[i] = make_float2(phi, th);
    if (th == th) { lo = fminf(lo, th); hi = fmaxf(hi, th); }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, o, 64));
    hi = fmaxf(hi, __shfl_xor(hi, o, 64));
  }
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = lo; red[threadIdx.x >> 6][1] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = fminf(fminf(red[0][0], red[1][0]), fminf(red[2][0], red[3][0]));
    partial[2 * blockIdx.x + 1] = fmaxf(fmaxf(red[0][1], red[1][1]), fmaxf(red[2][1], red[3][1]));
  }
}

// distance of the rays from the column centres for the two candidate azimuth grids
__global__ __launch_bounds__(256) void k_rs_fit(const float2* __restrict__ ang, int n, int nb_az0,
                                                rs_params* __restrict__ prm) {
  const float phi0 = n > 0 ? ang[0].x : 0.f;
  float dev[2] = {0.f, 0.f};
  for (int k = 0; k < 2; ++k) {
    const int K = nb_az0 - k;
    if (K < 1) continue;
    const float sc = (float)K / (2.0f * LT_PI_F), off = rs_az_off(phi0, sc);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
      const float phi = ang[i].x;
      if (phi == phi) {
        const float x = (phi + LT_PI_F) * sc - off;
        dev[k] = fmaxf(dev[k], fabsf(x - floorf(x + 0.5f)));
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    dev[0] = fmaxf(dev[0], __shfl_xor(dev[0], o, 64));
    dev[1] = fmaxf(dev[1], __shfl_xor(dev[1], o, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMax((int*)&prm->dev_fit[0], __float_as_int(dev[0]));
    atomicMax((int*)&prm->dev_fit[1], __float_as_int(dev[1]));
  }
}

// reduce the elevation partials (every workgroup redoes it: 2 KB), publish the grid, bin id per ray
__global__ __launch_bounds__(256) void k_rs_keys(const float2* __restrict__ ang, int n, int nb_az, int nb_el,
                                                 const float* __restrict__ partial, rs_params* __restrict__ prm,
                                                 uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  __shared__ float red[4][2];
  float lo = partial[2 * threadIdx.x], hi = partial[2 * threadIdx.x + 1];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, o, 64));
    hi = fmaxf(hi, __shfl_xor(hi, o, 64));
  }
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = lo; red[threadIdx.x >> 6][1] = hi; }
  __syncthreads();
  lo = fminf(fminf(red[0][0], red[1][0]), fminf(red[2][0], red[3][0]));
  hi = fmaxf(fmaxf(red[0][1], red[1][1]), fmaxf(red[2][1], red[3][1]));
  if (!(lo <= hi)) { lo = 0.f; hi = 0.f; }
  // W - 1 columns when that grid clearly fits the rays better (inclusive [-pi, pi] models)
  const int nb_max = nb_az;
  if (nb_az >= 5 && prm->dev_fit[1] + 0.01f < prm->dev_fit[0]) nb_az -= 1;
  rs_params p;
  p.nb_az = nb_az; p.nb_el = nb_el;
  p.az_scale = (float)nb_az / (2.0f * LT_PI_F);
  p.az_off = rs_az_off(n > 0 ? ang[0].x : 0.f, p.az_scale);
  p.el_lo = lo;
  p.el_scale = (nb_el > 1 && hi > lo) ? (float)(nb_el - 1) / (hi - lo) : 0.f;
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // dev_az / dev_el were zeroed by the host before this launch
    prm->nb_az = p.nb_az; prm->nb_el = p.nb_el;
    prm->az_scale = p.az_scale; prm->az_off = p.az_off; prm->el_lo = p.el_lo; prm->el_scale = p.el_scale;
  }
  const int i = blockIdx.x * 256 + threadIdx.x;
  float dev_a = 0.f, dev_e = 0.f;
  if (i < n) {
    const float2 a = ang[i];
    uint32_t key = (uint32_t)nb_max * (uint32_t)nb_el;  // NaN directions: one extra bin that no triangle visits
    if (a.x == a.x && a.y == a.y) {
      const float x = (a.x + LT_PI_F) * p.az_scale - p.az_off, y = (a.y - p.el_lo) * p.el_scale;
      const float cx = floorf(x + 0.5f);
      const float cy = fminf(fmaxf(floorf(y + 0.5f), 0.f), (float)(nb_el - 1));
      dev_a = fabsf(x - cx);
      dev_e = fabsf(y - cy);
      int ia = (int)cx;  // x in [-1, nb_az + 1): the columns are periodic
      if (ia < 0) ia += nb_az;
      else if (ia >= nb_az) ia -= nb_az;
      ia = min(max(ia, 0), nb_az - 1);
      key = (uint32_t)cy * (uint32_t)nb_az + (uint32_t)ia;
    }
    keys[i] = key;
    vals[i] = (uint32_t)i;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    dev_a = fmaxf(dev_a, __shfl_xor(dev_a, o, 64));
    dev_e = fmaxf(dev_e, __shfl_xor(dev_e, o, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMax((int*)&prm->dev_az, __float_as_int(dev_a));
    atomicMax((int*)&prm->dev_el, __float_as_int(dev_e));
  }
}

// bin_start[b] = first sorted slot whose key >= b, for b in [0, nbins]; keys are sorted
__global__ __launch_bounds__(256) void k_rs_starts(const uint32_t* __restrict__ keys, int n, int nbins,
                                                   int* __restrict__ bin_start) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p > n) return;
  const int lo = p == 0 ? 0 : (int)min(keys[p - 1], (uint32_t)nbins) + 1;
  const int hi = p == n ? nbins : (int)min(keys[p], (uint32_t)nbins);
  for (int b = lo; b <= hi; ++b) bin_start[b] = p;
}

// ---- triangle scatter ------------------------------------------------------------------------------------
struct tri_rec {
  float v0x, v0y, v0z, e1x, e1y, e1z, e2x, e2y, e2z;
};

// Moller-Trumbore with the reference's operation order (Triangle.h:27-50); returns t or NaN for "no hit"
__device__ __forceinline__ float sc_mt(const tri_rec& T, float ox, float oy, float oz, float dx, float dy, float dz) {
  const float eps = 0.000001f;
  const float hx = dy * T.e2z - dz * T.e2y, hy = dz * T.e2x - dx * T.e2z, hz = dx * T.e2y - dy * T.e2x;
  const float a = (T.e1x * hx + T.e1y * hy) + T.e1z * hz;
  if (a < eps && a > -eps) return NAN;
  const float inv_a = lt_rcp_ieee(a);  // = 1.0f / a, bit for bit (lt_internal.h)
  const float sx = ox - T.v0x, sy = oy - T.v0y, sz = oz - T.v0z;
  const float u = ((sx * hx + sy * hy) + sz * hz) * inv_a;
  if (u < 0 || u > 1) return NAN;
  const float qx = sy * T.e1z - sz * T.e1y, qy = sz * T.e1x - sx * T.e1z, qz = sx * T.e1y - sy * T.e1x;
  const float v = ((dx * qx + dy * qy) + dz * qz) * inv_a;
  if (v < 0 || u + v > 1) return NAN;
  const float t = ((T.e2x * qx + T.e2y * qy) + T.e2z * qz) * inv_a;
  if (t < eps) return NAN;
  return t;
}

// The angular bounds below only have to be CONSERVATIVE (they are padded by >= 3e-4 rad), so they use the
// hardware approximations (1 ulp rcp / sqrt) and a degree-11 minimax arctangent (max error 2e-6 rad) instead
// of the IEEE sequences the triangle test itself needs (fewer instructions on the dependent chain of a workgroup).
__device__ __forceinline__ float f_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float f_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float f_atan2(float y, float x) {
#pragma clang fp contract(fast)
  const float ax = fabsf(x), ay = fabsf(y);
  const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  const float a = mx > 0.f ? mn * f_rcp(mx) : 0.f;
  const float q = a * a;
  float r = a * (0.99997726f + q * (-0.33262347f + q * (0.19354346f + q * (-0.11643287f + q * (0.05265332f + q * -0.01172120f)))));
  r = ay > ax ? 1.57079632679f - r : r;
  r = x < 0.f ? 3.14159265359f - r : r;
  return y < 0.f ? -r : r;
}

// elevation pair atan2(za, ra), atan2(zb, rb) for ra, rb >= 0: when both |z / r| <= 1 in every lane of the wave
// (elevations within +-45 degrees -- everything a LiDAR sees except the ground right under it) the octant
// reduction of f_atan2 is not needed: the same polynomial on t = z / r directly, 9 instructions instead of 17.
__device__ __forceinline__ void f_atan_pair(float za, float ra, float zb, float rb, float& tha, float& thb) {
#pragma clang fp contract(fast)
  const float ta = za * f_rcp(ra), tb = zb * f_rcp(rb);
  const bool easy = fabsf(ta) <= 1.0f && fabsf(tb) <= 1.0f;  // false for NaN / inf (r == 0)
  if (__ballot(!easy) == 0ull) {  // wave-uniform: the common case has no general-path lane at all
    const float qa = ta * ta, qb = tb * tb;
    tha = ta * (0.99997726f + qa * (-0.33262347f + qa * (0.19354346f + qa * (-0.11643287f + qa * (0.05265332f + qa * -0.01172120f)))));
    thb = tb * (0.99997726f + qb * (-0.33262347f + qb * (0.19354346f + qb * (-0.11643287f + qb * (0.05265332f + qb * -0.01172120f)))));
  } else {
    tha = f_atan2(za, ra);
    thb = f_atan2(zb, rb);
  }
}

__device__ __forceinline__ float seg_dist2d_sq(float ax, float ay, float bx, float by) {  // |(0,0) - segment ab|^2
#pragma clang fp contract(fast)
  const float ex = bx - ax, ey = by - ay;
  const float l2 = ex * ex + ey * ey;
  float t = l2 > 0.f ? -(ax * ex + ay * ey) * f_rcp(l2) : 0.f;
  t = fminf(fmaxf(t, 0.f), 1.f);
  const float cx = ax + t * ex, cy = ay + t * ey;
  return cx * cx + cy * cy;
}

// Addressing: with WIDE == false every array of a launch is smaller than 4 GB (the host checks), so element
// addresses are base + 32-bit byte offset -- one VALU multiply instead of quarter-rate 64-bit multiply-adds on
// the gather addresses.
template <class T>
__device__ __forceinline__ unsigned byte_off(unsigned elem) {
  if (sizeof(T) == 12) {  // v_mul_lo_u32 is quarter rate and the optimiser folds (e << 3) + (e << 2) back into it
    unsigned r;
    asm("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(r) : "v"(elem), "v"(elem << 2));
    return r;
  }
  return elem * (unsigned)sizeof(T);
}
template <bool WIDE, class T>
__device__ __forceinline__ const T* at(const T* base, unsigned elem) {
  return WIDE ? base + (size_t)elem : (const T*)((const char*)base + byte_off<T>(elem));
}
template <bool WIDE, class T>
__device__ __forceinline__ T* at(T* base, unsigned elem) {
  return WIDE ? base + (size_t)elem : (T*)((char*)base + byte_off<T>(elem));
}
struct f3 { float x, y, z; };  // 12-B vertex / index triple (alignment 4)
struct i3 { int x, y, z; };

struct bin_rect { int a0, na, e0, e1; };  // azimuth: na bins starting at a0 (mod nb_az); elevation rows e0..e1

// conservative angular bounds of a triangle seen from the origin -> bin rectangle (na == 0: nothing to do).
// Everything in here is approximate-but-padded, so contraction to FMA is allowed (unlike the triangle test)
// and minima / maxima are taken on squared lengths (4 square roots per triangle instead of 15).
// (e1 = v1 - v0, e2 = v2 - v0 in the xy plane: the record's edge vectors, which the callers have anyway)
__device__ __forceinline__ bin_rect tri_bins(const rs_params& P, float x0, float y0, float z0, float x1, float y1,
                                             float z1, float x2, float y2, float z2, float e1x, float e1y, float e2x,
                                             float e2y) {
#pragma clang fp contract(fast)
  bin_rect R;  // only R.na is defined when the rectangle is empty (callers test R.na > 0 first)
  R.na = 0;
  const float q0 = x0 * x0 + y0 * y0, q1 = x1 * x1 + y1 * y1, q2 = x2 * x2 + y2 * y2;
  // The bounds below are only sound while no intermediate overflows or is NaN (fminf / fmaxf drop NaN
  // operands).  Sums, unlike maxima, propagate both, so one test guards the fast path:
  if (!(q0 + q1 + q2 < 1e36f && fabsf(z0) + fabsf(z1) + fabsf(z2) < 1e18f)) {
    //   a NaN or infinite coordinate (vertex or origin): the triangle test can never accept -- every chain
    //     of products reaches t as NaN, +-inf or 0, none of which is recorded (Triangle.h:47, BVH.cpp:59);
    //   finite coordinates beyond ~1e18 (a vertex flung far away: the sliver towards it CAN be hit): every bin.
    const float sum = ((x0 + y0) + (z0 + x1)) + ((y1 + z1) + (x2 + y2)) + z2;  // NaN or +-inf iff one of them is
    const float asum = ((fabsf(x0) + fabsf(y0)) + (fabsf(z0) + fabsf(x1))) + ((fabsf(y1) + fabsf(z1)) + (fabsf(x2) + fabsf(y2))) + fabsf(z2);
    if (sum != sum || asum == INFINITY) return R;
    R.e0 = 0; R.e1 = P.nb_el - 1; R.a0 = 0; R.na = P.nb_az;
    return R;
  }
  const float rho_max2 = fmaxf(q0, fmaxf(q1, q2));
  const float rho_max = f_sqrt(rho_max2);
  const float zmin = fminf(z0, fminf(z1, z2)), zmax = fmaxf(z0, fmaxf(z1, z2));
  // Where is the vertical axis through the origin relative to the triangle's xy-projection?
  //   mixed signs of the three sub-areas  -> strictly outside: the azimuth extent is an arc < pi
  //   all sub-areas ~ 0                   -> edge-on (projection is a segment): outside unless the segment
  //                                          itself reaches the axis (rho_min ~ 0)
  //   otherwise                           -> the axis pierces the triangle: every azimuth
  const float c0 = x0 * y1 - x1 * y0, c1 = x1 * y2 - x2 * y1, c2 = x2 * y0 - x0 * y2;
  const float tol = 1e-6f * rho_max2 + 1e-12f;
  const float cmin = fminf(c0, fminf(c1, c2)), cmax = fmaxf(c0, fmaxf(c1, c2));
  const bool outside = cmin < -tol && cmax > tol;
  bool edge_on = cmin >= -tol && cmax <= tol;
  // rho_edges2 = a lower bound of the squared distance from the axis to the triangle's edges.  The closest point
  // of an edge is at most half the edge away from one of its end points, at a right angle when it is interior:
  // d^2 >= min(q) - L^2 / 4 with L the longest edge.  For a triangle small against its distance from the axis
  // (every triangle of a 5 cm mesh beyond half a metre) that is within 0.13 % of the exact distance at a third
  // of its cost; waves holding a triangle for which it is loose take the three exact point-segment distances.
  const float e12x = e2x - e1x, e12y = e2y - e1y;
  const float lmax2 = fmaxf(e1x * e1x + e1y * e1y, fmaxf(e12x * e12x + e12y * e12y, e2x * e2x + e2y * e2y));
  const float qmin = fminf(q0, fminf(q1, q2));
  float rho_edges2;
  if (__ballot(!(lmax2 <= 0.01f * qmin)) == 0ull) {  // wave-uniform
    rho_edges2 = qmin - 0.25f * lmax2;
  } else {
    rho_edges2 =
        fminf(seg_dist2d_sq(x0, y0, x1, y1), fminf(seg_dist2d_sq(x1, y1, x2, y2), seg_dist2d_sq(x2, y2, x0, y0)));
  }
  const float rho_edges = f_sqrt(rho_edges2);
  const float rho_tiny = 1e-4f * rho_max + 1e-6f;
  // A triangle whose edges are all further from the axis than twice its longest edge cannot contain the axis, whatever
  // the sub-areas say: for a TINY triangle far away (marching cubes emits 40-micrometre triangles where the field is
  // ~0 at a grid point) they are all of the order of `tol`, neither clearly mixed nor clearly zero -- such triangles
  // were taken for pierced (every azimuth: 70 000 candidate bins each).  They take the three-azimuth path instead.
  if (!outside && rho_edges2 > 4.0f * lmax2) edge_on = true;
  const bool pierced = !(outside || (edge_on && rho_edges > rho_tiny));
  const float rho_min = pierced ? 0.f : rho_edges;
  // angular padding: float rounding of atan2 and of the ray bins, plus the positional slop (<= ~1e-4 m) with
  // which the float Moller-Trumbore test may accept a ray that passes just outside the triangle, seen from the
  // closest the triangle can be: every point of it has rho >= rho_min and |z| >= z_near
  const float z_near = __builtin_amdgcn_fmed3f(zmin, 0.f, zmax);  // 0 clamped into [zmin, zmax]
  const float dlo2 = fmaxf((pierced ? 0.f : rho_edges2) + z_near * z_near, 0.0025f);
  const float pad = 3e-4f + 2e-4f * __builtin_amdgcn_rsqf(dlo2);
  // the same positional slop seen in AZIMUTH subtends slop / rho (horizontal distance), not slop / distance: a
  // triangle high above or below the origin whose edge passes close to the vertical axis needs the wider pad
  const float pad_az = 3e-4f + 2e-4f * __builtin_amdgcn_rsqf(fmaxf(rho_edges2, 0.0025f));
  // elevation: z / rho over the triangle
  float th_hi, th_lo;
  f_atan_pair(zmax, zmax > 0.f ? rho_min : rho_max, zmin, zmin < 0.f ? rho_min : rho_max, th_hi, th_lo);
  th_hi += pad;
  th_lo -= pad;
  // rows whose rays can lie inside [th_lo, th_hi]; LT_BIN_SLACK covers the float rounding of the coordinates
  const float fe0 = ceilf(th_lo * P.el_scale + P.el_c0), fe1 = floorf(th_hi * P.el_scale + P.el_c1);  // (rs_derive)
  if (!(fe1 >= 0.f) || !(fe0 <= (float)(P.nb_el - 1)) || !(fe0 <= fe1)) return R;  // no row (or NaN)
  R.e0 = (int)fmaxf(fe0, 0.f);
  R.e1 = (int)fminf(fe1, (float)(P.nb_el - 1));
  // azimuth: full circle when pierced, else the shortest arc holding the three vertex azimuths
  if (pierced || rho_min <= rho_tiny || P.nb_az < 4) {
    R.a0 = 0;
    R.na = P.nb_az;
    return R;
  }
  float a_lo, a_hi;  // arc [a_lo, a_hi], a_hi may exceed pi (wraps)
  if (!edge_on) {
    // strictly outside: the sub-areas have mixed signs and name the extreme vertices without sorting angles
    // -- lo is the vertex both others are counter-clockwise of, hi the one both are clockwise of (a sign that
    // rounding could flip belongs to two vertices within ~1e-6 rad of each other: far below the padding)
    const bool l0 = c0 >= 0.f && c2 <= 0.f, l1 = c1 >= 0.f && c0 <= 0.f;
    const bool h0 = c0 <= 0.f && c2 >= 0.f, h1 = c1 <= 0.f && c0 >= 0.f;
    a_lo = f_atan2(l0 ? y0 : (l1 ? y1 : y2), l0 ? x0 : (l1 ? x1 : x2));
    a_hi = f_atan2(h0 ? y0 : (h1 ? y1 : y2), h0 ? x0 : (h1 ? x1 : x2));
    if (a_hi < a_lo) a_hi += 2.0f * LT_PI_F;
  } else {
    // all three azimuths (nearly) coincide: the shortest arc holding them
    const float p0 = f_atan2(y0, x0), p1 = f_atan2(y1, x1), p2 = f_atan2(y2, x2);
    const float lo = fminf(p0, fminf(p1, p2)), hi = fmaxf(p0, fmaxf(p1, p2));
    const float mid = (p0 + p1 + p2) - lo - hi;
    const float g0 = mid - lo, g1 = hi - mid, g2 = 2.0f * LT_PI_F - (hi - lo);
    if (g2 >= g0 && g2 >= g1) { a_lo = lo; a_hi = hi; }
    else if (g0 >= g1) { a_lo = mid; a_hi = lo + 2.0f * LT_PI_F; }
    else { a_lo = hi; a_hi = mid + 2.0f * LT_PI_F; }
  }
  a_lo -= pad_az;
  a_hi += pad_az;
  const float fa0 = ceilf(a_lo * P.az_scale + P.az_c0);
  const float fa1 = floorf(a_hi * P.az_scale + P.az_c1);
  const int na = (int)(fa1 - fa0) + 1;
  if (!(na >= 1)) return R;  // no ray column inside the arc (or NaN)
  if (na >= P.nb_az) { R.a0 = 0; R.na = P.nb_az; return R; }
  int a0 = (int)fa0;  // a_lo is in [-pi - pad, pi], |az_off| <= 0.5: at most one wrap either way (nb_az >= 4 here)
  if (a0 < 0) a0 += P.nb_az;
  else if (a0 >= P.nb_az) a0 -= P.nb_az;
  R.a0 = a0;
  R.na = na;
  return R;
}

// grid[bin] = the bin's ray when it holds exactly one (direction, ray index in .w), a zero direction with
// .w = -1 when it is empty (Moller-Trumbore rejects it at the determinant test), or (.x, .y) = the bin's slot
// range in sdirs with .w = -2 when it holds several rays (irregular ray sets; a sensor grid has one per bin).
// One 16-B load per candidate bin instead of the dependent bin_start -> direction chain.
#define LT_GRID_EMPTY (-1)
#define LT_GRID_MULTI (-2)
template <bool WIDE>
__device__ __forceinline__ void sc_hit(float t, int ray, int face, unsigned long long* __restrict__ cell) {
  // BVH.cpp:20, :59: a hit has to beat the initial t = 999999999.f (also false for sc_mt's NaN = "no hit")
  // (A relaxed look at the cell before the atomic -- skip it when the key cannot win -- was measured: 7.3 instead of
  // 10.0 Grays/s and 437 instead of 352 MB of traffic per batch; the loads pull every cell line into every XCD's L2.)
  if (t < 999999999.f) atomicMin(at<WIDE>(cell, (unsigned)ray), ((unsigned long long)__float_as_uint(t) << 32) | (unsigned)face);
}
template <bool WIDE>
__device__ __forceinline__ void sc_test_cell(const tri_rec& T, int face, const float4 g,
                                             const float4* __restrict__ sdirs, float ox, float oy, float oz,
                                             unsigned long long* __restrict__ cell, unsigned& n_tests) {
  const int w = __float_as_int(g.w);
  if (w != LT_GRID_MULTI) {
    n_tests += w >= 0 ? 1u : 0u;
    sc_hit<WIDE>(sc_mt(T, ox, oy, oz, g.x, g.y, g.z), w, face, cell);
  } else {
    for (int k = __float_as_int(g.x), e = __float_as_int(g.y); k < e; ++k) {
      const float4 d = *at<WIDE>(sdirs, (unsigned)k);
      ++n_tests;
      sc_hit<WIDE>(sc_mt(T, ox, oy, oz, d.x, d.y, d.z), __float_as_int(d.w), face, cell);
    }
  }
}

// sdirs = normalised directions in bin order, ray index in .w
__global__ __launch_bounds__(256) void k_rs_sortdirs(const float4* __restrict__ dirs,
                                                     const uint32_t* __restrict__ bin_rays, int n,
                                                     float4* __restrict__ sdirs) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const int ray = (int)bin_rays[k];
  float4 d = dirs[ray];
  d.w = __int_as_float(ray);
  sdirs[k] = d;
}

__global__ __launch_bounds__(256) void k_rs_grid(const int* __restrict__ bin_start,
                                                 const float4* __restrict__ sdirs, int nbins,
                                                 float4* __restrict__ grid) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= nbins) return;
  const int s = bin_start[b], e = bin_start[b + 1];
  float4 g = make_float4(0.f, 0.f, 0.f, __int_as_float(LT_GRID_EMPTY));
  if (e - s == 1) g = sdirs[s];
  else if (e - s > 1) g = make_float4(__int_as_float(s), __int_as_float(e), 0.f, __int_as_float(LT_GRID_MULTI));
  grid[b] = g;
}

#define LT_SC_BIG 512    // candidate bins above which a triangle goes to the wave-per-triangle queue
// Candidates a k_sc_tris workgroup tests itself; the rest of a heavier workgroup is queued in slices for k_sc_rest.
//   single-scan call: 1536 (6 rounds of 256; 1024 with round 3's 256-triangle workgroups: the same share of a
//     workgroup's candidates) -- a few near-field workgroups with 10-20 rounds would otherwise run long after the rest of
//     the grid has drained (measured: a 20 us tail on a 37 us kernel; round 4, k_sc_tris + k_sc_rest + k_sc_resolve of one
//     C2 scan: cap 1024 / 1536 / 8192 -> 37.6 / 34.7 / 40.4 us, profiles/r04/cap_iso.txt);
//   batch call: 8192 -- with the triangles of 8 scans in one grid there is no tail to speak of, and what is
//     deferred costs a second pass over its triangle block in k_sc_rest (7.6 -> 8.3 Grays/s on C2, where no
//     workgroup reaches 8192; the bound is there for a near field full of medium-sized triangles).
#define LT_SC_CAP_SINGLE 1536
#define LT_SC_CAP_BATCH 8192
#ifndef LT_SC_SLICE
#define LT_SC_SLICE 512  // candidates per queued slice (k_sc_rest)
#endif

// Triangles per workgroup of k_sc_tris / k_sc_rest.  A workgroup's life is a chain of dependent round trips -- index triple
// -> three vertices -> bounds -> prefix sum (two barriers) -> phase B -- at the hardware's 8 waves per SIMD (DESIGN.md section
// 5d), so the lever is how much work a wave has in flight per round trip: lanes 0 .. LT_SC_T - 257 take a SECOND triangle,
// both index triples are loaded first and then all six vertices, before the first of them is used.  448, not 512: the LDS
// of a workgroup must stay <= 20 KB for 8 workgroups (= 32 waves) per CU, which 448 records of 40 B + the prefix array are
// (20 000 B); -DLT_SC_T=256 is round 3's kernel (A/B).
#ifndef LT_SC_T
#define LT_SC_T 448
#endif
#define LT_SC_T2 (LT_SC_T - 256)  // lanes with a second triangle (a multiple of 64: whole waves)
static_assert(LT_SC_T >= 256 && LT_SC_T <= 512 && LT_SC_T2 % 64 == 0, "LT_SC_T: 256, 320, 384, 448 or 512");

// LDS state of one workgroup = LT_SC_T consecutive triangles; slot s = triangle first + s, owned by lane s & 255
struct sc_shared {
  // triangle record = 40 B in three arrays (one ds_read each; phase B is sensitive to the NUMBER of LDS instructions -- a
  // 4-ary search with 12 reads instead of the binary search's 9 cost 9 %):
  //   q0 = (v0.x, v0.y, v0.z, e1.x)   q1 = (e1.y, e1.z, e2.x, e2.y)   q2 = (e2.z, (na - 1) | e0 << 9)
  float4 q0[LT_SC_T], q1[LT_SC_T];
  float2 q2[LT_SC_T];
  // exclusive prefix sum of the candidate counts << 13 | a0 (first azimuth column of the slot's rectangle, < 8192); entry
  // LT_SC_T holds the total, the entries beyond are all-ones: the search below is a branch-free descent over 512 entries
  unsigned pre[513];
  unsigned wsum[4];
  int kept;
};

struct sc_one { int cnt; unsigned a0; };  // phase A's result for one triangle: candidate bins (0: none / invalid / big), a0

// bounds + LDS record of one triangle whose vertices are in registers
template <bool PUSH>
__device__ __forceinline__ sc_one sc_record_into(float4* s_q0, float4* s_q1, float2* s_q2, int slot, int f, const f3 A,
                                                 const f3 Bv, const f3 C, float ox, float oy, float oz, const rs_params& P,
                                                 int* __restrict__ large, int* __restrict__ large_count) {
  sc_one r;
  r.cnt = 0; r.a0 = 0;
  const float v0x = A.x, v0y = A.y, v0z = A.z, v1x = Bv.x, v1y = Bv.y, v1z = Bv.z;
  const float v2x = C.x, v2y = C.y, v2z = C.z;
  const float e1x = v1x - v0x, e1y = v1y - v0y, e2x = v2x - v0x, e2y = v2y - v0y;
  const bin_rect R = tri_bins(P, v0x - ox, v0y - oy, v0z - oz, v1x - ox, v1y - oy, v1z - oz, v2x - ox, v2y - oy,
                              v2z - oz, e1x, e1y, e2x, e2y);
  if (R.na > 0) {  // (rows e0..e1 are non-empty whenever na > 0)
    const int c32 = R.na * (R.e1 - R.e0 + 1);  // <= 8192 x 4096 bins (lt_rayset_create_dev)
    if (c32 > LT_SC_BIG) {
      if (PUSH) large[atomicAdd(large_count, 1)] = f;
    } else {
      r.cnt = c32;
      r.a0 = (unsigned)R.a0;  // < 8192
      s_q0[slot] = make_float4(v0x, v0y, v0z, e1x);
      s_q1[slot] = make_float4(e1y, v1z - v0z, e2x, e2y);
      s_q2[slot] = make_float2(v2z - v0z, __int_as_float((R.na - 1) | (R.e0 << 9)));  // na <= LT_SC_BIG = 512, e0 < 4096
    }
  }
  return r;
}
template <bool PUSH>
__device__ __forceinline__ sc_one sc_record(sc_shared& S, int slot, int f, const f3 A, const f3 Bv, const f3 C, float ox,
                                            float oy, float oz, const rs_params& P, int* __restrict__ large,
                                            int* __restrict__ large_count) {
  return sc_record_into<PUSH>(S.q0, S.q1, S.q2, slot, f, A, Bv, C, ox, oy, oz, P, large, large_count);
}

// Phase A: lane tid sets up triangle first + tid and (tid < LT_SC_T2) triangle first + 256 + tid: record + bin rectangle in
// LDS, candidate counts in rA / rB (big triangles are queued when PUSH)
template <bool PUSH, bool WIDE>
__device__ __forceinline__ void sc_setup(sc_shared& S, const float* __restrict__ verts, const int* __restrict__ faces,
                                         int n_verts, int n_faces, int first, float ox, float oy, float oz,
                                         const rs_params& P, int* __restrict__ large, int* __restrict__ large_count,
                                         unsigned* __restrict__ flags, sc_one& rA, sc_one& rB) {
  const int tid = threadIdx.x;
  rA.cnt = 0; rA.a0 = 0; rB.cnt = 0; rB.a0 = 0;
  const int fA = first + tid, fB = first + 256 + tid;
  const bool hasA = fA < n_faces, hasB = LT_SC_T2 > 0 && tid < LT_SC_T2 && fB < n_faces;
  if (n_verts <= 0) {  // (uniform) every index is out of range
    if (PUSH && (hasA || hasB)) atomicOr(flags, LT_FLAG_BAD_INDEX);
    return;
  }
  // Every load below is unconditional (a lane without a triangle, or with a bad index, reads element 0): nothing the
  // compiler could want to wait for stands between the two index loads and the six vertex loads.
  const i3 iA = *at<WIDE>((const i3*)faces, (unsigned)(hasA ? fA : 0));
  i3 iB = iA;
  if (LT_SC_T2 > 0) iB = *at<WIDE>((const i3*)faces, (unsigned)(hasB ? fB : 0));
  const unsigned nv = (unsigned)n_verts;
  const bool okA = hasA && (unsigned)iA.x < nv && (unsigned)iA.y < nv && (unsigned)iA.z < nv;
  const bool okB = hasB && (unsigned)iB.x < nv && (unsigned)iB.y < nv && (unsigned)iB.z < nv;
  const f3 A0 = *at<WIDE>((const f3*)verts, okA ? (unsigned)iA.x : 0u);
  const f3 A1 = *at<WIDE>((const f3*)verts, okA ? (unsigned)iA.y : 0u);
  const f3 A2 = *at<WIDE>((const f3*)verts, okA ? (unsigned)iA.z : 0u);
  f3 B0 = A0, B1 = A1, B2 = A2;
  if (LT_SC_T2 > 0) {
    B0 = *at<WIDE>((const f3*)verts, okB ? (unsigned)iB.x : 0u);
    B1 = *at<WIDE>((const f3*)verts, okB ? (unsigned)iB.y : 0u);
    B2 = *at<WIDE>((const f3*)verts, okB ? (unsigned)iB.z : 0u);
  }
  if (PUSH && ((hasA && !okA) || (hasB && !okB))) atomicOr(flags, LT_FLAG_BAD_INDEX);
  if (okA) rA = sc_record<PUSH>(S, tid, fA, A0, A1, A2, ox, oy, oz, P, large, large_count);
  if (LT_SC_T2 > 0 && okB) rB = sc_record<PUSH>(S, 256 + tid, fB, B0, B1, B2, ox, oy, oz, P, large, large_count);
}

// exclusive prefix sums of the candidate counts over the workgroup's slots (0 .. 255: the lanes' first triangles, 256 ..:
// their second ones) -> S.pre[0 .. 512]; returns the lane's two prefixes.  Contains one barrier; the caller issues the
// second one (after any extra LDS it wants published with it).
__device__ __forceinline__ void sc_prefix(sc_shared& S, const sc_one rA, const sc_one rB, int& preA, int& preB, int& total) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // inclusive wave scan with six DPP adds (row_shr 1/2/4/8 inside the rows of 16, then row_bcast 15 / 31 across
  // rows).  Written as v_add_u32_dpp so that a step is ONE instruction: lanes whose DPP source does not exist add
  // 0 (bound_ctrl:0), rows deselected by row_mask keep their value.  The s_nop is the VALU-write -> DPP-read
  // hazard of gfx9 (2 wait states), which the assembler does not insert inside inline asm.
  // Both counts ride in one word: a count is <= LT_SC_BIG = 512, a wave's sum <= 2^15 -- the halves never carry.
  unsigned inc = (unsigned)rA.cnt | ((unsigned)rB.cnt << 16);
  asm volatile(
      "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(inc));
  if (lane == 63) S.wsum[wave] = inc;
  __syncthreads();
  int woffA = 0, woffB = 0, totA = 0, totB = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const unsigned ws = S.wsum[w];
    const int a = (int)(ws & 0xFFFFu), b = (int)(ws >> 16);
    if (w < wave) { woffA += a; woffB += b; }
    totA += a; totB += b;
  }
  total = totA + totB;  // <= 512 x 512 = 2^18: the << 13 below fits
  preA = woffA + (int)(inc & 0xFFFFu) - rA.cnt;
  preB = totA + woffB + (int)(inc >> 16) - rB.cnt;
  S.pre[tid] = ((unsigned)preA << 13) | rA.a0;
  if (LT_SC_T2 > 0 && tid < LT_SC_T2) S.pre[256 + tid] = ((unsigned)preB << 13) | rB.a0;
  else S.pre[256 + tid] = (256 + tid == LT_SC_T) ? ((unsigned)total << 13) : 0xFFFFFFFFu;
  if (tid == 255) S.pre[512] = LT_SC_T == 512 ? ((unsigned)total << 13) : 0xFFFFFFFFu;
}

// Phase B over the candidates [c_begin, c_end) of the workgroup, dealt ROUND-ROBIN: in every iteration the 64
// lanes of a wave hold 64 consecutive candidates = consecutive bins of a few neighbouring triangles, so the
// grid loads and the atomics of a wave fall into a handful of cache lines, and every lane runs the same
// number of Moller-Trumbore tests no matter how unevenly the triangles cover the image.  (A contiguous chunk
// per thread needs no search but made every lane touch its own lines: L2 bound.)  The loop is
// software-pipelined: the search and the grid load of the NEXT candidate are issued before the triangle test
// of the current one.
// NT = lanes sharing the candidates (256: a workgroup; 64: one wave of k_sc_rest), NE = entries of the prefix array the
// descent covers (a power of two; the array has NE + 1 entries, all-ones beyond the slots in use), tix = this lane's index
template <bool COUNT, bool WIDE, int NT, int NE>
__device__ __forceinline__ void sc_round_robin(const unsigned* s_pre, const float4* s_q0, const float4* s_q1,
                                               const float2* s_q2, const rs_params& P, const float4* __restrict__ grid,
                                               const float4* __restrict__ sdirs, unsigned long long* __restrict__ cell,
                                               int first_face, int c_begin, int c_end, int tix, float ox, float oy, float oz,
                                               unsigned& n_tests, unsigned& n_cand) {
  // LDS slots are addressed by BYTE offset j4 = 4 * j throughout (one shift less per search step / array read)
  auto ldu = [](const unsigned* arr, unsigned j4) { return *(const unsigned*)((const char*)arr + j4); };
  auto ldq = [](const float4* arr, unsigned j4) { return *(const float4*)((const char*)arr + 4u * j4); };
  auto ld2 = [](const float2* arr, unsigned j4) { return *(const float2*)((const char*)arr + 2u * j4); };
  const unsigned pre0 = s_pre[0];
  // slot j = largest j with prefix[j] <= c (pre = prefix << 13 | a0: compare against c << 13 | all-ones); the entry found is
  // carried along, so the descent needs no read of pre[j] afterwards; returns the bin and the record's third word
  auto locate = [&](int c, unsigned& j4, float2& q2) -> int {
    const unsigned ck = ((unsigned)c << 13) | 0x1FFFu;
    j4 = 0;
    unsigned base = pre0;
#pragma unroll
    for (unsigned step4 = 2u * NE; step4 >= 4; step4 >>= 1) {
      const unsigned v = ldu(s_pre, j4 + step4);
      const bool take = v <= ck;
      j4 = take ? j4 + step4 : j4;
      base = take ? v : base;
    }
    const int local = c - (int)(base >> 13);
    q2 = ld2(s_q2, j4);
    const int pk = __float_as_int(q2.y);
    const int na = (pk & 511) + 1;
    // local / na: (local + 0.5) / na is >= 0.5 / na away from an integer and local / na <= LT_SC_BIG / na, so
    // a relative error of 2^-22 in the product (a 1-ulp reciprocal) cannot cross one
    const int row = (int)(((float)local + 0.5f) * f_rcp((float)na));
    // rows < 4096 and columns <= 8192 (lt_rayset_create_dev): 24-bit multiplies, full rate (v_mul_lo_u32 is 1/4)
    int az = (int)(base & 0x1FFFu) + (local - __mul24(row, na));
    if (az >= P.nb_az) az -= P.nb_az;
    return __mul24((pk >> 9) + row, P.nb_az) + az;
  };
  int c = c_begin + tix;
  if (c < c_end) {
    unsigned j4;
    float2 q2;
    float4 g = *at<false>(grid, (unsigned)locate(c, j4, q2));  // <= 8192 x 4096 bins x 16 B: always < 4 GB
    for (;;) {
      // prefetch of the next candidate (index clamped to the last one: no lane-level branch, the load stays in flight
      // across the test) -- unless NO lane of the wave has one: round-robin dealing makes that wave-uniform up to the
      // one wave holding c_end, and the search + bin arithmetic of a prefetch nobody uses were 9 % of the kernel's
      // vector instructions
      const int cn = c + NT;
      unsigned jn4 = 0;
      float2 q2n = make_float2(0.f, 0.f);
      float4 gn = make_float4(0.f, 0.f, 0.f, 0.f);
      if (__ballot(cn < c_end) != 0ull) gn = *at<false>(grid, (unsigned)locate(min(cn, c_end - 1), jn4, q2n));
      const float4 q0 = ldq(s_q0, j4), q1 = ldq(s_q1, j4);
      tri_rec T;
      T.v0x = q0.x; T.v0y = q0.y; T.v0z = q0.z;
      T.e1x = q0.w; T.e1y = q1.x; T.e1z = q1.y;
      T.e2x = q1.z; T.e2y = q1.w; T.e2z = q2.x;
      if (COUNT) ++n_cand;
      sc_test_cell<WIDE>(T, first_face + (int)(j4 >> 2), g, sdirs, ox, oy, oz, cell, n_tests);
      if (cn >= c_end) break;
      c = cn; j4 = jn4; g = gn; q2 = q2n;
    }
  }
}

template <bool COUNT>
__device__ __forceinline__ void sc_count(unsigned n_tests, unsigned n_cand, unsigned long long* __restrict__ counters) {
  if (COUNT) {
    unsigned long long vt = n_tests, vc = n_cand;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { vt += __shfl_xor(vt, o, 64); vc += __shfl_xor(vc, o, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&counters[1], vt); atomicAdd(&counters[0], vc); }
  }
}

// ---- one launch, several scans ------------------------------------------------------------------------------------
// A scan is three small kernels; with one launch per kernel and scan, the hardware queues spend more time
// between kernels than the kernels need.  Every kernel below therefore takes a BATCH of
// scans: the per-scan arguments live in a job table passed by value (kernel arguments: scalar registers), and a
// workgroup looks up its scan from its block index.  A batch of one is the single-scan API.
#ifndef LT_SC_MAX_BATCH
#define LT_SC_MAX_BATCH 8
#endif
struct sc_job {
  const float* verts; const int* faces; const int* colors; const float* rem;  // the scan's mesh
  rs_params P; const float4* grid; const float4* sdirs; const float4* dirs;  // its ray set (P: by value, see lt_rayset)
  unsigned long long* cell; int* large; int* large_count; int2* slices;
  unsigned* flags; unsigned long long* counters;
  float* endpoints; int* endcolors; float* range; float* endrem; int* tri;  // its images
  float ox, oy, oz;
  int n_verts, n_faces, n_rays, cap_slices;
  unsigned out_flags;
};
struct sc_batch {
  int n;
  int tris_blocks, resolve_blocks;  // grid sizes
  int cap;                          // LT_SC_CAP_SINGLE / LT_SC_CAP_BATCH
  int rest_blocks;                  // workgroups of k_sc_rest per scan
  // first workgroup of every scan in k_sc_tris / k_sc_resolve (INT_MAX beyond n).  Side by side at the head of the
  // argument block: a workgroup finds its scan with ONE scalar load + compares -- walking job[k].tris_block0 was a
  // chain of up to seven dependent scalar loads before the first vector load of the workgroup could be issued
  int tris_block0[LT_SC_MAX_BATCH];
  int resolve_block0[LT_SC_MAX_BATCH];
  sc_job job[LT_SC_MAX_BATCH];
};
__device__ __forceinline__ int sc_find_job(const int (&block0)[LT_SC_MAX_BATCH], int block) {
  int j = 0;
#pragma unroll
  for (int k = 1; k < LT_SC_MAX_BATCH; ++k) j += block >= block0[k] ? 1 : 0;
  return j;
}


// One workgroup = LT_SC_T consecutive triangles of one scan: phase A, prefix sum, phase B over its first ~B.cap
// candidates; what is left is queued as (workgroup, first candidate) slices for k_sc_rest.
template <bool COUNT, bool WIDE>
__global__ __launch_bounds__(256) void k_sc_tris(const sc_batch B) {
  __shared__ sc_shared S;
  const int j = sc_find_job(B.tris_block0, (int)blockIdx.x);
  const sc_job& J = B.job[j];
  const int lb = (int)blockIdx.x - B.tris_block0[j];  // workgroup index inside the scan
  const int tid = threadIdx.x;
  const int first = lb * LT_SC_T;
  const rs_params P = J.P;
  const float ox = J.ox, oy = J.oy, oz = J.oz;
  sc_one rA, rB;
  sc_setup<true, WIDE>(S, J.verts, J.faces, J.n_verts, J.n_faces, first, ox, oy, oz, P, J.large, J.large_count, J.flags, rA,
                       rB);
  int total, preA, preB;
  sc_prefix(S, rA, rB, preA, preB, total);
  // The workgroup keeps the triangles that start below the cap; exactly one slot sees the crossing and its lane queues
  // the rest in slices of LT_SC_SLICE candidates (large_count[1] = number of slices; if the queue is full the workgroup
  // keeps all).
  if (total < B.cap) {
    if (tid == 255) S.kept = total;
  } else {
    const bool crossA = preA < B.cap && preA + rA.cnt >= B.cap;
    const bool crossB = LT_SC_T2 > 0 && tid < LT_SC_T2 && preB < B.cap && preB + rB.cnt >= B.cap;
    if (crossA || crossB) {
      int kept = crossA ? preA + rA.cnt : preB + rB.cnt;
      const int n_sl = (total - kept + LT_SC_SLICE - 1) / LT_SC_SLICE;
      if (n_sl > 0) {
        const int base = atomicAdd(&J.large_count[1], n_sl);
        if (base + n_sl <= J.cap_slices) {
          for (int k = 0; k < n_sl; ++k) J.slices[base + k] = make_int2(lb, kept + k * LT_SC_SLICE);
        } else {
          atomicSub(&J.large_count[1], n_sl);
          kept = total;
        }
      }
      S.kept = kept;
    }
  }
  __syncthreads();
  const int kept = S.kept;
  unsigned n_tests = 0, n_cand = 0;
  sc_round_robin<COUNT, WIDE, 256, (LT_SC_T > 256 ? 512 : 256)>(S.pre, S.q0, S.q1, S.q2, P, J.grid, J.sdirs, J.cell, first, 0,
                                                                kept, tid, ox, oy, oz, n_tests, n_cand);
  sc_count<COUNT>(n_tests, n_cand, J.counters);
}

// The rest, LT_SC_REST_BLOCKS workgroups per scan: (1) queued slices of heavy workgroups -- a workgroup redoes
// phase A of that triangle block (same code, same prefix sums) and tests one slice of its candidates; (2) big
// triangles, up to LT_SC_PARTS waves each, lanes stride over the candidate bins.
// (Round 4 also built ONE WAVE per slice -- the slice naming its own triangles, aligned to the group of 64 slots a wave of
// k_sc_tris held, because tri_bins takes wave-uniform short cuts and a triangle's candidate count is only reproduced inside
// the same group -- to make deferring cheap and the batch cap low: bit-identical, but a single-scan k_sc_rest took 15-20 us
// instead of 10 (a wave's slice is a serial chain of two gathers, the bounds and eight rounds of 64 candidates), and lower
// batch caps did not pay either: deferred work is conserved, not saved.  DESIGN.md section 5d, profiles/r04.)
#define LT_SC_REST_BLOCKS 512        // single-scan call
#define LT_SC_REST_BLOCKS_BATCH 128  // per scan of a batch call: its queues are all but empty (cap 8192), and 4096 workgroups that only look at
                                    // two counters still wait for slots behind the other batches' k_sc_tris: 512 / 128 / 64 / 32 -> 10.79 / 10.93 / 10.93 /
                                    // 10.96 Grays/s (profiles/r04/rest_blocks.txt); 128 keeps headroom for scenes that do queue
#define LT_SC_PARTS 8
template <bool COUNT, bool WIDE>
__global__ __launch_bounds__(256) void k_sc_rest(const sc_batch B) {
  __shared__ sc_shared S;
  const int rest_blocks = B.rest_blocks;  // workgroups per scan (grid-stride loops below: any number works)
  const sc_job& J = B.job[blockIdx.x / rest_blocks];
  const int rb = blockIdx.x % rest_blocks;
  if (J.n_faces <= 0) return;
  const float* __restrict__ verts = J.verts;
  const int* __restrict__ faces = J.faces;
  const float4* __restrict__ grid = J.grid;
  const float4* __restrict__ sdirs = J.sdirs;
  unsigned long long* __restrict__ cell = J.cell;
  const float ox = J.ox, oy = J.oy, oz = J.oz;
  const rs_params P = J.P;
  const int n_large = J.large_count[0], n_slices = min(J.large_count[1], J.cap_slices);
  if (COUNT && rb == 0 && threadIdx.x == 0) J.counters[3] = (unsigned long long)n_large | ((unsigned long long)n_slices << 32);
  unsigned n_tests = 0, n_cand = 0;
  for (int q = rb; q < n_slices; q += rest_blocks) {
    const int2 sl = J.slices[q];
    const int first = sl.x * LT_SC_T;
    sc_one rA, rB;
    sc_setup<false, WIDE>(S, verts, faces, J.n_verts, J.n_faces, first, ox, oy, oz, P, nullptr, nullptr, nullptr, rA, rB);
    int total, preA, preB;
    sc_prefix(S, rA, rB, preA, preB, total);
    __syncthreads();
    sc_round_robin<COUNT, WIDE, 256, (LT_SC_T > 256 ? 512 : 256)>(S.pre, S.q0, S.q1, S.q2, P, grid, sdirs, cell, first, sl.y,
                                                                  min(sl.y + LT_SC_SLICE, total), (int)threadIdx.x, ox, oy,
                                                                  oz, n_tests, n_cand);
    __syncthreads();  // LDS is reused by the next slice
  }
  const int lane = threadIdx.x & 63;
  const int wave0 = rb * 4 + (threadIdx.x >> 6), nwaves = rest_blocks * 4;
  // A big triangle is shared by up to LT_SC_PARTS waves (a ground triangle under the sensor of a low-poly
  // scene covers tens of thousands of bins): work item v = (triangle q, part p); every wave of a triangle
  // recomputes its bounds, part p takes the candidates p*64 + lane, stride 64 * (parts this triangle needs).
  for (int v = wave0; v < n_large * LT_SC_PARTS; v += nwaves) {
    const int q = v / LT_SC_PARTS, part = v - q * LT_SC_PARTS;
    const int f = J.large[q];
    const float* pa = verts + 3 * (size_t)faces[3 * (size_t)f];
    const float* pb = verts + 3 * (size_t)faces[3 * (size_t)f + 1];
    const float* pc = verts + 3 * (size_t)faces[3 * (size_t)f + 2];
    tri_rec T;
    T.v0x = pa[0]; T.v0y = pa[1]; T.v0z = pa[2];
    const float v1x = pb[0], v1y = pb[1], v1z = pb[2], v2x = pc[0], v2y = pc[1], v2z = pc[2];
    T.e1x = v1x - T.v0x; T.e1y = v1y - T.v0y; T.e1z = v1z - T.v0z;
    T.e2x = v2x - T.v0x; T.e2y = v2y - T.v0y; T.e2z = v2z - T.v0z;
    const bin_rect R = tri_bins(P, T.v0x - ox, T.v0y - oy, T.v0z - oz, v1x - ox, v1y - oy, v1z - oz, v2x - ox,
                                v2y - oy, v2z - oz, T.e1x, T.e1y, T.e2x, T.e2y);
    const int total = R.na > 0 ? R.na * (R.e1 - R.e0 + 1) : 0;  // <= 8192 x 4096 bins (lt_rayset_create_dev)
    const int parts = min(max((total + 1023) / 1024, 1), LT_SC_PARTS);
    if (part >= parts) continue;
    for (int w = part * 64 + lane; w < total; w += 64 * parts) {
      const int row = w / R.na;
      const int e = R.e0 + row;
      int az = R.a0 + (w - row * R.na);
      if (az >= P.nb_az) az -= P.nb_az;
      if (COUNT) ++n_cand;
      sc_test_cell<WIDE>(T, f, grid[e * P.nb_az + az], sdirs, ox, oy, oz, cell, n_tests);
    }
  }
  sc_count<COUNT>(n_tests, n_cand, J.counters);
}

// one thread per ray: unpack the winning (t, face), write back as RayTracer.cpp:73-90, re-arm the cell
template <bool COUNT>
__global__ __launch_bounds__(256) void k_sc_resolve(const sc_batch B) {
  const int j = sc_find_job(B.resolve_block0, (int)blockIdx.x);
  const sc_job& J = B.job[j];
  const int ray = ((int)blockIdx.x - B.resolve_block0[j]) * 256 + threadIdx.x;
  unsigned long long* __restrict__ cell = J.cell;
  const int* __restrict__ faces = J.faces;
  float* __restrict__ endpoints = J.endpoints;
  int* __restrict__ endcolors = J.endcolors;
  float* __restrict__ range = J.range;
  float* __restrict__ endrem = J.endrem;
  int* __restrict__ tri_out = J.tri;
  const unsigned flags = J.out_flags;
  if (ray == 0) { J.large_count[0] = 0; J.large_count[1] = 0; }
  bool hit = false;
  if (ray < J.n_rays) {
    const unsigned long long key = cell[ray];
    const float4 d = J.dirs[ray];  // (does not wait for the key: issued with it)
    cell[ray] = LT_EMPTY_KEY;
    hit = key != LT_EMPTY_KEY;
    if (hit) {
      const float t = __uint_as_float((unsigned)(key >> 32));
      const int face = (int)(unsigned)(key & 0xFFFFFFFFull);
      const int i0 = faces[3 * (size_t)face], i1 = faces[3 * (size_t)face + 1], i2 = faces[3 * (size_t)face + 2];
      // everything the images need is loaded BEFORE the first store: with the stores in between the compiler waited for
      // each attribute in turn -- key, face, colour, remission were four dependent round trips, three are needed
      const int* __restrict__ colors = J.colors;
      const float* __restrict__ rem = J.rem;
      const bool label = (flags & LT_TRACE_LABEL_IMAGE) != 0;  // deform's unpack label_image = ray_colors[..., 2], laserscan.py:912
      int c0 = 0, c1 = 0, c2 = 0;
      float r0 = 0.f, r1 = 0.f, r2 = 0.f;
      if (endcolors) {
        c2 = colors[3 * (size_t)i0 + 2];
        if (!label) { c0 = colors[3 * (size_t)i0]; c1 = colors[3 * (size_t)i0 + 1]; }
      }
      if (endrem) { r0 = rem[i0]; r1 = rem[i1]; r2 = rem[i2]; }
      if (endpoints) {
        endpoints[3 * (size_t)ray] = J.ox + d.x * t;
        endpoints[3 * (size_t)ray + 1] = J.oy + d.y * t;
        endpoints[3 * (size_t)ray + 2] = J.oz + d.z * t;
      }
      if (endcolors) {
        if (label) {
          endcolors[ray] = (int)(float)c2;
        } else {
          endcolors[3 * (size_t)ray] = (int)(float)c0;
          endcolors[3 * (size_t)ray + 1] = (int)(float)c1;
          endcolors[3 * (size_t)ray + 2] = (int)(float)c2;
        }
      }
      if (endrem) endrem[ray] = ((r0 + r1) + r2) / 3.0f;
      if (range) range[ray] = t;
      if (tri_out) tri_out[ray] = face;
    } else if (flags & LT_TRACE_WRITE_MISSES) {
      if (endpoints) { endpoints[3 * (size_t)ray] = 0.f; endpoints[3 * (size_t)ray + 1] = 0.f; endpoints[3 * (size_t)ray + 2] = 0.f; }
      if (endcolors) {
        if (flags & LT_TRACE_LABEL_IMAGE) endcolors[ray] = 0;
        else { endcolors[3 * (size_t)ray] = 0; endcolors[3 * (size_t)ray + 1] = 0; endcolors[3 * (size_t)ray + 2] = 0; }
      }
      if (endrem) endrem[ray] = 0.f;
      if (range) range[ray] = 0.f;
      if (tri_out) tri_out[ray] = -1;
    }
  }
  if (COUNT) {
    unsigned long long vh = hit ? 1u : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vh += __shfl_xor(vh, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&J.counters[2], vh);
  }
}

// ---- host side ---------------------------------------------------------------------------------------------
// sort kernels of lt_build.hip
void lt_sort_pairs(uint32_t* keys[2], uint32_t* vals[2], uint32_t* hist, int n, int key_bits, hipStream_t stream,
                   int* out_buffer);

// el_c0 = -el_lo el_scale - (dev_el + slack), el_c1 = ... + ..., az_c0 = pi az_scale - az_off - (dev_az + slack), az_c1 = ... +
static void rs_derive(rs_params& p) {
  const float de = p.dev_el + LT_BIN_SLACK, da = p.dev_az + LT_BIN_SLACK;
  p.el_c0 = -p.el_lo * p.el_scale - de;
  p.el_c1 = -p.el_lo * p.el_scale + de;
  p.az_c0 = LT_PI_F * p.az_scale - p.az_off - da;
  p.az_c1 = LT_PI_F * p.az_scale - p.az_off + da;
}

struct lt_rayset {
  int device;
  int n_rays, height, nb_az, nb_el;
  unsigned norm_flags;
  float4* dirs;
  float2* ang;
  float* partial;
  rs_params* prm;      // bin grid parameters, fitted on the device ...
  rs_params prm_host;  // ... and copied back once: the kernels get them as arguments instead of through a
                       // dependent load at the start of every workgroup
  uint32_t* keys[2];
  uint32_t* vals[2];
  uint32_t* hist;
  int* bin_start;
  uint32_t* bin_rays;  // = vals[sorted buffer]
  float4* sdirs;       // directions in bin order, ray index in .w
  float4* grid;        // one entry per bin, see sc_test_cell
};  // read-only once created: the state of a render lives in the scene (lt_internal.h)

static void rs_free(lt_rayset* r) {
  void* ps[] = {r->grid, r->sdirs, r->dirs, r->ang, r->partial, r->prm, r->keys[0], r->keys[1], r->vals[0], r->vals[1], r->hist,
                r->bin_start};
  for (void* p : ps)
    if (p) (void)hipFree(p);
}

extern "C" int lt_rayset_destroy(lt_rayset* r) {
  if (!r) return LT_OK;
  (void)hipSetDevice(r->device);
  (void)hipDeviceSynchronize();
  rs_free(r);
  free(r);
  return LT_OK;
}

extern "C" int lt_rayset_create_dev(lt_rayset** out, const float* rays, int n_rays, int height, unsigned flags,
                                    void* stream_) {
  if (!out || n_rays < 0 || height <= 0 || (n_rays > 0 && !rays)) {
    lt_set_error("lt_rayset_create_dev: invalid argument (n_rays=%d height=%d)", n_rays, height);
    return LT_ERR_INVALID_ARG;
  }
  *out = nullptr;
  hipStream_t stream = (hipStream_t)stream_;
  int device = 0;
  LT_HIP(hipGetDevice(&device));
  lt_rayset* r = (lt_rayset*)calloc(1, sizeof(lt_rayset));
  if (!r) return LT_ERR_NO_MEMORY;
  r->device = device;
  const int W = n_rays / height;
  const int n = W * height;  // RayTracer.cpp:56: rays beyond W * height are ignored
  r->n_rays = n;
  r->height = height;
  r->norm_flags = flags & (LT_TRACE_NORM_EXACT | LT_TRACE_NORM_AMD);
  r->nb_az = W < 1 ? 1 : (W > 8192 ? 8192 : W);
  r->nb_el = height > 4096 ? 4096 : height;
  const size_t nbins = (size_t)r->nb_az * r->nb_el + 1;  // + the NaN bin
  const size_t nn = n > 0 ? n : 1;
  const int nb = (int)((nn + LT_SORT_TILE - 1) / LT_SORT_TILE);
  int rc = LT_OK;
#define RS_ALLOC(ptr, bytes)                                                  \
  if (rc == LT_OK && hipMalloc((void**)&(ptr), (bytes)) != hipSuccess) {      \
    lt_set_error("lt_rayset_create_dev: hipMalloc of %zu bytes failed", (size_t)(bytes)); \
    rc = LT_ERR_NO_MEMORY;                                                    \
  }
  RS_ALLOC(r->dirs, nn * sizeof(float4));
  RS_ALLOC(r->ang, nn * sizeof(float2));
  RS_ALLOC(r->sdirs, nn * sizeof(float4));
  RS_ALLOC(r->partial, 2 * 256 * sizeof(float));
  RS_ALLOC(r->prm, sizeof(rs_params));
  for (int k = 0; k < 2; ++k) {
    RS_ALLOC(r->keys[k], nn * sizeof(uint32_t));
    RS_ALLOC(r->vals[k], nn * sizeof(uint32_t));
  }
  RS_ALLOC(r->hist, ((size_t)1024 * nb + 1024) * sizeof(uint32_t));
  RS_ALLOC(r->bin_start, (nbins + 1) * sizeof(int));
  RS_ALLOC(r->grid, nbins * sizeof(float4));
#undef RS_ALLOC
  if (rc != LT_OK) {
    rs_free(r);
    free(r);
    return rc;
  }
  if (hipMemsetAsync(r->prm, 0, sizeof(rs_params), stream) != hipSuccess) {
    lt_set_error("lt_rayset_create_dev: hipMemsetAsync failed");
    rs_free(r);
    free(r);
    return LT_ERR_HIP;
  }
  hipLaunchKernelGGL(k_rs_dirs, dim3(256), dim3(256), 0, stream, rays, n, r->norm_flags, r->dirs, r->ang, r->partial);
  // (prm->dev_az / dev_el start at 0: hipMemsetAsync below, before k_rs_keys accumulates them)
  hipLaunchKernelGGL(k_rs_fit, dim3(64), dim3(256), 0, stream, r->ang, n, r->nb_az, r->prm);
  hipLaunchKernelGGL(k_rs_keys, dim3((n + 255) / 256 > 0 ? (n + 255) / 256 : 1), dim3(256), 0, stream, r->ang, n,
                     r->nb_az, r->nb_el, r->partial, r->prm, r->keys[0], r->vals[0]);
  int buf = 0;
  if (n > 0) lt_sort_pairs(r->keys, r->vals, r->hist, n, 30, stream, &buf);
  r->bin_rays = r->vals[buf];
  hipLaunchKernelGGL(k_rs_starts, dim3((n + 1 + 255) / 256), dim3(256), 0, stream, r->keys[buf], n, (int)nbins,
                     r->bin_start);
  if (n > 0)
    hipLaunchKernelGGL(k_rs_sortdirs, dim3((n + 255) / 256), dim3(256), 0, stream, r->dirs, r->bin_rays, n, r->sdirs);
  hipLaunchKernelGGL(k_rs_grid, dim3((unsigned)((nbins + 255) / 256)), dim3(256), 0, stream, r->bin_start, r->sdirs,
                     (int)nbins, r->grid);
  const hipError_t le = hipGetLastError();
  if (le != hipSuccess) {
    lt_set_error("lt_rayset_create_dev: kernel launch failed: %s", hipGetErrorString(le));
    (void)hipStreamSynchronize(stream);
    rs_free(r);
    free(r);
    return LT_ERR_HIP;
  }
  // the one synchronisation of a ray set's life (once per sensor model): from here on it is read-only and may be
  // used on any stream
  if (hipStreamSynchronize(stream) != hipSuccess ||
      hipMemcpy(&r->prm_host, r->prm, sizeof(rs_params), hipMemcpyDeviceToHost) != hipSuccess) {
    lt_set_error("lt_rayset_create_dev: ray set preparation failed: %s", hipGetErrorString(hipGetLastError()));
    rs_free(r);
    free(r);
    return LT_ERR_HIP;
  }
  rs_derive(r->prm_host);
  *out = r;
  return LT_OK;
}

// Per-scene state of a render: the z-min cells (one per ray, armed = all-ones; k_sc_resolve re-arms what it reads,
// so they are set once per allocation) and the queues of k_sc_tris -- every triangle is queued at most once as
// "big", and a block of LT_SC_T triangles with <= LT_SC_BIG candidates each leaves at most
// LT_SC_T * LT_SC_BIG / LT_SC_SLICE = LT_SC_T slices.
static int sc_reserve(lt_scene* s, int n_faces, int n_rays, hipStream_t stream) {
  if (!s->sc_large_count) {
    LT_HIP(hipMalloc((void**)&s->sc_large_count, 4 * sizeof(int)));
    LT_HIP(hipMemsetAsync(s->sc_large_count, 0, 4 * sizeof(int), stream));
  }
  if (n_rays > s->sc_cap_cells) {
    if (s->sc_cell) {
      LT_HIP(hipDeviceSynchronize());
      (void)hipFree(s->sc_cell);
      s->sc_cell = nullptr;
      s->sc_cap_cells = 0;
    }
    LT_HIP(hipMalloc((void**)&s->sc_cell, (size_t)n_rays * sizeof(unsigned long long)));
    s->sc_cap_cells = n_rays;
    LT_HIP(hipMemsetAsync(s->sc_cell, 0xFF, (size_t)n_rays * sizeof(unsigned long long), stream));
  }
  if (n_faces > s->sc_cap_queue) {
    if (s->sc_large || s->sc_slices) {
      LT_HIP(hipDeviceSynchronize());
      (void)hipFree(s->sc_large);
      (void)hipFree(s->sc_slices);
      s->sc_large = nullptr;
      s->sc_slices = nullptr;
      s->sc_cap_queue = 0;
    }
    const size_t cap = (size_t)n_faces + n_faces / 4 + 1024;
    LT_HIP(hipMalloc((void**)&s->sc_large, cap * sizeof(int)));
    LT_HIP(hipMalloc((void**)&s->sc_slices, cap * sizeof(int2)));
    s->sc_cap_queue = (int)cap;
  }
  return LT_OK;
}

// ---- launching a batch -----------------------------------------------------------------------------------------------
struct sc_item {  // host view of one scan of a batch
  lt_scene* s;
  lt_rayset* r;
  const float* origin;
  float* endpoints; int* endcolors; float* range; float* endrem; int* tri;
};

// Three launches for the whole batch (k_sc_tris, k_sc_rest, k_sc_resolve) on `stream`.  `probe` = the scene
// whose probe events (lt_scene_set_probe) are recorded around k_sc_tris, or nullptr.
static int sc_launch_batch(const sc_item* it, int n_items, unsigned flags, hipStream_t stream, bool count,
                           lt_scene* probe) {
  sc_batch B;
  memset(&B, 0, sizeof(B));
  for (int k = 0; k < LT_SC_MAX_BATCH; ++k) B.tris_block0[k] = B.resolve_block0[k] = 0x7fffffff;
  // LIDARHIP_FORCE_WIDE=1 selects the 64-bit addressing variants on any input (they are otherwise only reached
  // with > 357 M triangles): a test hook
  static const bool force_wide = getenv("LIDARHIP_FORCE_WIDE") != nullptr;
  bool wide = force_wide;
  int tb = 0, rb = 0;
  for (int i = 0; i < n_items; ++i) {
    lt_scene* s = it[i].s;
    lt_rayset* r = it[i].r;
    const int n = s->n_faces, R = r->n_rays;
    if (R <= 0) continue;  // nothing to write for this scan
    LT_CHECK(sc_reserve(s, n, R, stream));
    sc_job& J = B.job[B.n++];
    J.verts = s->verts; J.faces = s->faces; J.colors = s->colors; J.rem = s->rem;
    J.P = r->prm_host; J.grid = r->grid; J.sdirs = r->sdirs; J.dirs = r->dirs;
    J.cell = s->sc_cell; J.large = s->sc_large; J.large_count = s->sc_large_count; J.slices = s->sc_slices;
    J.flags = s->flags; J.counters = s->counters;
    J.endpoints = it[i].endpoints; J.endcolors = it[i].endcolors; J.range = it[i].range; J.endrem = it[i].endrem;
    J.tri = it[i].tri;
    J.ox = it[i].origin[0]; J.oy = it[i].origin[1]; J.oz = it[i].origin[2];
    J.n_verts = s->n_verts; J.n_faces = n; J.n_rays = R; J.cap_slices = s->sc_cap_queue;
    J.out_flags = flags;
    B.tris_block0[B.n - 1] = tb;
    B.resolve_block0[B.n - 1] = rb;
    tb += (n + LT_SC_T - 1) / LT_SC_T;
    rb += (R + 255) / 256;
    // 32-bit byte offsets unless an array of the launch reaches 4 GB (> 357 M triangles / vertices, > 268 M rays)
    wide = wide || (size_t)n * 12 >= (1ull << 32) || (size_t)s->n_verts * 12 >= (1ull << 32) ||
           (size_t)R * 16 >= (1ull << 32);
  }
  if (B.n == 0) return LT_OK;
  B.tris_blocks = tb;
  B.resolve_blocks = rb;
  static const int env_cap = []() {
    const char* e = getenv("LIDARHIP_SC_CAP");
    return e ? atoi(e) : 0;
  }();
  B.cap = env_cap > 0 ? env_cap : (n_items > 1 ? LT_SC_CAP_BATCH : LT_SC_CAP_SINGLE);
  static const int env_rest = []() {
    const char* e = getenv("LIDARHIP_SC_REST_BLOCKS");
    return e ? atoi(e) : 0;
  }();
  B.rest_blocks = env_rest > 0 ? env_rest : (n_items > 1 ? LT_SC_REST_BLOCKS_BATCH : LT_SC_REST_BLOCKS);
  const dim3 b(256);
#define SC_LAUNCH(KERNEL, GRID) \
  do { \
    if (count && wide) hipLaunchKernelGGL((KERNEL<true, true>), GRID, b, 0, stream, B); \
    else if (count) hipLaunchKernelGGL((KERNEL<true, false>), GRID, b, 0, stream, B); \
    else if (wide) hipLaunchKernelGGL((KERNEL<false, true>), GRID, b, 0, stream, B); \
    else hipLaunchKernelGGL((KERNEL<false, false>), GRID, b, 0, stream, B); \
  } while (0)
  if (tb > 0) {
    if (probe && probe->probe[0]) LT_HIP(hipEventRecord(probe->probe[0], stream));
    SC_LAUNCH(k_sc_tris, dim3(tb));
    if (probe) {
      if (probe->probe[1]) LT_HIP(hipEventRecord(probe->probe[1], stream));
      probe->probe[0] = probe->probe[1] = nullptr;
    }
    SC_LAUNCH(k_sc_rest, dim3(B.n * B.rest_blocks));
  }
#undef SC_LAUNCH
  if (count) hipLaunchKernelGGL(k_sc_resolve<true>, dim3(rb), b, 0, stream, B);
  else hipLaunchKernelGGL(k_sc_resolve<false>, dim3(rb), b, 0, stream, B);
  LT_HIP(hipGetLastError());
  return LT_OK;
}

// Render the scene's CURRENT mesh with the scatter strategy (no BVH needed).
extern "C" int lt_scene_render_dev(lt_scene* s, lt_rayset* r, const float* origin, float* endpoints, int* endcolors,
                                   float* range, float* endrem, int* tri, unsigned flags, void* stream_,
                                   lt_stats* stats) {
  if (!s || !r || !origin) {
    lt_set_error("lt_scene_render_dev: NULL scene / rayset / origin");
    return LT_ERR_INVALID_ARG;
  }
  if (s->device != r->device) {
    lt_set_error("lt_scene_render_dev: scene on device %d, rayset on device %d", s->device, r->device);
    return LT_ERR_INVALID_ARG;
  }
  hipStream_t stream = (hipStream_t)stream_;
  LT_HIP(hipSetDevice(s->device));
  s->last_stream = stream;
  const int n = s->n_faces, R = r->n_rays;
  const bool timed = stats != nullptr, count = (flags & LT_TRACE_COUNT) != 0;
  s->stats.n_rays = R;
  s->stats.n_faces = n;
  if (R > 0) {
    if (count) LT_HIP(hipMemsetAsync(s->counters, 0, 4 * sizeof(unsigned long long), stream));
    if (timed) LT_HIP(hipEventRecord(s->ev[7], stream));
    const sc_item it = {s, r, origin, endpoints, endcolors, range, endrem, tri};
    LT_CHECK(sc_launch_batch(&it, 1, flags, stream, count, count ? nullptr : s));
    if (timed) LT_HIP(hipEventRecord(s->ev[8], stream));
  }
  if (timed || count) {
    LT_HIP(hipStreamSynchronize(stream));
    if (timed && R > 0) LT_HIP(hipEventElapsedTime(&s->stats.ms_trace, s->ev[7], s->ev[8]));
    if (count && R > 0) {
      unsigned long long c[4];
      LT_HIP(hipMemcpy(c, s->counters, sizeof(c), hipMemcpyDeviceToHost));
      s->stats.nodes_visited = c[0];  // candidate bins visited
      s->stats.tris_tested = c[1];    // Moller-Trumbore evaluations
      s->stats.n_hits = (int)c[2];
      s->stats.stack_overflows = 0;
      s->stats.entries_culled = 0;
    }
    if (stats) *stats = s->stats;
  }
  return LT_OK;
}

// The same for up to LT_SC_MAX_BATCH scans at once: three launches for all of them instead of three each.
extern "C" int lt_scene_render_batch_dev(int n_scans, lt_scene* const* scenes, lt_rayset* const* raysets,
                                         const float* origins, float* const* endpoints, int* const* endcolors,
                                         float* const* range, float* const* endrem, int* const* tri, unsigned flags,
                                         void* stream_) {
  if (n_scans < 0 || n_scans > LT_SC_MAX_BATCH || (n_scans > 0 && (!scenes || !raysets || !origins))) {
    lt_set_error("lt_scene_render_batch_dev: invalid argument (n_scans=%d, at most %d per call)", n_scans,
                 LT_SC_MAX_BATCH);
    return LT_ERR_INVALID_ARG;
  }
  if (flags & LT_TRACE_COUNT) {
    lt_set_error("lt_scene_render_batch_dev: LT_TRACE_COUNT is a single-scan diagnostic (lt_scene_render_dev)");
    return LT_ERR_INVALID_ARG;
  }
  if (n_scans == 0) return LT_OK;
  hipStream_t stream = (hipStream_t)stream_;
  sc_item it[LT_SC_MAX_BATCH];
  lt_scene* probe = nullptr;
  for (int i = 0; i < n_scans; ++i) {
    if (!scenes[i] || !raysets[i] || scenes[i]->device != raysets[i]->device ||
        scenes[i]->device != scenes[0]->device) {
      lt_set_error("lt_scene_render_batch_dev: scan %d: NULL scene / rayset, or not all on one device", i);
      return LT_ERR_INVALID_ARG;
    }
    for (int k = 0; k < i; ++k)
      if (scenes[k] == scenes[i]) {
        lt_set_error("lt_scene_render_batch_dev: scans %d and %d are the same scene (a scene renders one scan at a time)", k, i);
        return LT_ERR_INVALID_ARG;
      }
    it[i] = {scenes[i], raysets[i], origins + 3 * i, endpoints ? endpoints[i] : nullptr,
             endcolors ? endcolors[i] : nullptr, range ? range[i] : nullptr, endrem ? endrem[i] : nullptr,
             tri ? tri[i] : nullptr};
    scenes[i]->last_stream = stream;
    scenes[i]->stats.n_rays = raysets[i]->n_rays;
    scenes[i]->stats.n_faces = scenes[i]->n_faces;
    if (!probe && scenes[i]->probe[0]) probe = scenes[i];
  }
  LT_HIP(hipSetDevice(scenes[0]->device));
  return sc_launch_batch(it, n_scans, flags, stream, false, probe ? probe : scenes[0]);
}

// debug helpers (not part of the documented ABI): queue lengths of the last LT_TRACE_COUNT render (big triangles,
// slices), and the face indices of its big-triangle queue
extern "C" int lt_debug_scatter_queues(lt_scene* s, int* out2) {
  if (!s || !out2) return LT_ERR_INVALID_ARG;
  LT_HIP(hipSetDevice(s->device));
  LT_HIP(hipDeviceSynchronize());
  unsigned long long c = 0;
  LT_HIP(hipMemcpy(&c, s->counters + 3, sizeof(c), hipMemcpyDeviceToHost));
  out2[0] = (int)(c & 0xFFFFFFFFull);
  out2[1] = (int)(c >> 32);
  return LT_OK;
}
extern "C" int lt_debug_scatter_large(lt_scene* s, int* out, int n) {
  if (!s || !out || n < 0 || n > s->sc_cap_queue) return LT_ERR_INVALID_ARG;
  LT_HIP(hipSetDevice(s->device));
  LT_HIP(hipDeviceSynchronize());
  LT_HIP(hipMemcpy(out, s->sc_large, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
  return LT_OK;
}
