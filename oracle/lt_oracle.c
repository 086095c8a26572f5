/*
 * lt_oracle.c -- CPU ORACLE for the lidar_transfer ray-cast hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker.  The product path
 * (lidar_transfer_amd/) never imports, links or calls it.
 *
 * What it is: a from-scratch scalar C restatement of the reference's
 * algorithm for `ctrace` (reference: auxiliary/raytracer/RayTracer.cpp:19-124),
 * i.e. triangle set-up -> top-down midpoint-split BVH -> near-first stack
 * traversal -> hit write-back, with the reference's exact float32 operation
 * order (no FMA contraction: build with -ffp-contract=off).
 *
 * Parity pinning: tests/test_oracle_vs_ref.py checks `lto_trace(mode=REF_BVH,
 * norm=SSE)` bit-for-bit against the real reference compiled from
 * /root/reference into oracle/_ref/ (see oracle/Makefile), and against the
 * golden vectors under tests/golden/ that were generated from that build.
 *
 * Three search structures, one triangle test:
 *   LTO_MODE_REF_BVH   restatement of BVH::build / BVH::getIntersection
 *                      (BVH.cpp:143-243, :19-110), incl. its non-conservative
 *                      SSE slab test (BBox.cpp:52-100) and "first visited wins"
 *                      tie rule (strict <, BVH.cpp:59).
 *   LTO_MODE_BRUTE     every ray against every triangle; closest hit, ties on
 *                      equal t broken by the LOWER face index.  This is the
 *                      tree-independent definition the HIP path is held to.
 *   LTO_MODE_LBVH      CPU model of the HIP path's own structure (implicit
 *                      balanced BVH over Morton-sorted triangles, padded boxes,
 *                      bit-trail stackless traversal); used to debug kernels
 *                      and to count nodes/triangles per ray for the roofline.
 *
 * Two direction normalisations (Vector3.h:73-89 uses _mm_rsqrt_ps + 1 NR step,
 * whose bits are CPU-vendor specific):
 *   LTO_NORM_SSE       _mm_rsqrt_ss seed  (bit-identical to the reference on
 *                      the CPU it runs on)
 *   LTO_NORM_EXACT     correctly rounded 1/sqrtf seed, same NR step (vendor
 *                      independent; LT_TRACE_NORM_EXACT in the HIP library)
 *   LTO_NORM_SSE_TABLE the RSQRTSS seed replayed from a 2x1024 table measured on an
 *                      Intel CPU and verified exhaustively (oracle/gen_rsqrt_table.c);
 *                      equals LTO_NORM_SSE on Intel hosts, and is what the HIP
 *                      library computes by default, so that its images equal the
 *                      reference's as run on the machine the goldens came from.
 *   LTO_NORM_AMD_TABLE the same for an AMD host (2x4096 table measured on an EPYC 9575F,
 *                      the MI355X box's CPU; LT_TRACE_NORM_AMD in the HIP library);
 *                      equals LTO_NORM_SSE on AMD Zen 5 hosts.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <xmmintrin.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define LTO_MODE_REF_BVH 0
#define LTO_MODE_BRUTE 1
#define LTO_MODE_LBVH 2

#define LTO_NORM_SSE 0
#define LTO_NORM_EXACT 1
#define LTO_NORM_SSE_TABLE 2 /* RSQRTSS emulated from lidar_transfer_amd/csrc/lt_rsqrt_sse_table.h (measured on an Intel CPU) */

#define LTO_NORM_AMD_TABLE 3 /* ... from lidar_transfer_amd/csrc/lt_rsqrt_amd_table.h (measured on the GPU box's AMD EPYC host) */
/* ONE copy of each generated table: the product's (the Makefile adds -I../lidar_transfer_amd/csrc) */
#include "lt_rsqrt_sse_table.h"
#include "lt_rsqrt_amd_table.h"

typedef struct {
  double t_setup_ms, t_build_ms, t_trace_ms;
  long long nodes_popped;   /* nodes popped (REF_BVH) / visited (LBVH)      */
  long long tris_tested;    /* Moller-Trumbore evaluations                   */
  long long box_tests;      /* slab tests                                    */
  int n_nodes, n_leaves, max_stack;
  int n_hits;
} lto_stats;

typedef struct { float x, y, z; } v3;

/* ---- timing ------------------------------------------------------------ */
#include <time.h>
static double now_ms(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

/* ---- Vector3.h arithmetic, scalarised ---------------------------------- */
/* _mm_min_ps / _mm_max_ps semantics: (a < b) ? a : b  -- second operand on NaN */
static inline float sse_min(float a, float b) { return (a < b) ? a : b; }
static inline float sse_max(float a, float b) { return (a > b) ? a : b; }
static inline v3 v3sub(v3 a, v3 b) { v3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
/* dot: x*bx + y*by + z*bz, left to right (Vector3.h:32-34) */
static inline float v3dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
/* cross (Vector3.h:37-46) */
static inline v3 v3cross(v3 a, v3 b) {
  v3 r = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
  return r;
}

/* RSQRTSS replayed from a measured table of 2 x 2^bits entries; D is a sum of squares (never negative) */
static inline float lto_rsqrt_table(const unsigned int* table, int bits, float x) {
  uint32_t b;
  memcpy(&b, &x, 4);
  const int e = (int)((b >> 23) & 255u);
  if (e == 0) return INFINITY;               /* zero / denormal source: treated as zero */
  if (e == 255) return (b & 0x7fffffu) ? x : 0.0f; /* NaN -> NaN, inf -> 0 */
  const int p = (e - 127) & 1;
  const int k = (e - 127 - p) / 2;
  const uint32_t t = table[(p << bits) + ((b >> (23 - bits)) & ((1u << bits) - 1u))] - ((uint32_t)k << 23);
  float r;
  memcpy(&r, &t, 4);
  return r;
}

/* normalize (Vector3.h:73-89): D = (x^2+y^2)+(z^2+0); r = 1.5 r0 + ((D*-0.5)*r0)*(r0*r0) */
static inline v3 lto_normalize(v3 a, int norm_mode) {
  float D = (a.x * a.x + a.y * a.y) + (a.z * a.z + 0.0f);
  float r0;
  if (norm_mode == LTO_NORM_SSE) {
    r0 = _mm_cvtss_f32(_mm_rsqrt_ss(_mm_set_ss(D)));
  } else if (norm_mode == LTO_NORM_SSE_TABLE) {
    r0 = lto_rsqrt_table(LT_RSQRT_SSE_TABLE, LT_RSQRT_SSE_TABLE_BITS, D);
  } else if (norm_mode == LTO_NORM_AMD_TABLE) {
    r0 = lto_rsqrt_table(LT_RSQRT_AMD_TABLE, LT_RSQRT_AMD_TABLE_BITS, D);
  } else {
    r0 = 1.0f / sqrtf(D);
  }
  float r = (1.5f * r0) + (((D * -0.5f) * r0) * (r0 * r0));
  v3 o = {a.x * r, a.y * r, a.z * r};
  return o;
}

/* ---- triangles ---------------------------------------------------------- */
typedef struct {
  v3 v0, v1, v2;
  int face;          /* original face index */
} tri_t;

/* Triangle::getIntersection (Triangle.h:27-50); returns 1 and *t on hit */
static inline int tri_hit(const tri_t* T, v3 o, v3 d, float* t_out) {
  v3 e1 = v3sub(T->v1, T->v0);
  v3 e2 = v3sub(T->v2, T->v0);
  v3 h = v3cross(d, e2);
  float a = v3dot(e1, h);
  const float eps = 0.000001f;
  if (a < eps && a > -eps) return 0;
  float inv_a = 1.0f / a;
  v3 s = v3sub(o, T->v0);
  float u = v3dot(s, h) * inv_a;
  if (u < 0 || u > 1) return 0;
  v3 q = v3cross(s, e1);
  float v = v3dot(d, q) * inv_a;
  if (v < 0 || u + v > 1) return 0;
  float t = v3dot(e2, q) * inv_a;
  if (t < eps) return 0;
  *t_out = t;
  return 1;
}

/* ---- reference BVH (BVH.h:12-15, BVH.cpp:143-243) ----------------------- */
typedef struct {
  v3 bmin, bmax;
  uint32_t start, nPrims, rightOffset;
} rnode_t;

typedef struct {
  rnode_t* nodes;
  uint32_t n_nodes, n_leaves;
  tri_t* prims; /* reordered in place by the build */
  uint32_t n_prims;
} rbvh_t;

static inline v3 tri_centroid(const tri_t* T) {
  /* (v0 + v1 + v2) / 3  (Triangle.h:78-80) */
  v3 c = {((T->v0.x + T->v1.x) + T->v2.x) / 3.0f, ((T->v0.y + T->v1.y) + T->v2.y) / 3.0f,
          ((T->v0.z + T->v1.z) + T->v2.z) / 3.0f};
  return c;
}
static inline void tri_bbox(const tri_t* T, v3* mn, v3* mx) {
  /* min(v0, min(v1, v2)) (Triangle.h:72-76) */
  mn->x = sse_min(T->v0.x, sse_min(T->v1.x, T->v2.x));
  mn->y = sse_min(T->v0.y, sse_min(T->v1.y, T->v2.y));
  mn->z = sse_min(T->v0.z, sse_min(T->v1.z, T->v2.z));
  mx->x = sse_max(T->v0.x, sse_max(T->v1.x, T->v2.x));
  mx->y = sse_max(T->v0.y, sse_max(T->v1.y, T->v2.y));
  mx->z = sse_max(T->v0.z, sse_max(T->v1.z, T->v2.z));
}

typedef struct { uint32_t parent, start, end; } build_entry_t;

static int rbvh_build(rbvh_t* B, uint32_t leafSize) {
  const uint32_t Untouched = 0xffffffffu, TouchedTwice = 0xfffffffdu;
  build_entry_t todo[128];
  uint32_t stackptr = 0;
  uint32_t cap = B->n_prims * 2 + 2;
  rnode_t* bn = (rnode_t*)malloc(sizeof(rnode_t) * (size_t)cap);
  v3* cent = (v3*)malloc(sizeof(v3) * (size_t)(B->n_prims ? B->n_prims : 1));
  if (!bn || !cent) { free(bn); free(cent); return -1; }
  /* centroids are pure functions of the triangle; cache them, swap alongside */
  for (uint32_t i = 0; i < B->n_prims; ++i) cent[i] = tri_centroid(&B->prims[i]);
  uint32_t nNodes = 0, nLeafs = 0;
  todo[stackptr].start = 0;
  todo[stackptr].end = B->n_prims;
  todo[stackptr].parent = 0xfffffffcu;
  stackptr++;
  while (stackptr > 0) {
    build_entry_t bnode = todo[--stackptr];
    uint32_t start = bnode.start, end = bnode.end, nPrims = end - start;
    nNodes++;
    rnode_t node;
    node.start = start;
    node.nPrims = nPrims;
    node.rightOffset = Untouched;
    v3 bbmin, bbmax, bcmin, bcmax;
    tri_bbox(&B->prims[start], &bbmin, &bbmax);
    bcmin = bcmax = cent[start];
    for (uint32_t p = start + 1; p < end; ++p) {
      v3 mn, mx;
      tri_bbox(&B->prims[p], &mn, &mx);
      /* expandToInclude: min = ::min(min, b.min) (BBox.cpp:16-20) */
      bbmin.x = sse_min(bbmin.x, mn.x); bbmin.y = sse_min(bbmin.y, mn.y); bbmin.z = sse_min(bbmin.z, mn.z);
      bbmax.x = sse_max(bbmax.x, mx.x); bbmax.y = sse_max(bbmax.y, mx.y); bbmax.z = sse_max(bbmax.z, mx.z);
      v3 c = cent[p];
      bcmin.x = sse_min(bcmin.x, c.x); bcmin.y = sse_min(bcmin.y, c.y); bcmin.z = sse_min(bcmin.z, c.z);
      bcmax.x = sse_max(bcmax.x, c.x); bcmax.y = sse_max(bcmax.y, c.y); bcmax.z = sse_max(bcmax.z, c.z);
    }
    node.bmin = bbmin;
    node.bmax = bbmax;
    if (nPrims <= leafSize) { node.rightOffset = 0; nLeafs++; }
    if (nNodes > cap) { free(bn); free(cent); return -2; }
    bn[nNodes - 1] = node;
    if (bnode.parent != 0xfffffffcu) {
      bn[bnode.parent].rightOffset--;
      if (bn[bnode.parent].rightOffset == TouchedTwice)
        bn[bnode.parent].rightOffset = nNodes - 1 - bnode.parent;
    }
    if (node.rightOffset == 0) continue;
    /* maxDimension on extent = max - min (BBox.cpp:22-30) */
    float ex = bcmax.x - bcmin.x, ey = bcmax.y - bcmin.y, ez = bcmax.z - bcmin.z;
    uint32_t split_dim = 0;
    if (ey > ex) { split_dim = 1; if (ez > ey) split_dim = 2; }
    else if (ez > ex) split_dim = 2;
    float lo = split_dim == 0 ? bcmin.x : (split_dim == 1 ? bcmin.y : bcmin.z);
    float hi = split_dim == 0 ? bcmax.x : (split_dim == 1 ? bcmax.y : bcmax.z);
    float split_coord = .5f * (lo + hi);
    uint32_t mid = start;
    for (uint32_t i = start; i < end; ++i) {
      float c = split_dim == 0 ? cent[i].x : (split_dim == 1 ? cent[i].y : cent[i].z);
      if (c < split_coord) {
        tri_t tt = B->prims[i]; B->prims[i] = B->prims[mid]; B->prims[mid] = tt;
        v3 tc = cent[i]; cent[i] = cent[mid]; cent[mid] = tc;
        ++mid;
      }
    }
    if (mid == start || mid == end) mid = start + (end - start) / 2;
    if (stackptr + 2 > 128) { free(bn); free(cent); return -3; } /* reference has no guard */
    todo[stackptr].start = mid; todo[stackptr].end = end; todo[stackptr].parent = nNodes - 1; stackptr++;
    todo[stackptr].start = start; todo[stackptr].end = mid; todo[stackptr].parent = nNodes - 1; stackptr++;
  }
  B->nodes = (rnode_t*)malloc(sizeof(rnode_t) * (size_t)(nNodes ? nNodes : 1));
  memcpy(B->nodes, bn, sizeof(rnode_t) * (size_t)nNodes);
  B->n_nodes = nNodes;
  B->n_leaves = nLeafs;
  free(bn);
  free(cent);
  return 0;
}

/* BBox::intersect (BBox.cpp:52-100), lanes x,y,z only (the w lane is masked
 * out by the horizontal min/max there). */
static inline int ref_box_hit(const rnode_t* n, v3 o, v3 inv, float* tnear, float* tfar) {
  const float pinf = INFINITY, ninf = -INFINITY;
  float l1x = (n->bmin.x - o.x) * inv.x, l2x = (n->bmax.x - o.x) * inv.x;
  float l1y = (n->bmin.y - o.y) * inv.y, l2y = (n->bmax.y - o.y) * inv.y;
  float l1z = (n->bmin.z - o.z) * inv.z, l2z = (n->bmax.z - o.z) * inv.z;
  float lmaxx = sse_max(sse_min(l1x, pinf), sse_min(l2x, pinf));
  float lmaxy = sse_max(sse_min(l1y, pinf), sse_min(l2y, pinf));
  float lmaxz = sse_max(sse_min(l1z, pinf), sse_min(l2z, pinf));
  float lminx = sse_min(sse_max(l1x, ninf), sse_max(l2x, ninf));
  float lminy = sse_min(sse_max(l1y, ninf), sse_max(l2y, ninf));
  float lminz = sse_min(sse_max(l1z, ninf), sse_max(l2z, ninf));
  float lmax = sse_min(sse_min(lmaxx, lmaxy), lmaxz);
  float lmin = sse_max(sse_max(lminx, lminy), lminz);
  *tnear = lmin;
  *tfar = lmax;
  return (lmax >= 0.0f) & (lmax >= lmin);
}

typedef struct { long long nodes, tris, boxes; int max_stack; } cnt_t;

/* BVH::getIntersection (BVH.cpp:19-110); returns prim slot or -1 */
static inline int rbvh_intersect(const rbvh_t* B, v3 o, v3 d, v3 inv, float* t_out, cnt_t* C) {
  float best_t = 999999999.f;
  int best = -1;
  struct { uint32_t i; float mint; } todo[64];
  int32_t sp = 0;
  todo[0].i = 0;
  todo[0].mint = -9999999.f;
  while (sp >= 0) {
    int ni = (int)todo[sp].i;
    float near = todo[sp].mint;
    sp--;
    const rnode_t* node = &B->nodes[ni];
    C->nodes++;
    if (near > best_t) continue;
    if (node->rightOffset == 0) {
      for (uint32_t k = 0; k < node->nPrims; ++k) {
        float t;
        C->tris++;
        if (tri_hit(&B->prims[node->start + k], o, d, &t)) {
          if (t < best_t) { best_t = t; best = (int)(node->start + k); }
        }
      }
    } else {
      float bb[4];
      C->boxes += 2;
      int h0 = ref_box_hit(&B->nodes[ni + 1], o, inv, &bb[0], &bb[1]);
      int h1 = ref_box_hit(&B->nodes[ni + node->rightOffset], o, inv, &bb[2], &bb[3]);
      if (h0 && h1) {
        int closer = ni + 1, other = ni + (int)node->rightOffset;
        if (bb[2] < bb[0]) {
          float x = bb[0]; bb[0] = bb[2]; bb[2] = x;
          x = bb[1]; bb[1] = bb[3]; bb[3] = x;
          int y = closer; closer = other; other = y;
        }
        if (sp + 2 >= 64) return -2; /* reference has no guard */
        ++sp; todo[sp].i = (uint32_t)other; todo[sp].mint = bb[2];
        ++sp; todo[sp].i = (uint32_t)closer; todo[sp].mint = bb[0];
      } else if (h0) {
        ++sp; todo[sp].i = (uint32_t)(ni + 1); todo[sp].mint = bb[0];
      } else if (h1) {
        ++sp; todo[sp].i = (uint32_t)(ni + node->rightOffset); todo[sp].mint = bb[2];
      }
      if (sp + 1 > C->max_stack) C->max_stack = sp + 1;
    }
  }
  *t_out = best_t;
  return best;
}

/* ---- LBVH model (mirrors lidar_transfer_amd/csrc) -------- */
#include "lt_lbvh_model.h"

/* ---- shared write-back (RayTracer.cpp:73-90) ---------------------------- */
static inline void write_hit(size_t ray, v3 o, v3 d, float t, int face, const int* faces,
                             const int* colors, const float* rem, float* endpoints, int* endcolors,
                             float* range, float* endrem, int* tri) {
  int i0 = faces[3 * face + 0], i1 = faces[3 * face + 1], i2 = faces[3 * face + 2];
  /* hit = o + d*t (BVH.cpp:107): mul then add, no fma */
  float dx = d.x * t, dy = d.y * t, dz = d.z * t;
  endpoints[3 * ray + 0] = o.x + dx;
  endpoints[3 * ray + 1] = o.y + dy;
  endpoints[3 * ray + 2] = o.z + dz;
  /* colour of vertex 0, int -> float -> int (RayTracer.cpp:36, :80-82) */
  endcolors[3 * ray + 0] = (int)(float)colors[3 * i0 + 0];
  endcolors[3 * ray + 1] = (int)(float)colors[3 * i0 + 1];
  endcolors[3 * ray + 2] = (int)(float)colors[3 * i0 + 2];
  /* (r0+r1+r2)/3 (Triangle.h:69) */
  endrem[ray] = ((rem[i0] + rem[i1]) + rem[i2]) / 3;
  range[ray] = t;
  if (tri) tri[ray] = face;
}

/*
 * lto_trace: same 14 parameters and layout as the reference `ctrace`
 * (RayTracer.cpp:116-124) + hit-triangle output + mode switches.
 * Outputs are written only for hits (caller pre-zeros; tri pre-set to -1 by
 * the caller if wanted).  Returns 0, or <0 on internal overflow/alloc failure.
 */
int lto_trace(const float* rays, const float* origin_in, const float* verts, const int* faces,
              const int* colors, const float* rem, int n_rays, int n_verts, int n_faces, int height,
              float* endpoints, int* endcolors, float* range, float* endrem, int* tri, int mode,
              int norm_mode, int nthreads, float lbvh_pad, lto_stats* st) {
  (void)n_verts;
  lto_stats local;
  if (!st) st = &local;
  memset(st, 0, sizeof(*st));
  if (height <= 0 || n_rays <= 0) return 0;
  double t0 = now_ms();
  tri_t* prims = (tri_t*)malloc(sizeof(tri_t) * (size_t)(n_faces > 0 ? n_faces : 1));
  if (!prims) return -1;
  for (int i = 0; i < n_faces; ++i) {
    int a = faces[3 * i + 0] * 3, b = faces[3 * i + 1] * 3, c = faces[3 * i + 2] * 3;
    prims[i].v0.x = verts[a]; prims[i].v0.y = verts[a + 1]; prims[i].v0.z = verts[a + 2];
    prims[i].v1.x = verts[b]; prims[i].v1.y = verts[b + 1]; prims[i].v1.z = verts[b + 2];
    prims[i].v2.x = verts[c]; prims[i].v2.y = verts[c + 1]; prims[i].v2.z = verts[c + 2];
    prims[i].face = i;
  }
  double t1 = now_ms();
  st->t_setup_ms = t1 - t0;

  rbvh_t B;
  memset(&B, 0, sizeof(B));
  lbvh_t L;
  memset(&L, 0, sizeof(L));
  int rc = 0;
  if (mode == LTO_MODE_REF_BVH && n_faces > 0) {
    B.prims = prims;
    B.n_prims = (uint32_t)n_faces;
    rc = rbvh_build(&B, 4);
    st->n_nodes = (int)B.n_nodes;
    st->n_leaves = (int)B.n_leaves;
  } else if (mode == LTO_MODE_LBVH && n_faces > 0) {
    rc = lbvh_build(&L, prims, n_faces, lbvh_pad);
    st->n_nodes = L.n_nodes;
    st->n_leaves = L.n_leaves;
  }
  if (rc) { free(prims); free(B.nodes); lbvh_free(&L); return rc; }
  double t2 = now_ms();
  st->t_build_ms = t2 - t1;

  const int width = n_rays / height;
  v3 origin = {origin_in[0], origin_in[1], origin_in[2]};
  long long tot_nodes = 0, tot_tris = 0, tot_boxes = 0;
  int max_stack = 0, n_hits = 0, err = 0;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#else
  (void)nthreads;
#endif
#pragma omp parallel for schedule(static) reduction(+ : tot_nodes, tot_tris, tot_boxes, n_hits) reduction(max : max_stack)
  for (int i = 0; i < width; ++i) {
    for (int j = 0; j < height; ++j) {
      size_t ray = (size_t)width * j + i;
      v3 r = {rays[3 * ray], rays[3 * ray + 1], rays[3 * ray + 2]};
      v3 d = lto_normalize(r, norm_mode);
      v3 inv = {1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
      cnt_t C = {0, 0, 0, 0};
      float t = 0;
      int face = -1;
      if (n_faces <= 0) {
        face = -1;
      } else if (mode == LTO_MODE_REF_BVH) {
        int slot = rbvh_intersect(&B, origin, d, inv, &t, &C);
        if (slot == -2) { err = 1; slot = -1; }
        face = slot >= 0 ? B.prims[slot].face : -1;
      } else if (mode == LTO_MODE_BRUTE) {
        float best = 999999999.f;
        for (int k = 0; k < n_faces; ++k) {
          float tt;
          if (tri_hit(&prims[k], origin, d, &tt) && tt < best) { best = tt; face = k; }
        }
        C.tris = n_faces;
        t = best;
      } else {
        face = lbvh_intersect(&L, origin, d, inv, &t, &C);
      }
      tot_nodes += C.nodes; tot_tris += C.tris; tot_boxes += C.boxes;
      if (C.max_stack > max_stack) max_stack = C.max_stack;
      if (face >= 0) {
        n_hits++;
        write_hit(ray, origin, d, t, face, faces, colors, rem, endpoints, endcolors, range, endrem, tri);
      }
    }
  }
  double t3 = now_ms();
  st->t_trace_ms = t3 - t2;
  st->nodes_popped = tot_nodes;
  st->tris_tested = tot_tris;
  st->box_tests = tot_boxes;
  st->max_stack = max_stack;
  st->n_hits = n_hits;
  free(prims);
  free(B.nodes);
  lbvh_free(&L);
  return err ? -4 : 0;
}

/* normalised direction only (for unit tests of the two normalisations) */
void lto_normalize_rays(const float* rays, int n_rays, int norm_mode, float* out) {
  for (int i = 0; i < n_rays; ++i) {
    v3 r = {rays[3 * i], rays[3 * i + 1], rays[3 * i + 2]};
    v3 d = lto_normalize(r, norm_mode);
    out[3 * i] = d.x; out[3 * i + 1] = d.y; out[3 * i + 2] = d.z;
  }
}

int lto_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
