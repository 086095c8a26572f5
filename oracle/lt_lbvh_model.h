/*
 * lt_lbvh_model.h -- CPU model of the HIP path's own search structure
 * (TEST INFRASTRUCTURE, included only by oracle/lt_oracle.c).
 *
 * Mirrors lidar_transfer_amd/csrc step by step (DESIGN.md "LBVH"):
 *   1. scene bounds over all vertices; cubic 10-bit/axis Morton code of each
 *      triangle centroid ((v0+v1)+v2)/3;
 *   2. stable sort by code; unique 64-bit key = code<<32 | sorted position;
 *   3. per sorted slot: triangle record (v0, e1, e2, face) and padded box;
 *      min/max segment tree over the padded boxes;
 *   4. Karras radix-tree topology: internal node i finds its key range and
 *      split by binary search; children covering <= 4 slots become leaf
 *      references; each node stores BOTH child boxes (range queries on the
 *      segment tree) -> no inter-thread communication anywhere in the build;
 *   5. near-first traversal with a small per-ray stack of node references.
 *
 * The triangle test is the reference's (Triangle.h:27-50).  The result is
 * defined as the minimum over (t, face index) of all accepted triangles, so it
 * does not depend on the tree: identical to LTO_MODE_BRUTE whenever the padded
 * boxes are conservative.
 */
#ifndef LT_LBVH_MODEL_H
#define LT_LBVH_MODEL_H

#define LBVH_LEAF 4

typedef struct { v3 mn, mx; } lbox_t;
typedef struct { lbox_t b0, b1; int c0, c1; } lnode_t; /* child ref >= 0: node; < 0: ~(start | (cnt-1)<<28) */

typedef struct {
  lnode_t* nodes;    /* [max(n-1,1)] Karras numbering, root = 0 */
  tri_t* sorted;     /* [n] triangles in Morton order */
  lbox_t* seg;       /* [2*np] segment tree, leaves at seg[np + p] */
  uint64_t* keys;    /* [n] unique sorted keys */
  int n, np, n_nodes, n_leaves;
} lbvh_t;

static inline uint32_t lbvh_expand10(uint32_t v) {
  v &= 0x3ffu;
  v = (v | (v << 16)) & 0x030000FFu;
  v = (v | (v << 8)) & 0x0300F00Fu;
  v = (v | (v << 4)) & 0x030C30C3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}

typedef struct { uint32_t key; uint32_t idx; } lkey_t;

static int lkey_cmp(const void* a, const void* b) {
  const lkey_t* x = (const lkey_t*)a;
  const lkey_t* y = (const lkey_t*)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return x->idx < y->idx ? -1 : (x->idx > y->idx);
}

static void lbvh_free(lbvh_t* L) {
  free(L->nodes); free(L->sorted); free(L->seg); free(L->keys);
  L->nodes = NULL; L->sorted = NULL; L->seg = NULL; L->keys = NULL;
}

static inline lbox_t lbox_empty(void) {
  lbox_t b = {{INFINITY, INFINITY, INFINITY}, {-INFINITY, -INFINITY, -INFINITY}};
  return b;
}
static inline lbox_t lbox_union(lbox_t a, lbox_t c) {
  lbox_t b;
  b.mn.x = fminf(a.mn.x, c.mn.x); b.mn.y = fminf(a.mn.y, c.mn.y); b.mn.z = fminf(a.mn.z, c.mn.z);
  b.mx.x = fmaxf(a.mx.x, c.mx.x); b.mx.y = fmaxf(a.mx.y, c.mx.y); b.mx.z = fmaxf(a.mx.z, c.mx.z);
  return b;
}

/* union of the padded boxes of sorted slots [a, b] (inclusive) */
static inline lbox_t lbvh_range_box(const lbvh_t* L, int a, int b) {
  lbox_t acc = lbox_empty();
  int l = a + L->np, r = b + L->np + 1;
  while (l < r) {
    if (l & 1) acc = lbox_union(acc, L->seg[l++]);
    if (r & 1) acc = lbox_union(acc, L->seg[--r]);
    l >>= 1; r >>= 1;
  }
  return acc;
}

static inline int lbvh_delta(const lbvh_t* L, int i, int j) {
  if (j < 0 || j >= L->n) return -1;
  return __builtin_clzll(L->keys[i] ^ L->keys[j]);
}

static inline int lbvh_leaf_ref(int start, int cnt) { return ~(start | ((cnt - 1) << 28)); }

/* pad = a_pad + r_pad * max|coordinate|  (same formula as the HIP build) */
static int lbvh_build(lbvh_t* L, const tri_t* prims, int n, float pad_unused) {
  (void)pad_unused;
  L->n = n;
  v3 lo = {INFINITY, INFINITY, INFINITY}, hi = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = 0; i < n; ++i) {
    const v3* vs[3] = {&prims[i].v0, &prims[i].v1, &prims[i].v2};
    for (int k = 0; k < 3; ++k) {
      lo.x = fminf(lo.x, vs[k]->x); lo.y = fminf(lo.y, vs[k]->y); lo.z = fminf(lo.z, vs[k]->z);
      hi.x = fmaxf(hi.x, vs[k]->x); hi.y = fmaxf(hi.y, vs[k]->y); hi.z = fmaxf(hi.z, vs[k]->z);
    }
  }
  float ext = fmaxf(fmaxf(hi.x - lo.x, hi.y - lo.y), hi.z - lo.z);
  float scale = ext > 0.0f ? 1024.0f / ext : 0.0f;
  float maxabs = fmaxf(fmaxf(fmaxf(fabsf(lo.x), fabsf(hi.x)), fmaxf(fabsf(lo.y), fabsf(hi.y))),
                       fmaxf(fabsf(lo.z), fabsf(hi.z)));
  float pad = 1e-4f + 2e-6f * maxabs;
  lkey_t* keys = (lkey_t*)malloc(sizeof(lkey_t) * (size_t)n);
  if (!keys) return -1;
  for (int i = 0; i < n; ++i) {
    v3 c = tri_centroid(&prims[i]);
    float fx = fminf(fmaxf((c.x - lo.x) * scale, 0.0f), 1023.0f);
    float fy = fminf(fmaxf((c.y - lo.y) * scale, 0.0f), 1023.0f);
    float fz = fminf(fmaxf((c.z - lo.z) * scale, 0.0f), 1023.0f);
    uint32_t qx = (uint32_t)fx, qy = (uint32_t)fy, qz = (uint32_t)fz;
    keys[i].key = (lbvh_expand10(qx) << 2) | (lbvh_expand10(qy) << 1) | lbvh_expand10(qz);
    keys[i].idx = (uint32_t)i;
  }
  qsort(keys, (size_t)n, sizeof(lkey_t), lkey_cmp);
  int np = 1;
  while (np < n) np <<= 1;
  L->np = np;
  L->sorted = (tri_t*)malloc(sizeof(tri_t) * (size_t)n);
  L->keys = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)n);
  L->seg = (lbox_t*)malloc(sizeof(lbox_t) * (size_t)np * 2);
  L->nodes = (lnode_t*)malloc(sizeof(lnode_t) * (size_t)(n > 1 ? n - 1 : 1));
  if (!L->sorted || !L->keys || !L->seg || !L->nodes) { free(keys); return -1; }
  for (int p = 0; p < np; ++p) L->seg[np + p] = lbox_empty();
  for (int p = 0; p < n; ++p) {
    L->sorted[p] = prims[keys[p].idx];
    L->keys[p] = ((uint64_t)keys[p].key << 32) | (uint32_t)p;
    v3 mn, mx;
    tri_bbox(&L->sorted[p], &mn, &mx);
    lbox_t b = {{mn.x - pad, mn.y - pad, mn.z - pad}, {mx.x + pad, mx.y + pad, mx.z + pad}};
    L->seg[np + p] = b;
  }
  free(keys);
  for (int k = np - 1; k >= 1; --k) L->seg[k] = lbox_union(L->seg[2 * k], L->seg[2 * k + 1]);
  L->n_nodes = 0;
  L->n_leaves = 0;
  if (n == 1) {
    lnode_t* N = &L->nodes[0];
    N->c0 = lbvh_leaf_ref(0, 1); N->b0 = L->seg[np];
    N->c1 = lbvh_leaf_ref(0, 1);
    N->b1.mn.x = N->b1.mn.y = N->b1.mn.z = N->b1.mx.x = N->b1.mx.y = N->b1.mx.z = INFINITY; /* never hit */
    L->n_nodes = 1; L->n_leaves = 1;
    return 0;
  }
  for (int i = 0; i < n - 1; ++i) {
    int d = (lbvh_delta(L, i, i + 1) - lbvh_delta(L, i, i - 1)) >= 0 ? 1 : -1;
    int dmin = lbvh_delta(L, i, i - d);
    int lmax = 2;
    while (lbvh_delta(L, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
      if (lbvh_delta(L, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dn = lbvh_delta(L, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
      if (lbvh_delta(L, i, i + (s + t) * d) > dn) s += t;
      if (t == 1) break;
    }
    int g = i + s * d + (d < 0 ? -1 : 0);
    int first = i < j ? i : j, last = i < j ? j : i;
    lnode_t* N = &L->nodes[i];
    if (last - first + 1 <= LBVH_LEAF && i != 0) { N->c0 = N->c1 = 0x7fffffff; continue; } /* never referenced */
    int lc = g - first + 1, rc = last - g;
    N->c0 = (lc <= LBVH_LEAF) ? lbvh_leaf_ref(first, lc) : g;
    N->c1 = (rc <= LBVH_LEAF) ? lbvh_leaf_ref(g + 1, rc) : g + 1;
    N->b0 = lbvh_range_box(L, first, g);
    N->b1 = lbvh_range_box(L, g + 1, last);
    L->n_nodes++;
    L->n_leaves += (lc <= LBVH_LEAF) + (rc <= LBVH_LEAF);
  }
  return 0;
}

/* slab test, NaN-safe (fminf/fmaxf drop the NaN of 0*inf); returns tnear, or
 * +inf when the box is missed or lies beyond best_t */
static inline float lbvh_box(const lbox_t* b, v3 o, v3 inv, float best_t) {
  float l1x = (b->mn.x - o.x) * inv.x, l2x = (b->mx.x - o.x) * inv.x;
  float l1y = (b->mn.y - o.y) * inv.y, l2y = (b->mx.y - o.y) * inv.y;
  float l1z = (b->mn.z - o.z) * inv.z, l2z = (b->mx.z - o.z) * inv.z;
  float tn = fmaxf(fmaxf(fminf(l1x, l2x), fminf(l1y, l2y)), fmaxf(fminf(l1z, l2z), 0.0f));
  float tf = fminf(fminf(fmaxf(l1x, l2x), fmaxf(l1y, l2y)), fminf(fmaxf(l1z, l2z), best_t));
  return (tn <= tf) ? tn : INFINITY;
}

static inline int lbvh_intersect(const lbvh_t* L, v3 o, v3 d, v3 inv, float* t_out, cnt_t* C) {
  float best_t = 999999999.f;
  int best_face = 0x7fffffff;
  int stack[64];
  int sp = 0, cur = 0;
  for (;;) {
    if (cur >= 0) {
      const lnode_t* N = &L->nodes[cur];
      C->nodes++;
      C->boxes += 2;
      float t0 = lbvh_box(&N->b0, o, inv, best_t), t1 = lbvh_box(&N->b1, o, inv, best_t);
      if (t0 != INFINITY && t1 != INFINITY) {
        int nearc = (t1 < t0) ? N->c1 : N->c0, farc = (t1 < t0) ? N->c0 : N->c1;
        stack[sp++] = farc;
        if (sp > C->max_stack) C->max_stack = sp;
        cur = nearc;
        continue;
      }
      if (t0 != INFINITY) { cur = N->c0; continue; }
      if (t1 != INFINITY) { cur = N->c1; continue; }
    } else {
      int ref = ~cur;
      int s = ref & 0x0fffffff, cnt = (ref >> 28) + 1;
      for (int p = s; p < s + cnt; ++p) {
        float t;
        C->tris++;
        if (tri_hit(&L->sorted[p], o, d, &t)) {
          int f = L->sorted[p].face;
          if (t < best_t || (t == best_t && f < best_face)) { best_t = t; best_face = f; }
        }
      }
    }
    if (sp == 0) break;
    cur = stack[--sp];
  }
  *t_out = best_t;
  return best_face == 0x7fffffff ? -1 : best_face;
}

#endif
