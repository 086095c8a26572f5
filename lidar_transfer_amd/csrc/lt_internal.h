// lt_internal.h -- shared declarations of liblidarhip.so (gfx950 only; no portability layer).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "lidarhip.h"

// ---- error plumbing ---------------------------------------------------------------------------
void lt_set_error(const char* fmt, ...);
// compute units of a device, cached per device index (a process may hold volumes / meshes on several GPUs)
int lt_cu_count(int device);

#define LT_HIP(call)                                                                          \
  do {                                                                                        \
    hipError_t e__ = (call);                                                                  \
    if (e__ != hipSuccess) {                                                                  \
      lt_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
      return e__ == hipErrorOutOfMemory ? LT_ERR_NO_MEMORY : LT_ERR_HIP;                      \
    }                                                                                         \
  } while (0)

#define LT_CHECK(expr)              \
  do {                              \
    int rc__ = (expr);              \
    if (rc__ != LT_OK) return rc__; \
  } while (0)

// ---- device data layout -----------------------------------------------------------------------
// Sorted triangle record, 48 B = 3 x float4 (one 16-B-aligned dwordx4 load each):
//   q0 = (v0.x, v0.y, v0.z, e1.x)   q1 = (e1.y, e1.z, e2.x, e2.y)   q2 = (e2.z, face, -, -)
// e1 = v1 - v0, e2 = v2 - v0 are the first two operations of Triangle::getIntersection
// (Triangle.h:28-29) hoisted into the build; float subtraction is deterministic, so the result
// bits are those of the reference.
//
// Segment-tree box, 32 B = 2 x float4: lo = (mn.xyz, 0), hi = (mx.xyz, 0); heap layout, leaves at
// [np, 2np), node k = union(2k, 2k+1).
//
// BVH node, 64 B = 4 x float4, Karras numbering (node i splits the sorted key range it owns):
//   q0 = (b0.mn.x, b0.mn.y, b0.mn.z, b0.mx.x)   q1 = (b0.mx.y, b0.mx.z, b1.mn.x, b1.mn.y)
//   q2 = (b1.mn.z, b1.mx.x, b1.mx.y, b1.mx.z)   q3 = (c0, c1, -, -) as int bits
// child reference c >= 0: node index; c < 0: leaf, ~c = start | (count-1) << 28.

#define LT_LEAF_MAX 4
#ifndef LT_SORT_TILE
#define LT_SORT_TILE 4096      // keys per workgroup per radix pass (256 threads x 16)
#endif
#define LT_SORT_THREADS 256
#define LT_SEG_SUB 512         // segment-tree leaves reduced per workgroup
#define LT_STACK_LDS 32        // per-ray stack entries kept in LDS; deeper entries spill to HBM
#define LT_STACK_MAX 64        // Karras depth bound for 62-bit unique keys
#define LT_STACK4_LDS 40       // quad traversal (4-wide nodes): stack entries per ray kept in LDS
#define LT_STACK4_MAX 104      // 40 + 64 spill entries: <= 3 pushes per level, <= 31 levels of a collapsed Karras tree
#define LT_TAIL_SAVE (LT_STACK4_MAX + 1)  // a handed-over ray: its stack + the reference it was about to visit
#define LT_TAIL_STACK 512      // k_trace4_tail: work stack of a wave (LDS); above LT_TAIL_DFS entries one quad pops at a time
#define LT_TAIL_DFS 256
#define LT_TRACE4_STEP_CAP 40  // node + leaf steps a ray may take in k_trace4 before it is handed over (LIDARHIP_STEP_CAP)

struct lt_scene {
  int device;
  // mesh (borrowed or owned)
  const float* verts;
  const int* faces;
  const int* colors;
  const float* rem;
  int n_verts, n_faces;
  void* owned_mesh;  // single allocation backing the four arrays when set from host
  size_t owned_mesh_bytes;
  // workspace, sized for cap_faces
  int cap_faces;
  int np;               // segment-tree leaf count (power of two >= n_faces)
  uint32_t* keys[2];
  uint32_t* vals[2];
  uint32_t* hist;       // [1024 digits * n_sort_tiles + 1024 digit totals] (lt_build.hip, LT_RD)
  float4* tris;         // [3 * n_faces]
  float4* seg;          // [2 * 2 * np]
  float4* nodes;        // [4 * max(n_faces - 1, 1)]   binary nodes (64 B)
  float4* nodes4;       // [8 * max(n_faces - 1, 1)]   4-wide nodes (128 B), same numbering
  float* partial;       // per-workgroup bounds partials [6 * LT_BOUNDS_BLOCKS]
  float* params;        // device: lo.xyz, scale, pad
  unsigned* flags;      // device: [0] error bits
  unsigned long long* counters;  // device: nodes, tris, hits, overflows (LT_TRACE_COUNT), then LT_DBG_WAVES clock pairs
  int* overflow;        // [cap_rays * (LT_STACK4_MAX - LT_STACK4_LDS)] stack spill area
  int cap_rays;
  // hand-over of the rays whose walk exceeds the step cap of k_trace4 to k_trace4_tail (one wave per ray)
  int* tail_stack;      // [cap_rays * LT_TAIL_SAVE] saved stack entries (node / leaf references, top last)
  float4* tail_meta;    // [cap_rays] (best_t, best_face bits, entries saved, -)
  int* tail_queue;      // [cap_rays] ray indices
  int* tail_count;      // [2] rays queued by the launch in flight; used alternately (tail_parity), each launch's
                        // tail kernel zeroes the counter of the next one
  int tail_parity;
  // scatter strategy (lt_scatter.hip): state of the render in flight -- per SCENE, so that a ray set is read-only
  // during a render and one ray set (one sensor model) serves any number of scenes and streams at once
  unsigned long long* sc_cell;  // [sc_cap_cells] packed (t, face) z-min per ray; all-ones = empty, re-armed by k_sc_resolve
  int* sc_large;                // [sc_cap_queue] queue of big triangles
  int2* sc_slices;              // [sc_cap_queue] queue of (block, first candidate) slices
  int* sc_large_count;          // [4]: [0] queued big triangles, [1] queued slices; reset by k_sc_resolve
  int sc_cap_cells, sc_cap_queue;
  int built;
  hipStream_t last_stream;
  hipEvent_t probe[2];   // caller's events to record around the dominant kernel of the next cast (one shot)
  hipEvent_t ev[10];
  int have_events;
  lt_stats stats;
};

// TSDF volume (lt_tsdf.hip): four float32 fields [dim_x][dim_y][dim_z], z fastest (numpy C order)
struct lt_tsdf {
  int device;
  int dim[3];
  float origin[3];
  float voxel_size, trunc_margin;
  double voxel_size_d;  // as given (the numpy branch of the reference computes with the Python float, fusion_lidar.py:300)
  double fov_up_deg, fov_down_deg;
  double bnds_given[6];  // vol_bnds as passed to lt_tsdf_create (lt_mergemesh_scan_dev: is this the volume of a scan's geometry?)
  size_t n;
  float *tsdf, *weight, *color, *rem;
  // sparse bookkeeping (lt_tsdf.hip): an (x, y) COLUMN of dim_z voxels is "dirty" when an integrate since the last
  // reset wrote into it -- col_epoch[c] == epoch.  Reset re-initialises dirty columns only and then bumps `epoch`
  // (nothing is cleared); marching cubes skips clean columns (their tsdf is the initial 1 everywhere).
  unsigned* col_epoch;  // [dim_x * dim_y]
  unsigned* col_zw;     // [2][dim_x * dim_y] z range written in the column since the last reset: 0x7fff - lo, hi + 1 (lt_tsdf.hip)
  unsigned epoch;
  int all_dirty;        // the fields were written without column stamps (lt_tsdf_touch): every column counts as written
  unsigned long long* bits;  // [dim_x * dim_y][ceil(dim_z / 64)] sign bit of every voxel's tsdf (the layout of lt_mc.hip),
                             // kept up to date by the column-aware integrate and by reset; invalid while all_dirty
  int* colinfo;         // [dim_x * dim_y] per integrate call: image column px of the voxel column, or -1 = dead
  float* colmax;        // [cap_w] per integrate call: largest depth of every image column
  int cap_w;
  float2* dct;          // [cap_dct] per integrate call: (depth, colour) of pixel (row, px) at [px * im_h + row] -- the voxels of
  size_t cap_dct;       // a column walk read consecutive rows of ONE image column: contiguous here, a line apart in the image
  // the pixel-centric integrate of a fresh volume (k_tsdf_integrate_pix, lt_tsdf.hip)
  int n_obs;            // observations integrated since the last reset (0: every voxel holds its initial value)
  int wd_w;             // image width the wedge table was built for (0: none yet)
  int wd_rho_bits, wd_n_quirk;
  float wd_qscale;
  int* wd_px;           // [dim_x * dim_y] image column of every voxel column (geometry + image width only)
  int* wd_start;        // [wd_w + 1] first table entry of every image column's wedge
  int2* wd_ent;         // [dim_x * dim_y] (voxel column, rho^2 bits) sorted by (image column, rho); column -1: quirk tail
  unsigned* wd_key;     // [dim_x * dim_y] rho quantum of every entry (the binary searches compare these)
  unsigned* wd_qcols;   // [wd_n_quirk] the columns k_tsdf_integrate_quirk evaluates voxel by voxel
  float4* rowtab;       // [rowtab_h] per image row: tan / cos of its pitch range with margins (host, double)
  int rowtab_h;         // capacity
  int rowtab_for_h;     // image height the table holds (the field of view is fixed per volume)
  unsigned* zw_snap;    // [2][dim_x * dim_y] col_zw as it stood before the observation being integrated (non-fresh volumes)
  unsigned* chunk_epoch;  // [ceil(dim_x * dim_y / 64)] == epoch: a column of this chunk of 64 was written since the last reset
  float4* obs4;         // [cap_obs4] transposed (depth, colour, remission, -) of the observations of lt_tsdf_integrate_multi_dev
  size_t cap_obs4;
};

#define LT_BOUNDS_BLOCKS 256
#define LT_DBG_WAVES 16384   // wave start/end clocks kept by a LT_TRACE_COUNT launch (debug)
#define LT_FLAG_BAD_INDEX 1u
#define LT_FLAG_HIER_OVERFLOW 2u  // lt_build.hip, k_agg_*: a window left more roots than LT_AGG_C (cannot happen for <= 62-bit strings)
#define LT_TRACE_DEBUG_STEPS 0x4000u  // internal trace flag: `tri` receives the node steps of each ray (k_trace4)
#define LT_TRACE_DEBUG_TIMES 0x8000u  // internal trace flag: per-wave wall-clock stamps into the counters area

int lt_build_launch(lt_scene* s, hipStream_t stream, lt_stats* stats);
int lt_trace_launch(lt_scene* s, const float* rays, const float* origin, int n_rays, int height,
                    float* endpoints, int* endcolors, float* range, float* endrem, int* tri,
                    unsigned flags, hipStream_t stream, lt_stats* stats);
int lt_check_mesh_args(const char* who, const void* verts, const void* faces, const void* colors, const void* rem,
                        int n_verts, int n_faces);
unsigned lt_env_norm_flag();  // LIDARHIP_NORMALIZE -> 0 | LT_TRACE_NORM_EXACT | LT_TRACE_NORM_AMD
bool lt_binary_path();  // LIDARHIP_TRACE=binary: one ray per lane over binary nodes (A/B cross-check)
int lt_scene_reserve(lt_scene* s, int n_faces);
int lt_scene_reserve_rays(lt_scene* s, int n_rays);

// ---- correctly rounded reciprocal in 5 instructions ------------------------------------------------------------
// The reference multiplies by inv_a = 1.0f / a (Triangle.h:35, an IEEE division).  hipcc's IEEE f32 division is a
// 10-instruction sequence (div_scale x2, rcp, 4 fma, div_fmas, div_fixup) whose scaling steps only matter for
// denormal or huge operands.  For 2^-64 <= |a| <= 2^64 two Newton steps on v_rcp_f32 (1 ulp) with exact fma
// residuals give the correctly rounded quotient: after the first step the relative error is ~2^-46, after the
// second the value fed to the final rounding is within 2^-90 of 1/a, and 1/a cannot be that close to a rounding
// boundary (|a x midpoint - 1| >= 2^-48).  Not taken on trust: lt_debug_verify_rcp compares it with the
// division for EVERY float in that range (tests/test_trace_gpu.py); outside the range the division is used.
#ifdef __HIPCC__
__device__ __forceinline__ float lt_rcp_ieee(float a) {
  if (!(fabsf(a) >= 5.421010862e-20f && fabsf(a) <= 1.8446744e19f)) return 1.0f / a;  // incl. NaN
  float r = __builtin_amdgcn_rcpf(a);
  float e = __builtin_fmaf(-a, r, 1.0f);
  r = __builtin_fmaf(e, r, r);
  e = __builtin_fmaf(-a, r, 1.0f);
  return __builtin_fmaf(e, r, r);
}
// Sweeps over the CHUNKS of a volume (64 consecutive columns (x, y) each; three in four of a street scene are clean): a
// wave (or workgroup) takes `per` chunks, a stride apart -- chunk = k * n + id, k < per, n = the number of waves --, reads
// their stamps with its first lanes and walks the written ones.  Written chunks come in clusters along x AND along y, so the
// stride must not be a whole number of volume rows (dim_y / 64 chunks each): n is nudged upwards, in steps of `multiple`,
// until the `per` chunks of a wave are spread over y (their y offsets advance by ~1 / per of a row).
static inline int lt_deal_count(int n_chunks, int per, int dim_y, int multiple) {
  int n = (n_chunks + per - 1) / per;
  n = ((n + multiple - 1) / multiple) * multiple;
  const double row = dim_y / 64.0;  // chunks per volume row
  if (row > 1.0 && per > 1)
    for (int tries = 0; tries < 256; ++tries, n += multiple) {
      double f = n / row;
      f -= (double)(long long)f;
      if (f >= 0.75 / per && f <= 1.25 / per) break;
    }
  return n;
}

#endif
