// lt_mc.hip -- marching cubes on the device: the mesh of a TSDF volume is born in HBM.
//
// Replaces `TSDFVolume.get_mesh` of the reference (auxiliary/fusion_lidar.py:403-424): get_volume's device-to-host
// copies of three volumes (:395-400), scikit-image's CPU marching cubes (:407), the numpy attribute look-ups
// (:409-423) -- and, downstream, the upload of the mesh for the ray cast (fusion_lidar.py:433-451).  The mesh is
// written as the four arrays the ray cast consumes (verts [V,3] f32 world, faces [F,3] i32, colors [V,3] i32,
// rem [V] f32), indexed (one vertex per sign-changing lattice edge, shared by the cells around it) like
// scikit-image emits it.
//
// The 800 M-voxel default volume (2000 x 2000 x 200, 3.2 GB per field) is 99.8 % empty space, so the float field is
// read ONCE, as a stream, and everything else works on one SIGN BIT per voxel:
//
//   k_mc_signs    tsdf -> one bit per voxel, set when the value is NOT above the level (wave ballot; rows of nz bits padded
//                 to 64-bit words): 3.2 GB in, 128 MB out
//   k_mc_words    one thread per 64-voxel word: crossing-edge masks ex / ey / ez of the edges its voxels own
//                 (m ^ neighbour word, shifted for z), active-cell mask (the 8 corner words are not all equal), number
//                 of vertices (popcounts) and triangles (case table, only on the set bits); the cells of Lewiner's
//                 ambiguous cases are queued; per-workgroup totals
//   k_mc_amb      one lane per queued cell: the face / interior tests on its eight values pick the tiling; its triangles
//                 and centre vertex are added to the counts, the tiling is filed for k_mc_emit_batch
//   k_mc_scan1/2  exclusive scan of the workgroup totals in two levels -> V, F
//   k_mc_compact  word -> compact index of the active words (the ~2 % that own a vertex or a triangle); per active
//                 word a record {word, vertex base, triangle base, ex, ey, ez}
//   k_mc_emit_batch  one WAVE per 8 active words, one lane per VERTEX / per TRIANGLE of the batch (positions by
//                 scikit-image's centre-of-mass rule in double, attributes from the nearest voxel); the index of a vertex
//                 owned by a neighbouring word comes from that word's record (neighbour records staged in LDS): base +
//                 popcount of the edge masks below the bit; a cell's centre vertex follows its word's edge vertices
//
// WHICH mesh: scikit-image 0.18's `marching_cubes_lewiner` (what the reference calls, fusion_lidar.py:407) -- Lewiner's 33
// cases with their face / interior tests and centre vertices, decided on the cell's eight VALUES where the signs do not
// (1-2 % of a street scene's cells; the rest is table look-up on the sign bits as before), every face's vertices in
// scikit-image's order (gradient_direction = "descent").  The vertices (positions bit for bit, colours, remissions) and the
// FACE STREAM -- face k: the same three vertices in the same order -- equal the reference's get_mesh output (golden F10
// made by the real scikit-image: tests/test_pin_f10_f11_gpu.py; the CPU oracle oracle/lt_mc_oracle.c reproduces the
// reference's arrays and is identical to this file up to the numbering of the vertices).  Faces come by (cell ascending in
// (a0, a1, a2), tiling order), which IS scikit-image's order; vertices are numbered by (word of 64 voxels; owner voxel x, y, z
// ascending, edge axis; then the word's centre vertices) where scikit-image numbers them by first use --
// deterministic (the only atomics, k_mc_amb's, are commutative adds and slot claims whose outcome does not reach the
// output's order); scikit-image numbers vertices by first use in a serial face stream.
// Tables: lt_mc_lewiner_table.h (Lewiner's LookUpTable.h as scikit-image ships it, tools/gen_mc_lewiner.py).
#include "lt_internal.h"
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define LT_LW_ATTR __device__
#define LT_LW_NO_RAW
#include "lt_mc_lewiner_table.h"  // Lewiner's tables as the device reads them (tools/gen_mc_lewiner.py, derived())

// ---- Lewiner's case selection, as scikit-image 0.18 runs it ---------------------------------------------------------------
// (MarchingCubes.cpp process_cube / test_face / test_interior; skimage/measure/_marching_cubes_lewiner_cy.pyx the_big_switch /
// test_face / test_internal.)  Everything in double on the float32 field values, level 0; `eps` = np.spacing(1.0) -- the
// value scikit-image's `FLT_EPSILON` holds, despite its name.  Two properties of scikit-image's port are reproduced because
// the reference runs IT (both measured against the real library, tools/mc_lewiner_fuzz.py): its divisions are guarded by
// `+ eps` in the denominator, which decides exact ties, and `test_internal` falls off its end (returns 0) where Lewiner's
// C++ returns `s < 0` -- case 4 with a negative TEST4 entry therefore takes tiling 4.1.2 for the 5 / 10 patterns.
#define LT_MC_EPS 2.220446049250313e-16
__device__ __forceinline__ double lw_pick(const double* v, int i) {  // v[i] without a private-memory array
  double r = v[0];
#pragma unroll
  for (int k = 1; k < 8; ++k) r = (i == k) ? v[k] : r;
  return r;
}
__device__ __forceinline__ bool lw_test_face(const double* v, int face) {
  const int f = face < 0 ? -face : face;
  // faces 1..6: (A, B, C, D) = corners (0,4,5,1) (1,5,6,2) (2,6,7,3) (3,7,4,0) (0,3,2,1) (4,7,6,5)
  const unsigned abcd = f == 1 ? 0x0451u : f == 2 ? 0x1562u : f == 3 ? 0x2673u : f == 4 ? 0x3740u : f == 5 ? 0x0321u : 0x4765u;
  const double A = lw_pick(v, (abcd >> 12) & 7), B = lw_pick(v, (abcd >> 8) & 7), C = lw_pick(v, (abcd >> 4) & 7),
               D = lw_pick(v, abcd & 7);
  const double acbd = A * C - B * D;
  if (acbd > -LT_MC_EPS && acbd < LT_MC_EPS) return face >= 0;
  return (double)face * A * acbd >= 0;
}
// edge < 0: cases 4 and 10 (the interior's saddle along the 0-4 direction); else the reference edge of cases 6, 7, 12, 13
__device__ __forceinline__ bool lw_test_interior(const double* v, int edge, int s) {
  double t, At = 0, Bt, Ct, Dt;
  if (edge < 0) {
    const double a = (v[4] - v[0]) * (v[6] - v[2]) - (v[7] - v[3]) * (v[5] - v[1]);
    const double b = v[2] * (v[4] - v[0]) + v[0] * (v[6] - v[2]) - v[1] * (v[7] - v[3]) - v[3] * (v[5] - v[1]);
    t = -b / (2 * a + LT_MC_EPS);
    if (t < 0 || t > 1) return s > 0;
    At = v[0] + (v[4] - v[0]) * t;
    Bt = v[3] + (v[7] - v[3]) * t;
    Ct = v[2] + (v[6] - v[2]) * t;
    Dt = v[1] + (v[5] - v[1]) * t;
  } else {
    // the reference edge (p, q) and the three edges parallel to it, round the cell: 8 corner numbers, 3 bits each
    const unsigned E = edge == 0 ? 001327645u : edge == 1 ? 012034756u : edge == 2 ? 023105467u : edge == 3 ? 030216574u
                     : edge == 4 ? 045763201u : edge == 5 ? 056470312u : edge == 6 ? 067541023u : edge == 7 ? 074652130u
                     : edge == 8 ? 004372615u : edge == 9 ? 015043726u : edge == 10 ? 026150437u : 037261504u;
    const double p = lw_pick(v, (E >> 21) & 7), q = lw_pick(v, (E >> 18) & 7);
    const double b0 = lw_pick(v, (E >> 15) & 7), b1 = lw_pick(v, (E >> 12) & 7), c0 = lw_pick(v, (E >> 9) & 7),
                 c1 = lw_pick(v, (E >> 6) & 7), d0 = lw_pick(v, (E >> 3) & 7), d1 = lw_pick(v, E & 7);
    t = p / (p - q + LT_MC_EPS);
    Bt = b0 + (b1 - b0) * t;
    Ct = c0 + (c1 - c0) * t;
    Dt = d0 + (d1 - d0) * t;
  }
  const int test = (At >= 0 ? 1 : 0) | (Bt >= 0 ? 2 : 0) | (Ct >= 0 ? 4 : 0) | (Dt >= 0 ? 8 : 0);
  // 0 1 2 3 4 6 8 9 12 -> s > 0;  7 11 13 14 15 -> s < 0;  5 / 10: the saddle's sign, else scikit-image's fall-through (0)
  if ((0x135Fu >> test) & 1u) return s > 0;
  if (test == 5) return (At * Ct - Bt * Dt < LT_MC_EPS) ? s > 0 : false;
  if (test == 10) return (At * Ct - Bt * Dt >= LT_MC_EPS) ? s > 0 : false;
  return s < 0;
}
// a tiling = its rows's offset in LT_LWF | triangles << 16 | uses the centre vertex << 20
#define LWT(T, row) ((unsigned)(LT_LWF_##T + (row) * LT_LWF_##T##_LEN) | ((unsigned)(LT_LWF_##T##_LEN / 3) << 16) | ((unsigned)LT_LWF_##T##_C << 20))
// v[p]: the cell's values in Lewiner's corner order; cs: the device's case index (an AMBIGUOUS one: LT_LWC_FIXED[cs] == ~0)
__device__ __noinline__ unsigned lw_select(const double* v, int cs) {
  const int c = LT_LWC_CASE[cs], g = LT_LWC_CONFIG[cs];
  int sub = 0;
  switch (c) {
    case 3: return lw_test_face(v, LT_LWD_TEST3[g]) ? LWT(TILING3_2, g) : LWT(TILING3_1, g);
    case 4: return lw_test_interior(v, -1, LT_LWD_TEST4[g]) ? LWT(TILING4_1, g) : LWT(TILING4_2, g);
    case 6:
      if (lw_test_face(v, LT_LWD_TEST6[g][0])) return LWT(TILING6_2, g);
      return lw_test_interior(v, LT_LWD_TEST6[g][2], LT_LWD_TEST6[g][1]) ? LWT(TILING6_1_1, g) : LWT(TILING6_1_2, g);
    case 7:
      if (lw_test_face(v, LT_LWD_TEST7[g][0])) sub += 1;
      if (lw_test_face(v, LT_LWD_TEST7[g][1])) sub += 2;
      if (lw_test_face(v, LT_LWD_TEST7[g][2])) sub += 4;
      switch (sub) {
        case 0: return LWT(TILING7_1, g);
        case 1: return LWT(TILING7_2, g * 3 + 0);
        case 2: return LWT(TILING7_2, g * 3 + 1);
        case 3: return LWT(TILING7_3, g * 3 + 0);
        case 4: return LWT(TILING7_2, g * 3 + 2);
        case 5: return LWT(TILING7_3, g * 3 + 1);
        case 6: return LWT(TILING7_3, g * 3 + 2);
        default: return lw_test_interior(v, LT_LWD_TEST7[g][4], LT_LWD_TEST7[g][3]) ? LWT(TILING7_4_2, g) : LWT(TILING7_4_1, g);
      }
    case 10:
      if (lw_test_face(v, LT_LWD_TEST10[g][0])) return lw_test_face(v, LT_LWD_TEST10[g][1]) ? LWT(TILING10_1_1_, g) : LWT(TILING10_2, g);
      if (lw_test_face(v, LT_LWD_TEST10[g][1])) return LWT(TILING10_2_, g);
      return lw_test_interior(v, -1, LT_LWD_TEST10[g][2]) ? LWT(TILING10_1_1, g) : LWT(TILING10_1_2, g);
    case 12:
      if (lw_test_face(v, LT_LWD_TEST12[g][0])) return lw_test_face(v, LT_LWD_TEST12[g][1]) ? LWT(TILING12_1_1_, g) : LWT(TILING12_2, g);
      if (lw_test_face(v, LT_LWD_TEST12[g][1])) return LWT(TILING12_2_, g);
      return lw_test_interior(v, LT_LWD_TEST12[g][3], LT_LWD_TEST12[g][2]) ? LWT(TILING12_1_1, g) : LWT(TILING12_1_2, g);
    case 13: {
#pragma unroll
      for (int k = 0; k < 6; ++k)
        if (lw_test_face(v, LT_LWD_TEST13[g][k])) sub += 1 << k;
      const int sc = LT_LWD_SUBCONFIG13[sub];
      if (sc < 0) return 0u;  // "Impossible case 13?": nothing is added
      if (sc == 0) return LWT(TILING13_1, g);
      if (sc <= 6) return LWT(TILING13_2, g * 6 + sc - 1);
      if (sc <= 18) return LWT(TILING13_3, g * 12 + sc - 7);
      if (sc <= 22) return LWT(TILING13_4, g * 4 + sc - 19);
      if (sc <= 26)
        return lw_test_interior(v, LT_LWD_EDGE13_5[g][sc - 23], LT_LWD_TEST13[g][6]) ? LWT(TILING13_5_1, g * 4 + sc - 23)
                                                                                        : LWT(TILING13_5_2, g * 4 + sc - 23);
      if (sc <= 38) return LWT(TILING13_3_, g * 12 + sc - 27);
      if (sc <= 44) return LWT(TILING13_2_, g * 6 + sc - 39);
      return LWT(TILING13_1_, g);
    }
    default: return 0u;
  }
}
// the tiling of an AMBIGUOUS cell at voxel (x, y, z) with case index cs: the tests on the cell's eight values (Lewiner's
// corner p sits at (a0, a1, a2) = LW_CORNER[p] of the generator / the oracle: 0 (0,0,0)  1 (0,0,1)  2 (0,1,1)  3 (0,1,0)
// 4 (1,0,0)  5 (1,0,1)  6 (1,1,1)  7 (1,1,0) -- scikit-image's x is the last axis).  Only k_mc_amb calls this: the double
// precision tests cost 100 vector registers, which the sweep and the emission kernels must not pay for 1-2 % of the cells.
__device__ __forceinline__ unsigned lw_cell_eval(const float* __restrict__ c, size_t sx, size_t sy, size_t sz, int cs) {
  double v[8];  // (sx, sy, sz: floats between neighbours along x, y, z)
  v[0] = (double)c[0];       v[1] = (double)c[sz];          v[2] = (double)c[sy + sz];      v[3] = (double)c[sy];
  v[4] = (double)c[sx];      v[5] = (double)c[sx + sz];     v[6] = (double)c[sx + sy + sz]; v[7] = (double)c[sx + sy];
  return lw_select(v, cs);
}
// ---- the ambiguous cells' side channel -------------------------------------------------------------------------------
// k_mc_words queues every ambiguous cell it meets (voxel index, case index, its block of the scan) and counts it with 0
// triangles; k_mc_amb evaluates the queue -- one lane per cell --, adds the cell's triangles / centre vertex to the counts
// of its word and of its block (atomics: the one place of the extraction that has them; the RESULT does not depend on their
// order), and files the tiling under the voxel index in an open-addressing table that k_mc_emit_batch reads.
// The queue is SHARDED: 64 segments of `cap` entries, a counter each in a 128-byte line of its own; a block of k_mc_words
// appends to segment (block & 63).  One counter for all took ~5 000 returning atomics in a row on one address -- 10 of
// k_mc_words' 43 us (measured by taking the atomic out); the order of the queue does not reach the output.
#define LT_MC_SHARDS 64
#define LT_MC_CTR_STRIDE 32  // ints between two shards' counters
struct mc_amb {
  uint2* queue;      // [LT_MC_SHARDS][cap]: (voxel index, case | block << 8)
  int* counter;      // [LT_MC_SHARDS * LT_MC_CTR_STRIDE]: cells queued per shard (may exceed cap: the host then grows the
                     // buffers and starts over), then [LT_MC_SHARDS] the counts of the LAST extraction (k_mc_clear)
  uint2* table;      // [tcap], tcap a power of two >= 4 x LT_MC_SHARDS x cap: (voxel index + 1, tiling); 0 = empty
  unsigned cap, tmask;
};
__device__ __forceinline__ unsigned amb_slot(unsigned i, unsigned tmask) { return (i * 2654435761u) & tmask; }
__device__ __forceinline__ unsigned amb_lookup(const mc_amb& A, unsigned i) {
  for (unsigned h = amb_slot(i, A.tmask);; h = (h + 1) & A.tmask) {
    const uint2 e = A.table[h];
    if (e.x == i + 1u) return e.y;
    if (e.x == 0u) return 0u;  // (not filed: cannot happen after k_mc_amb; no triangles rather than a wild read)
  }
}
#define LW_NT(sel) (((sel) >> 16) & 15u)
#define LW_C(sel) (((sel) >> 20) & 1u)
#define LW_OFF(sel) ((sel) & 0xFFFFu)

// debug (-DLT_MC_STAMP=1: k_mc_words, =2: k_mc_compact; tools/mc_wave_times.py): wall clock (100 MHz) at the start of a wave,
// after its stamp ballot and at its end, and the number of blocks it walked
#ifdef LT_MC_STAMP
__device__ unsigned long long g_mc_stamp[4 << 16];
extern "C" int lt_debug_mc_stamps(unsigned long long* out, int n_waves) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mc_stamp), (size_t)min(n_waves, 1 << 16) * 4 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#define LT_MC_STAMP_AT(which, w, i, v) do { if (LT_MC_STAMP == (which) && (threadIdx.x & 63) == 0 && (w) < (1 << 16)) g_mc_stamp[(w) * 4 + (i)] = (v); } while (0)
// =3: k_mc_emit_batch by section -- g_mc_stamp[8 b + i]: the wall clock between marks i - 1 and i of batch b (plain stores:
// atomics on eight shared words serialise the waves behind them and measure themselves)
#define LT_MC_SECTION(i) do { if (LT_MC_STAMP == 3) { const unsigned long long t_ = wall_clock64(); if ((threadIdx.x & 63) == 0 && bi < (1 << 15)) g_mc_stamp[bi * 8 + (i)] = t_ - st_last; st_last = t_; } } while (0)
#else
#define LT_MC_STAMP_AT(which, w, i, v) do { } while (0)
#define LT_MC_SECTION(i) do { } while (0)
#endif

typedef unsigned long long u64;

struct mc_dims {
  int nx, ny, nz;
  int wz;        // 64-bit words per (x, y) row
  int n_words;   // nx * ny * wz
  unsigned ny_m; int ny_s;  // row / ny likewise (mc_x_of)
  int vs;        // floats between two voxels of a field array: 1 (three separate arrays, lt_marching_cubes_dev) or 4 (a
                 // TSDF volume's interleaved (tsdf, weight, colour, remission) records, lt_tsdf.hip LT_VOX)
  unsigned wz_m; int wz_s;  // w / wz by multiplication (mc_row_of): an integer division is ~40 vector instructions, and
                            // k_mc_words runs one per word of the volume (16 M on the default one)
};

// row = w / D.wz for 0 <= w < 2^31 (round-up method: m = floor(2^32 (2^s - d) / d) + 1, s = ceil(log2 d))
__device__ __forceinline__ int mc_udiv(int n_, unsigned m, int s) {
  if (s == 0) return n_;  // d == 1
  const unsigned n = (unsigned)n_, t = __umulhi(n, m);
  return (int)((t + ((n - t) >> 1)) >> (s - 1));
}
__device__ __forceinline__ int mc_row_of(const mc_dims& D, int w) { return mc_udiv(w, D.wz_m, D.wz_s); }
__device__ __forceinline__ int mc_x_of(const mc_dims& D, int row) { return mc_udiv(row, D.ny_m, D.ny_s); }  // x = row / ny
static void mc_magic(unsigned d, unsigned* m, int* s) {  // (host)
  int sh = 0;
  while ((1u << sh) < d) ++sh;
  *s = sh;
  *m = sh == 0 ? 0u : (unsigned)((((unsigned long long)1 << 32) * (((unsigned long long)1 << sh) - d)) / d + 1ull);
}

struct mc_rec {  // one per active word
  int w, vbase, tbase, pad;  // pad: the word's triangle count | its number of centre vertices << 16
  u64 ex, ey, ez;
};

struct lt_mesh {
  int device;
  float* verts; int* faces; int* colors; float* rem;
  int cap_v, cap_f, n_verts, n_faces;
  u64* bits; unsigned* cnt; int* cmap; size_t cap_words;
  int* blk; size_t cap_blocks;          // 3 ints per workgroup of 256 words, then 4 totals
  int* totals_host;                     // pinned: {active words, vertices, triangles, -}
  mc_rec* rec; size_t cap_rec;
  // cnt / cmap hold 0 / -1 for every word that is not an active word of the LAST extraction (k_mc_clear undoes those
  // before the next one): the 2 x 64 MB of the default volume are never swept again.  n_prev: active words of the last
  // extraction (their records are still in `rec`); state_dirty: an extraction did not finish -- sweep everything once
  int n_prev, state_dirty;
  int* wave_na; size_t cap_wave_na;     // active words per wave of 64 rows (k_mc_words -> k_mc_compact)
  mc_amb amb; size_t amb_cap;           // the ambiguous cells' queue and tiling table (k_mc_amb); counter = amb.counter
  int n_amb_prev;                       // cells of the last extraction: their table slots are emptied by k_mc_clear
  // lt_mesh_renumber_dev: the second set of vertex arrays (swapped with the first), first use / new number per vertex, block sums
  float* verts2; int* colors2; float* rem2; int cap_v2;
  int* rn_first; int* rn_newid; size_t cap_rn; int* rn_bsum; size_t cap_bsum;
  float ms_signs, ms_rest;              // last extraction (when timed)
  hipEvent_t ev[3];
};

// ---- k_mc_signs ----------------------------------------------------------------------------------------------------
// col_epoch / epoch (may be NULL): stamps of the (x, y) columns written since the volume's last reset (lt_tsdf.hip) --
// a clean column still holds the initial tsdf = 1 everywhere, its sign bits are 0 without reading it.  A wave takes 64
// rows (= columns of the volume) at a time: lane r zeroes the words of row r when that row is clean; the dirty rows
// are then walked one by one, z along the lanes, four words (loads) in flight.
__global__ __launch_bounds__(256) void k_mc_signs(const float* __restrict__ tsdf, mc_dims D, u64* __restrict__ bits,
                                                  const unsigned* __restrict__ col_epoch, unsigned epoch) {
  const int lane = threadIdx.x & 63;
  const int n_rows = D.nx * D.ny;
  const int n_chunks = (n_rows + 63) / 64;
  for (int chunk = blockIdx.x * 4 + (threadIdx.x >> 6); chunk < n_chunks; chunk += gridDim.x * 4) {
    const int row = chunk * 64 + lane;
    bool dirty = false;
    if (row < n_rows) {
      dirty = !col_epoch || col_epoch[row] == epoch;
      if (!dirty)
        for (int k = 0; k < D.wz; ++k) bits[(size_t)row * D.wz + k] = 0ull;
    }
    u64 m = __ballot(dirty);
    while (m) {
      const int r = chunk * 64 + (__ffsll((long long)m) - 1);
      m &= m - 1;
      const float* src = tsdf + (size_t)r * D.nz * D.vs;
      for (int k0 = 0; k0 < D.wz; k0 += 4) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int z = (k0 + k) * 64 + lane;
          v[k] = z < D.nz ? src[(size_t)z * D.vs] : 1.0f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const u64 w = __ballot(!(v[k] > 0.0f));  // level 0: the bit says "NOT above the level" (Lewiner's index bit, inverted)
          if (lane == 0 && k0 + k < D.wz) bits[(size_t)r * D.wz + k0 + k] = w;
        }
      }
    }
  }
}

// ---- masks of one word ---------------------------------------------------------------------------------------------
struct mc_masks {
  u64 m[2][2];  // sign bits of the rows (x + dx, y + dy)
  u64 s[2][2];  // the same shifted by one z (bit b = voxel z + 1)
  u64 ex, ey, ez, ac;
};

__device__ __forceinline__ u64 low_bits(int n) { return n >= 64 ? ~0ull : (n <= 0 ? 0ull : ((1ull << n) - 1ull)); }

// masks from the 8 sign words of a word's cell corners, w8[dx | dy << 1 | dw << 2] (0 where the volume ends)
__device__ __forceinline__ mc_masks mc_build(const u64* w8, const mc_dims& D, int x, int y, int wz) {
  mc_masks M;
  const bool hx = x + 1 < D.nx, hy = y + 1 < D.ny;
#pragma unroll
  for (int dx = 0; dx < 2; ++dx)
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      const u64 m = w8[dx | (dy << 1)], nxt = w8[dx | (dy << 1) | 4];
      M.m[dx][dy] = m;
      M.s[dx][dy] = (m >> 1) | (nxt << 63);
    }
  const u64 vzn = low_bits(D.nz - wz * 64);      // voxels that exist
  const u64 vz = low_bits(D.nz - 1 - wz * 64);   // voxels with a +z neighbour
  M.ex = hx ? ((M.m[0][0] ^ M.m[1][0]) & vzn) : 0ull;
  M.ey = hy ? ((M.m[0][0] ^ M.m[0][1]) & vzn) : 0ull;
  M.ez = (M.m[0][0] ^ M.s[0][0]) & vz;
  M.ac = 0ull;
  if (hx && hy) {
    const u64 any = M.m[0][0] | M.m[0][1] | M.m[1][0] | M.m[1][1] | M.s[0][0] | M.s[0][1] | M.s[1][0] | M.s[1][1];
    const u64 all = M.m[0][0] & M.m[0][1] & M.m[1][0] & M.m[1][1] & M.s[0][0] & M.s[0][1] & M.s[1][0] & M.s[1][1];
    M.ac = (any & ~all) & vz;
  }
  return M;
}

__device__ __forceinline__ mc_masks mc_load(const u64* __restrict__ bits, const mc_dims& D, int x, int y, int wz) {
  const bool hx = x + 1 < D.nx, hy = y + 1 < D.ny, hz = wz + 1 < D.wz;
  const int row = x * D.ny + y;
  u64 w8[8];
#pragma unroll
  for (int dx = 0; dx < 2; ++dx)
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      const bool have = (dx == 0 || hx) && (dy == 0 || hy);
      const size_t i = (size_t)(row + dx * D.ny + dy) * D.wz + wz;
      w8[dx | (dy << 1)] = have ? bits[i] : 0ull;
      w8[dx | (dy << 1) | 4] = (have && hz) ? bits[i + 1] : 0ull;
    }
  return mc_build(w8, D, x, y, wz);
}

// case index of the cell at bit b: corner i at (dx, dy, dz) = (i & 1, (i >> 1) & 1, (i >> 2) & 1)
__device__ __forceinline__ int mc_case(const mc_masks& M, int b) {
  int cs = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const u64 m = (i & 4) ? M.s[i & 1][(i >> 1) & 1] : M.m[i & 1][(i >> 1) & 1];
    cs |= (int)((m >> b) & 1ull) << i;
  }
  return cs;
}

// block-wide sum / exclusive scan of three small counters packed into one 64-bit word (20 bits each)
__device__ __forceinline__ u64 pack3(unsigned a, unsigned v, unsigned t) { return (u64)a | ((u64)v << 20) | ((u64)t << 40); }

__device__ __forceinline__ u64 wave_incl_scan(u64 p) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const u64 q = __shfl_up(p, o, 64);
    if (lane >= o) p += q;
  }
  return p;
}

// ---- k_mc_words -----------------------------------------------------------------------------------------------------
// a word can only own a crossing edge or an active cell if its row (x, y) or one of the rows (x, y + 1), (x + 1, y),
// (x + 1, y + 1) was written since the volume's last reset; the stamps are 4 B per row against 8 sign words per thread
__device__ __forceinline__ bool mc_rows_dirty(const unsigned* __restrict__ col_epoch, unsigned epoch, const mc_dims& D,
                                              int row) {
  if (!col_epoch) return true;
  const int last = D.nx * D.ny - 1;
  return col_epoch[row] == epoch || col_epoch[min(row + 1, last)] == epoch || col_epoch[min(row + D.ny, last)] == epoch ||
         col_epoch[min(row + D.ny + 1, last)] == epoch;
}

// One thread per ROW (x, y) of wz words, one block of the scan per WAVE (64 rows): a thread per word was 250 000 waves of
// four stamp loads and a store -- 30 rounds of resident waves waiting for a load each, 103 us -- and the four words of a
// row share their stamps.
//
// Which blocks a wave works on.  The volume stamps every CHUNK of 64 columns it writes into (lt_tsdf.hip), and a block of 64
// rows r .. r + 63 depends on the chunks of the rows r, r + 1, r + ny, r + ny + 1: four flags.  Five blocks in six of a
// street scene are clean (their counts already hold 0, see lt_mesh).  A wave takes LT_MC_BLOCKS_PER_WAVE blocks: its first
// lanes read the flags of one block each, a ballot finds the live ones, and the wave walks those -- 7 816 waves instead of
// 62 500, of which most loaded their flags and left.  The blocks of a wave are a stride apart (block = lane x number of
// waves + wave, lt_deal_count): written chunks come in clusters, dealt this way every wave gets its share (1.4 live
// blocks on average, 4 at most).  What the kernel's time is made of (per-wave wall-clock stamps, tools/mc_wave_times.py):
// a live block is a chain of four dependent round trips to cold memory -- case table, chunk flags, the rows' stamps, the
// sign words: 0.9 us each in this kernel, 0.35 us on an idle chip -- plus 3-4 us of cell loops for the slowest lane: 7.6 us
// for a wave with one block, 11 us with two; the slots are 0.2 busy.
#ifndef LT_MC_BLOCKS_PER_WAVE
#define LT_MC_BLOCKS_PER_WAVE 8
#endif
__device__ __forceinline__ bool mc_block_live(const unsigned* __restrict__ chunk_epoch, unsigned epoch, const mc_dims& D,
                                              int b, int n_chunks) {
  if (!chunk_epoch) return true;
  const int q = (b * 64 + D.ny) >> 6;
  bool any = false;
  const int ch[4] = {b, b + 1, q, q + 1};
#pragma unroll
  for (int k = 0; k < 4; ++k) any = any || (ch[k] < n_chunks && chunk_epoch[min(ch[k], n_chunks - 1)] == epoch);
  return any;
}

__global__ __launch_bounds__(256) void k_mc_words(mc_amb A, const u64* __restrict__ bits, mc_dims D, unsigned* __restrict__ cnt,
                                                  int* __restrict__ blk, const unsigned* __restrict__ col_epoch,
                                                  unsigned epoch, const unsigned* __restrict__ chunk_epoch,
                                                  int* __restrict__ wave_na, int n_blocks) {
  const int n_rows = D.nx * D.ny;
  const int n_chunks = (n_rows + 63) / 64;
  __shared__ unsigned char s_nt[256];  // triangles per case: the loops below look it up once per active cell, a chain of
  {                                    // dependent loads that is three times shorter through LDS.  0x80: an ambiguous case of
    const unsigned fx = LT_LWC_FIXED[threadIdx.x];  // Lewiner's (3, 4, 6, 7, 10, 12, 13) -- the cell's eight VALUES decide
    s_nt[threadIdx.x] = fx == 0xFFFFFFFFu ? 0x80 : (unsigned char)LW_NT(fx);  // (k_mc_amb; 1-2 % of a street scene's cells)
  }
  // the ambiguous cells a wave meets in one block of 64 rows, collected in LDS and appended to the queue with ONE global
  // atomic (a returning atomic per cell on the one counter -- 14 500 of them on a street scene -- doubled the kernel's time)
#define LT_MC_QW 96
  __shared__ uint2 s_q[4][LT_MC_QW];
  __shared__ int s_qn[4];
  const int wv = threadIdx.x >> 6;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int n_waves = gridDim.x * 4, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  LT_MC_STAMP_AT(1, wave, 0, wall_clock64());
  u64 live;
  {
    const int b = lane * n_waves + wave;
    const bool mine = lane < LT_MC_BLOCKS_PER_WAVE && b < n_blocks;
    const bool act = mine && mc_block_live(chunk_epoch, epoch, D, b, n_chunks);
    if (mine && !act) {  // (cnt already holds 0 for the words of a clean block)
      blk[3 * b] = 0; blk[3 * b + 1] = 0; blk[3 * b + 2] = 0;
      wave_na[b] = 0;
    }
    live = __ballot(act);
  }
  LT_MC_STAMP_AT(1, wave, 1, wall_clock64());
  LT_MC_STAMP_AT(1, wave, 3, (unsigned long long)__popcll(live));
  while (live) {
  const int blkid = (__ffsll((long long)live) - 1) * n_waves + wave;  // (wave-uniform)
  live &= live - 1;
  const int row = blkid * 64 + lane;
  unsigned na = 0, nv = 0, nt = 0;
  if (lane == 0) s_qn[wv] = 0;
  __builtin_amdgcn_wave_barrier();
  // (a clean row writes nothing: its words hold 0 -- no active word of the last extraction is left, k_mc_clear)
  const bool rowlive = row < n_rows && mc_rows_dirty(col_epoch, epoch, D, row);
  const int rowc = min(row, n_rows - 1);
  const int x = mc_x_of(D, rowc), y = rowc - x * D.ny;
  // Triangles of a word = sum over its active cells of the case's count.  A lane walking its own cells is a chain of
  // ~60 instructions and an LDS look-up per cell, and a row through a wall along z has 60 active cells a word: its wave
  // waits for that one lane, and the launch for its slowest waves.  Words with more than LT_MC_HEAVY cells are therefore
  // counted by the WAVE, a lane per cell: the owner's eight corner masks are broadcast (v_readlane), every lane looks up
  // its cell, three ballots add the counts up (42 -> 32 us on the default volume, with the threshold swept and the lanes'
  // own loops on 32-bit half words).
#ifndef LT_MC_HEAVY_LANES
#define LT_MC_HEAVY_LANES 8   // (swept: 4 .. 12: 31.5 us, 20 .. 32: 32.4, 64 = never walked: 37.5)
#endif
#ifndef LT_MC_HEAVY
#define LT_MC_HEAVY 16  // (swept on the default volume: 2: 50 us, 4: 39, 6: 38, 10 .. 24: 32-33, 40: 35, never: 39)
#endif
  // -> triangles | centre vertices << 16 of the word k of this lane's row
  auto count_tris = [&](const mc_masks& M, int k) -> unsigned {  // (called by all 64 lanes)
    unsigned t = 0;
    auto ambiguous = [&](int xx, int yy, int b, unsigned cs) -> unsigned {  // queued for k_mc_amb; counts 0 here
      const uint2 e = make_uint2((unsigned)((xx * D.ny + yy) * D.nz + k * 64 + b), cs | ((unsigned)blkid << 8));
      const int slot = atomicAdd(&s_qn[wv], 1);  // (LDS)
      if (slot < LT_MC_QW) s_q[wv][slot] = e;
      else {  // (more than the buffer holds in one block of rows: straight to the queue)
        const unsigned sh = (unsigned)blkid & (LT_MC_SHARDS - 1);
        const unsigned pos = (unsigned)atomicAdd(A.counter + sh * LT_MC_CTR_STRIDE, 1);
        if (pos < A.cap) A.queue[(size_t)sh * A.cap + pos] = e;
      }
      return 0u;
    };
    // ... unless MANY lanes hold such a word (a wall across the block's rows): the wave's turn costs ~90 instructions
    // per heavy word, the lanes' own loops ~25 per cell of the fullest lane -- 64 heavy words at once are cheaper walked
    bool heavy = __popcll(M.ac) > LT_MC_HEAVY;
    if (__popcll(__ballot(heavy)) > LT_MC_HEAVY_LANES) heavy = false;
    if (!heavy) {
      // the word's halves one after the other: with 32-bit masks a cell is 8 x (v_bfe_u32, v_lshl_or_b32) + a 32-bit
      // find-first / clear-lowest, a third of the instructions of the 64-bit shifts of mc_case
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        unsigned a = (unsigned)(M.ac >> (32 * h));
        if (a == 0u) continue;
        unsigned m[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
          m[i] = (unsigned)(((i & 4) ? M.s[i & 1][(i >> 1) & 1] : M.m[i & 1][(i >> 1) & 1]) >> (32 * h));
        for (; a; a &= a - 1) {
          const int b = __ffs((int)a) - 1;
          unsigned cs = 0;
#pragma unroll
          for (int i = 0; i < 8; ++i) cs |= ((m[i] >> b) & 1u) << i;
          const unsigned n = s_nt[cs];
          t += (n & 0x80u) ? ambiguous(x, y, 32 * h + b, cs) : n;
        }
      }
    }
    for (u64 hm = __ballot(heavy); hm; hm &= hm - 1) {
      const int r = __ffsll((long long)hm) - 1;  // (wave-uniform)
      int cs = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const u64 m = (i & 4) ? M.s[i & 1][(i >> 1) & 1] : M.m[i & 1][(i >> 1) & 1];
        const u64 mr = (u64)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)m, r) |
                       ((u64)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(m >> 32), r) << 32);
        cs |= (int)((mr >> lane) & 1ull) << i;
      }
      const u64 ac = (u64)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)M.ac, r) |
                     ((u64)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(M.ac >> 32), r) << 32);
      unsigned n = ((ac >> lane) & 1ull) ? s_nt[cs] : 0u;  // (<= 12, or the marker)
      if (n & 0x80u)  // (the OWNER's x, y: broadcast like its masks)
        n = ambiguous(__builtin_amdgcn_readlane(x, r), __builtin_amdgcn_readlane(y, r), lane, (unsigned)cs);
      const unsigned tot = __popcll(__ballot(n & 1u)) + 2u * __popcll(__ballot(n & 2u)) + 4u * __popcll(__ballot(n & 4u)) +
                           8u * __popcll(__ballot(n & 8u)) + ((unsigned)__popcll(__ballot(n >> 16)) << 16);
      if (lane == r) t = tot;
    }
    return t;
  };
  if (D.wz <= 4) {
    // the sign words of the four rows (x + dx, y + dy), ALL loaded before the first is used -- a word's masks need
    // its row neighbours' words at k and k + 1, so a loop over k loaded every word twice, in wz dependent rounds
    const bool hx = x + 1 < D.nx, hy = y + 1 < D.ny;
    u64 w[4][5];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int dx = q & 1, dy = q >> 1;
      const bool have = rowlive && (dx == 0 || hx) && (dy == 0 || hy);
      const size_t base = (size_t)(rowc + (have ? dx * D.ny + dy : 0)) * D.wz;
#pragma unroll
      for (int k = 0; k < 4; ++k) w[q][k] = (have && k < D.wz) ? bits[base + k] : 0ull;
      w[q][4] = 0ull;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k >= D.wz) break;  // (wave-uniform)
      u64 w8[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) { w8[q] = w[q][k]; w8[q | 4] = w[q][k + 1]; }
      const mc_masks M = mc_build(w8, D, x, y, k);  // (a row that is not live holds zeros: no edge, no cell)
      const unsigned tc = count_tris(M, k), t = tc & 0xFFFFu;
      const unsigned v = __popcll(M.ex) + __popcll(M.ey) + __popcll(M.ez) + (tc >> 16);
      if (rowlive) cnt[(size_t)row * D.wz + k] = v | (t << 16);
      na += (v | t) ? 1u : 0u; nv += v; nt += t;
    }
  } else
  for (int k = 0; k < D.wz; ++k) {
    mc_masks M = mc_load(bits, D, x, y, k);
    if (!rowlive) { M.ex = M.ey = M.ez = M.ac = 0ull; }
    const unsigned tc = count_tris(M, k), t = tc & 0xFFFFu;
    const unsigned v = __popcll(M.ex) + __popcll(M.ey) + __popcll(M.ez) + (tc >> 16);
    if (rowlive) cnt[(size_t)row * D.wz + k] = v | (t << 16);
    na += (v | t) ? 1u : 0u; nv += v; nt += t;
  }
  {  // the block's ambiguous cells -> the queue
    __builtin_amdgcn_wave_barrier();
    const int nq = min(s_qn[wv], LT_MC_QW);  // (wave-uniform)
    if (nq > 0) {
      unsigned base = 0;
      const unsigned sh = (unsigned)blkid & (LT_MC_SHARDS - 1);
      if (lane == 0) base = (unsigned)atomicAdd(A.counter + sh * LT_MC_CTR_STRIDE, nq);
      base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
      for (int j = lane; j < nq; j += 64)
        if (base + (unsigned)j < A.cap) A.queue[(size_t)sh * A.cap + base + j] = s_q[wv][j];
    }
  }
  u64 p = pack3(na, nv, nt);  // (20 bits each: 64 rows x wz words x <= 768 triangles -- the host checks wz)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) p += __shfl_xor(p, o, 64);
  if (lane == 0) {
    blk[3 * blkid] = (int)(p & 0xFFFFF);
    blk[3 * blkid + 1] = (int)((p >> 20) & 0xFFFFF);
    blk[3 * blkid + 2] = (int)((p >> 40) & 0xFFFFF);
    wave_na[blkid] = (int)(p & 0xFFFFF);
  }
  }
  LT_MC_STAMP_AT(1, wave, 2, wall_clock64());
}

// ---- k_mc_amb: the queued ambiguous cells, one lane each ------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mc_amb(const float* __restrict__ tsdf, mc_dims D, mc_amb A, unsigned* __restrict__ cnt,
                                                int* __restrict__ blk, int* __restrict__ wave_na) {
  // workgroup g: shard g % 64, its chunks of 256 cells g / 64, g / 64 + gridDim / 64, ...
  const unsigned sh = blockIdx.x & (LT_MC_SHARDS - 1), per = max(gridDim.x / LT_MC_SHARDS, 1u);
  const unsigned n = min((unsigned)A.counter[sh * LT_MC_CTR_STRIDE], A.cap);
  const uint2* __restrict__ queue = A.queue + (size_t)sh * A.cap;
  const size_t sz = (size_t)D.vs, sy = (size_t)D.nz * sz, sx = (size_t)D.ny * sy;
  // a workgroup takes 256 queued cells and deals them to its lanes SORTED by Lewiner's case (counting sort in LDS): the seven
  // ambiguous cases run different tests, and a wave pays for every case its lanes hold
  __shared__ uint2 s_e[256];
  __shared__ int s_n[16], s_o[16];
  const int tid = threadIdx.x;
  for (unsigned base = (blockIdx.x / LT_MC_SHARDS) * 256; base < n; base += per * 256) {  // (workgroup-uniform)
    if (tid < 16) s_n[tid] = 0;
    __syncthreads();
    const bool valid = base + tid < n;
    uint2 e0 = make_uint2(0u, 0u);
    int bucket = 0, rank = 0;
    if (valid) {
      e0 = queue[base + tid];
      bucket = LT_LWC_CASE[e0.y & 255u];  // 3, 4, 6, 7, 10, 12, 13
      rank = atomicAdd(&s_n[bucket], 1);
    }
    __syncthreads();
    if (tid == 0) {
      int o = 0;
      for (int k = 0; k < 16; ++k) { s_o[k] = o; o += s_n[k]; }
    }
    __syncthreads();
    if (valid) s_e[s_o[bucket] + rank] = e0;
    __syncthreads();
    if (!valid) continue;  // (no barrier below)
    const uint2 e = s_e[tid];
    const unsigned i = e.x;
    const int cs = (int)(e.y & 255u), b = (int)(e.y >> 8);
    const unsigned sel = lw_cell_eval(tsdf + (size_t)i * sz, sx, sy, sz, cs);
    for (unsigned h = amb_slot(i, A.tmask);; h = (h + 1) & A.tmask)  // (every cell is queued once: the slot is free or foreign)
      if (atomicCAS(&A.table[h].x, 0u, i + 1u) == 0u) { A.table[h].y = sel; break; }
    const unsigned nt = LW_NT(sel), c = LW_C(sel);
    if (nt) {
      const unsigned row = i / (unsigned)D.nz, z = i - row * (unsigned)D.nz;
      const unsigned old = atomicAdd(&cnt[(size_t)row * D.wz + (z >> 6)], c | (nt << 16));
      if (c) atomicAdd(&blk[3 * b + 1], (int)c);
      atomicAdd(&blk[3 * b + 2], (int)nt);
      if (old == 0u) { atomicAdd(&blk[3 * b], 1); atomicAdd(&wave_na[b], 1); }  // the word becomes an active word
    }
  }
}

// undo the last extraction's entries of cnt / cmap (its records are still there): both arrays are back to 0 / -1
// ... and of the ambiguous cells' tiling table (its queue is still there too): every queued cell empties its own slot
__global__ __launch_bounds__(256) void k_mc_clear(const mc_rec* __restrict__ rec, int n, unsigned* __restrict__ cnt,
                                                  int* __restrict__ cmap, mc_amb A, int n_amb) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  // n_amb: the largest shard's count of the last extraction; thread i < 64 n_amb looks at entry i % n_amb of shard i / n_amb
  const int sh = n_amb > 0 ? i / n_amb : LT_MC_SHARDS, j = n_amb > 0 ? i - sh * n_amb : 0;
  if (sh < LT_MC_SHARDS && j < A.counter[LT_MC_SHARDS * LT_MC_CTR_STRIDE + sh]) {
    const unsigned key = A.queue[(size_t)sh * A.cap + j].x + 1u;
    unsigned h = amb_slot(key - 1u, A.tmask);
    for (unsigned tries = 0; tries <= A.tmask; ++tries, h = (h + 1) & A.tmask)  // (not stopped by slots others emptied)
      if (A.table[h].x == key) { A.table[h].x = 0u; break; }
  }
  if (i >= n) return;
  const int w = rec[i].w;
  cnt[w] = 0u;
  cmap[w] = -1;
}

// ---- k_mc_scan1 / k_mc_scan2: exclusive scan of the per-workgroup totals in two levels -------------------------------
// level 1: workgroup g scans its segment of 256 totals in place (coalesced) and leaves the segment's sums in seg[3g..];
// level 2: one workgroup scans the (<= 1024) segment sums in place and writes the grand totals.  k_mc_compact adds
// seg[3 * (block / 256) + k] to the block's in-segment prefix.  (A single workgroup walking all 62 500 totals of the
// default volume took 143 us: one CU's L2 bandwidth.)
__global__ __launch_bounds__(256) void k_mc_scan1(int* __restrict__ blk, int n_blocks, int* __restrict__ seg) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  unsigned a = 0, v = 0, t = 0;
  if (b < n_blocks) { a = (unsigned)blk[3 * b]; v = (unsigned)blk[3 * b + 1]; t = (unsigned)blk[3 * b + 2]; }
  // (the 20-bit packing of pack3 would overflow here -- 256 x 81 920 triangles: three 64-bit scans)
  u64 p[3] = {a, v, t}, inc[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) inc[k] = wave_incl_scan(p[k]);
  __shared__ u64 ws[3][4];
  if ((threadIdx.x & 63) == 63)
    for (int k = 0; k < 3; ++k) ws[k][threadIdx.x >> 6] = inc[k];
  __syncthreads();
  u64 off[3] = {0, 0, 0}, tot[3];
  for (int k = 0; k < 3; ++k) {
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) off[k] += ws[k][w];
    tot[k] = (ws[k][0] + ws[k][1]) + (ws[k][2] + ws[k][3]);
  }
  if (b < n_blocks)
    for (int k = 0; k < 3; ++k) blk[3 * b + k] = (int)min(off[k] + inc[k] - p[k], (u64)2147483647);
  if (threadIdx.x == 0)
    for (int k = 0; k < 3; ++k) seg[3 * blockIdx.x + k] = (int)min(tot[k], (u64)2147483647);
}

__global__ __launch_bounds__(1024) void k_mc_scan2(int* __restrict__ seg, int n_seg, int* __restrict__ totals,
                                                   int* __restrict__ amb_counter, int amb_cap) {
  __shared__ long long part[3][1024];
  const int t = threadIdx.x;
  long long s[3] = {0, 0, 0};
  if (t < n_seg)
    for (int k = 0; k < 3; ++k) s[k] = seg[3 * t + k];
  for (int k = 0; k < 3; ++k) part[k][t] = s[k];
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    long long v[3];
    for (int k = 0; k < 3; ++k) v[k] = t >= o ? part[k][t - o] : 0;
    __syncthreads();
    for (int k = 0; k < 3; ++k) part[k][t] += v[k];
    __syncthreads();
  }
  if (t < n_seg)
    for (int k = 0; k < 3; ++k) seg[3 * t + k] = (int)min(part[k][t] - s[k], 2147483647ll);
  if (t < 64) {  // the ambiguous cells queued per shard (k_mc_amb is done): kept for k_mc_clear, re-armed for the next extraction
    const int c = amb_counter[t * LT_MC_CTR_STRIDE];
    amb_counter[LT_MC_SHARDS * LT_MC_CTR_STRIDE + t] = min(c, amb_cap);
    amb_counter[t * LT_MC_CTR_STRIDE] = 0;
    int mx = c;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
    if (t == 0) totals[3] = mx;  // the fullest shard (beyond the capacity: the host grows the buffers and starts over)
  }
  if (t == 1023)
    for (int k = 0; k < 3; ++k) totals[k] = (int)min(part[k][1023], 2147483647ll);
}

// ---- k_mc_compact ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mc_compact(const u64* __restrict__ bits, mc_dims D,
                                                    const unsigned* __restrict__ cnt, const int* __restrict__ blk,
                                                    const int* __restrict__ seg, int* __restrict__ cmap,
                                                    mc_rec* __restrict__ rec, int cap_rec,
                                                    const int* __restrict__ wave_na, int n_blocks) {
  // thread = row, scan block = a wave's turn (as in k_mc_words, and dealt the same way); compact order = word order
  const int n_rows = D.nx * D.ny;
  const int lane = threadIdx.x & 63;
  const int n_waves = gridDim.x * 4, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  LT_MC_STAMP_AT(2, wave, 0, wall_clock64());
  u64 live;
  {  // no active word in a block's 64 rows (most): nothing to rank, and their cmap entries are -1 already (lt_mesh)
    const int b = lane * n_waves + wave;
    live = __ballot(lane < LT_MC_BLOCKS_PER_WAVE && b < n_blocks && wave_na[min(b, n_blocks - 1)] != 0);
  }
  LT_MC_STAMP_AT(2, wave, 1, wall_clock64());
  LT_MC_STAMP_AT(2, wave, 3, (unsigned long long)__popcll(live));
  while (live) {
  const int blkid = (__ffsll((long long)live) - 1) * n_waves + wave;  // (wave-uniform)
  live &= live - 1;
  const int row = blkid * 64 + lane;
  unsigned na = 0, nv = 0, nt = 0;
  // the row's counts: up to four words ALL loaded before the first is used and kept for the loop below (a rolled loop
  // waited for every load before it issued the next, twice: eight dependent round trips per block of rows)
  const bool few = D.wz <= 4;  // (wave-uniform)
  unsigned c4[4] = {0u, 0u, 0u, 0u};
  if (row < n_rows) {
    if (few) {
#pragma unroll
      for (int k = 0; k < 4; ++k) c4[k] = k < D.wz ? cnt[(size_t)row * D.wz + k] : 0u;
#pragma unroll
      for (int k = 0; k < 4; ++k) { na += c4[k] ? 1u : 0u; nv += c4[k] & 0xFFFFu; nt += c4[k] >> 16; }
    } else {
      for (int k = 0; k < D.wz; ++k) {
        const unsigned c = cnt[(size_t)row * D.wz + k];
        na += c ? 1u : 0u; nv += c & 0xFFFFu; nt += c >> 16;
      }
    }
  }
  const u64 mine = pack3(na, nv, nt);
  const u64 ex = wave_incl_scan(mine) - mine;
  if (row >= n_rows) continue;
  const int* sg = seg + 3 * (blkid >> 8);
  int ci = sg[0] + blk[3 * blkid] + (int)(ex & 0xFFFFF);
  int vb = sg[1] + blk[3 * blkid + 1] + (int)((ex >> 20) & 0xFFFFF);
  int tb = sg[2] + blk[3 * blkid + 2] + (int)((ex >> 40) & 0xFFFFF);
  const int x = mc_x_of(D, row), y = row - x * D.ny;
  for (int k = 0; k < D.wz; ++k) {
    const int w = row * D.wz + k;
    const unsigned c = few ? (k == 0 ? c4[0] : k == 1 ? c4[1] : k == 2 ? c4[2] : c4[3]) : cnt[w];
    int mine_ci = -1;
    if (c) {
      if (ci < cap_rec) {
        const mc_masks M = mc_load(bits, D, x, y, k);
        mc_rec r;
        r.w = w;
        r.vbase = vb;
        r.tbase = tb;
        r.ex = M.ex; r.ey = M.ey; r.ez = M.ez;
        // triangles of the word (k_mc_emit_batch: the extent of a batch) | its centre vertices << 16
        r.pad = (int)((c >> 16) | (((c & 0xFFFFu) - (unsigned)(__popcll(M.ex) + __popcll(M.ey) + __popcll(M.ez))) << 16));
        rec[ci] = r;
        mine_ci = ci;
      }
      ++ci; vb += (int)(c & 0xFFFFu); tb += (int)(c >> 16);
    }
    cmap[w] = mine_ci;
  }
  }
  LT_MC_STAMP_AT(2, wave, 2, wall_clock64());
}

// ---- emission ----------------------------------------------------------------------------------------------------------
struct mc_nb { u64 ex, ey, ez; int vbase; int have; };

// float32 vertex coordinate along the edge from lattice coordinate c (value v1) to c + 1 (value v2): scikit-image's
// centre-of-mass rule (Cell._add_face_from_edge_index), evaluated in double, stored as float32
// (its `FLT_EPSILON` is, despite the name, `np.spacing(1.0)` = 2^-52 -- a double; C's FLT_EPSILON, 1.19e-7, would move
// every vertex by ~1e-7 voxel and exact-zero samples visibly.  Pinned: golden F10 of the real scikit-image, bit for bit.)
__device__ __forceinline__ float mc_edge_coord(int c, float v1, float v2) {
  const double w1 = 1.0 / (LT_MC_EPS + fabs((double)v1));
  const double w2 = 1.0 / (LT_MC_EPS + fabs((double)v2));
  return (float)((double)c + w2 / (w1 + w2));
}

// ---- k_mc_emit_batch: one wave per K consecutive active words, one lane per VERTEX / per TRIANGLE ----------------------------
// A wave per active word with a lane per voxel (rounds 1-2) spent its time issuing instructions for idle lanes: on the default volume's street scene an active word owns
// 3.6 vertices and 7.3 triangles, so the double-precision vertex rule (three IEEE divisions), unrolled over the three edge
// axes, and the 5 x 3 unrolled index computations run with 2-4 of 64 lanes live -- 252 000 waves x ~3 300 cycles = the
// kernel's 265 us (more words per wave or a lane per word change nothing: 272 / 332 us measured).  Here a wave takes K
// words, lists their vertices and triangles in LDS (in output order: the lists ARE the output ranges, a batch's words
// are consecutive), and then lane j computes vertex j / triangle j: every lane live, every store coalesced.
// LDS decides how many batches a CU holds, in granules of 1 280 B per WORKGROUP (tools/occ_probe.hip fine): the 30 224 B of
// four waves are 24 granules = 5 workgroups = 20 waves per CU (one-wave workgroups of 7 752 B held 18).  Six workgroups
// (<= 26 880 B and <= 80 registers: lists of 128, the tiling table back in global memory) were built and measured: 90 us
// against 80 -- a batch is ~1 400 instructions (924 vector: staging 155, cells 222, vertices 283, triangles 263,
// tools/mc_sections.sh; 362 scalar, 102 LDS) and ONE wave issues an instruction every 5-7 cycles whatever runs beside it
// (profiles/r03/valu_calib.txt: 4.7 cycles alone, 7.2 with eight waves per SIMD): 3-4 us of a batch's 8.4 are its own
// instruction stream, which more waves do not shorten and fewer registers (= more instructions) lengthen.
#define LT_MC_VCAP 256   // list windows; a batch with more vertices / triangles is emitted in several passes
#define LT_MC_TCAP 256
#define LT_MC_EW 4  // waves per workgroup of the emission
#define LT_MC_WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
// AOS: the three field pointers are one array of (tsdf, weight, colour, remission) records (a TSDF volume, D.vs == 4): an
// edge vertex's two field samples and its attributes -- which are those of one of the edge's two ends -- come with TWO
// 16-byte loads, nothing waits for the vertex rule (separate arrays: four loads, the last two behind the rule)
template <int K, bool AOS>
__global__ __launch_bounds__(64 * LT_MC_EW) void k_mc_emit_batch(const float* __restrict__ tsdf, const float* __restrict__ color_vol,
                                                      const float* __restrict__ rem_vol, const u64* __restrict__ bits,
                                                      mc_dims D, const int* __restrict__ cmap,
                                                      const mc_rec* __restrict__ rec, int n_active, float voxel_size,
                                                      float ox, float oy, float oz, float* __restrict__ verts,
                                                      int* __restrict__ faces, int* __restrict__ colors,
                                                      float* __restrict__ rem, int cap_v, int cap_f, mc_amb A, int xcd_map) {
  static_assert(K <= 16, "list entries hold the word in 4 bits");
  static_assert(64 * LT_MC_EW == 256, "the tiling table is staged by 256 threads");
  // LT_MC_EW waves per workgroup, each with arrays of its own and no business with the others: what would be a workgroup
  // barrier orders ONE wave's LDS accesses, which the hardware executes in order anyway -- LT_MC_WSYNC only keeps the
  // compiler from moving them.  (Four waves instead of one per workgroup change nothing by themselves -- 82.7 us both;
  // they share the tiling table below, 256 B of LDS per wave instead of 1 KB: 80 us; 75 on a volume of records, AOS.)
  __shared__ mc_rec S_rec[LT_MC_EW][K];
  __shared__ int S_xyz[LT_MC_EW][K][3];       // x, y, wz of the words
  __shared__ size_t S_base[LT_MC_EW][K];      // ... and the voxel index of their first voxel
  __shared__ mc_nb S_nb[LT_MC_EW][K][8];      // records of the 8 words a cell's triangles can reference
  __shared__ u64 S_cm[LT_MC_EW][K][9];        // corner masks m[dx][dy], s[dx][dy] (mc_masks) and the active-cell mask of the words
  // sg: sign words of the cell corners, [dx | dy << 1 | dw << 2] -- dead once the corner masks are built, where
  // vl: vertex j of the window: k | b << 4 | axis << 10 (12 bits; axis 3 = the cell's centre vertex) begins its life
  union sg_vl { u64 sg[K][8]; unsigned short vl[LT_MC_VCAP]; };
  __shared__ sg_vl S_sv[LT_MC_EW];
  __shared__ unsigned short S_tl[LT_MC_EW][LT_MC_TCAP];  // triangle j of the window: index into s_cl | t << 10
  __shared__ unsigned S_cl[LT_MC_EW][K * 64];      // active cells of the batch in order: k | b << 4 | triangles << 10 | centre << 14 | tiling offset << 15
  __shared__ unsigned short S_ct[LT_MC_EW][K * 64];  // ... and the (batch-relative) index of their first triangle (< 16 x 64 x 12)
  __shared__ u64 S_ccm[LT_MC_EW][K];               // the words' cells that own a centre vertex
  __shared__ int S_cpre[LT_MC_EW][K + 1];          // active cells before word k
  __shared__ unsigned s_lwc[256];  // LT_LWC_FIXED, once per workgroup: a cell's tiling without a global round trip
  s_lwc[threadIdx.x & 255u] = LT_LWC_FIXED[threadIdx.x & 255u];
  __syncthreads();  // (the only workgroup barrier: before any wave's first batch)
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  auto& s_rec = S_rec[wv]; auto& s_xyz = S_xyz[wv]; auto& s_base = S_base[wv]; auto& s_sg = S_sv[wv].sg; auto& s_nb = S_nb[wv]; auto& s_cm = S_cm[wv];
  auto& s_vl = S_sv[wv].vl; auto& s_tl = S_tl[wv]; auto& s_cl = S_cl[wv]; auto& s_ct = S_ct[wv]; auto& s_ccm = S_ccm[wv];
  auto& s_cpre = S_cpre[wv];
  const int lane = (int)(threadIdx.x & 63u);
  // (a wave takes batches until none is left; by default the grid has a wave per batch -- see the launch)
  const int n_batches = (n_active + K - 1) / K;
  // Batch order: active words are numbered x-major, so the batch list walks the volume plane by plane.  Workgroups are dealt
  // to the eight XCDs round-robin (workgroup id % 8) and the L2s are private: with `xcd_map` XCD q takes the q-th EIGHTH of
  // the batch list (its waves in turn within it), so that the columns (x + 1, y) / (x, y + 1) a batch samples are the OWN
  // columns of batches the same L2 serves moments later -- instead of every plane's lines being pulled into all eight L2s.
  // xcd_map = C > 0: the list is cut into CHUNKS of C batches (a few planes), chunk c belongs to XCD c % 8, and the waves of
  // an XCD take the batches of its chunks in turn -- eight contiguous eighths (one per XCD) moved the fewest bytes but the
  // XCDs finished at different times (a batch costs 1.5 - 42 us depending on where in the scene it lies).
  const int per_xcd = (int)(gridDim.x >> 3) * LT_MC_EW, xq = blockIdx.x & 7, xslot = (int)(blockIdx.x >> 3) * LT_MC_EW + wv;
  if (xcd_map && xslot >= per_xcd) return;  // (a grid that is not a multiple of eight: the tail waves have no share)
  for (int it = xcd_map ? xslot : (int)blockIdx.x * LT_MC_EW + wv;; it += xcd_map ? per_xcd : (int)gridDim.x * LT_MC_EW) {
    int bi = it;
    if (xcd_map) bi = ((it / xcd_map) * 8 + xq) * xcd_map + it % xcd_map;
    if (xcd_map ? (it / xcd_map) * 8 * xcd_map >= n_batches : bi >= n_batches) break;
    if (bi >= n_batches) continue;
#ifdef LT_MC_STAMP
  unsigned long long st_last = wall_clock64();
#endif
  const int ci0 = bi * K;
  const int nw = min(K, n_active - ci0);
  const size_t sy = (size_t)D.nz, sx = (size_t)D.ny * D.nz;
  if (lane < nw) {
    const mc_rec r = rec[ci0 + lane];
    s_rec[lane] = r;
    const int row = mc_row_of(D, r.w), x = mc_x_of(D, row);
    s_xyz[lane][2] = r.w - row * D.wz;
    s_xyz[lane][0] = x;
    s_xyz[lane][1] = row - x * D.ny;
    s_base[lane] = (size_t)x * sx + (size_t)(row - x * D.ny) * sy + (size_t)(r.w - row * D.wz) * 64;
  }
  LT_MC_WSYNC();
  LT_MC_SECTION(0);  // records
  for (int p = lane; p < 8 * nw; p += 64) {  // (k, slot): sign word and record of the word (x + dx, y + dy, wz + dw)
    const int k = p >> 3, slot = p & 7;
    const int x = s_xyz[k][0], y = s_xyz[k][1], wz = s_xyz[k][2];
    const int dx = slot & 1, dy = (slot >> 1) & 1, dw = slot >> 2;
    const bool have = x + dx < D.nx && y + dy < D.ny && wz + dw < D.wz;
    const int w2 = ((x + dx) * D.ny + (y + dy)) * D.wz + wz + dw;
    u64 sg = 0ull;
    int c2 = -1;
    if (have) {
      sg = bits[w2];
      c2 = slot == 0 ? ci0 + k : cmap[w2];
    }
    mc_nb e;
    e.ex = e.ey = e.ez = 0; e.vbase = 0; e.have = 0;
    if (c2 >= 0) {
      const mc_rec r2 = rec[c2];
      e.ex = r2.ex; e.ey = r2.ey; e.ez = r2.ez; e.vbase = r2.vbase; e.have = 1;
    }
    s_sg[k][slot] = sg;
    s_nb[k][slot] = e;
  }
  LT_MC_WSYNC();
  LT_MC_SECTION(1);  // sign words, compact indices, neighbour records
  if (lane < nw) {  // the cell masks of word `lane`, once
    u64 w8[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) w8[q] = s_sg[lane][q];
    const mc_masks M = mc_build(w8, D, s_xyz[lane][0], s_xyz[lane][1], s_xyz[lane][2]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      s_cm[lane][q] = M.m[q & 1][q >> 1];
      s_cm[lane][4 + q] = M.s[q & 1][q >> 1];
    }
    s_cm[lane][8] = M.ac;
  }
  {  // active cells before each word (lanes 0 .. nw - 1 hold the words' counts)
    int c = lane < nw ? __popcll(s_cm[lane][8]) : 0;  // (own write: same lane)
    const int mine = c;
#pragma unroll
    for (int o = 1; o < K; o <<= 1) {
      const int q = __shfl_up(c, o, 64);
      if (lane >= o) c += q;
    }
    if (lane < nw) s_cpre[lane + 1] = c;
    if (lane == 0) s_cpre[0] = 0;
    if (lane < K) s_ccm[lane] = 0ull;
    (void)mine;
  }
  LT_MC_WSYNC();
  const mc_rec first = s_rec[0], last = s_rec[nw - 1];
  const int vbase0 = first.vbase, tbase0 = first.tbase;
  const int nvt = last.vbase + __popcll(last.ex) + __popcll(last.ey) + __popcll(last.ez) + (last.pad >> 16) - vbase0;
  const int ntt = last.tbase + (last.pad & 0xFFFF) - tbase0;
#if defined(LT_MC_STOP) && LT_MC_STOP == 1  // instruction-count experiment (tools/mc_sections.sh): staging only
  continue;
#endif
  // ---- the batch's active cells in order (word, z) with their TILING, and the (batch-relative) index of every cell's first
  // triangle: a lane takes one (word, SEGMENT of 8 voxels) pair, lists the cells of its 8 voxels (the word's cell base + a
  // popcount below the segment) -- case index from the corner masks, tiling from the table or, for Lewiner's ambiguous cases,
  // from k_mc_amb's side table (the tests on the cell's eight values ran there) -- and adds up their triangle counts; ONE wave scan over the lanes
  // turns the sums into offsets, and the lane hands them to its cells.  The cells whose tiling has a centre vertex are
  // collected per word (s_ccm): their vertices follow the word's edge vertices in the output.
  const int ncell = s_cpre[nw];  // (wave-uniform)
  constexpr int NPV = (8 * K + 63) / 64;
  {
    int run = 0;  // triangles of the pairs before this round of 64
#pragma unroll
    for (int i = 0; i < NPV; ++i) {
      const int p = lane + 64 * i, k = p >> 3, seg = p & 7;
      unsigned ac8 = 0, ntp = 0;  // ntp: the triangle counts of the segment's cells, 4 bits each, in cell order
      int c0 = 0, tsum = 0;
      if (k < nw) {
        const u64 ac = s_cm[k][8];
        ac8 = (unsigned)(ac >> (8 * seg)) & 255u;
        if (ac8) {
          c0 = s_cpre[k] + __popcll(ac & ((1ull << (8 * seg)) - 1ull));
          unsigned m8[8];  // the corner masks' bits of this segment
#pragma unroll
          for (int q = 0; q < 8; ++q) m8[q] = (unsigned)(s_cm[k][q] >> (8 * seg)) & 255u;
          int c = c0, sh = 0;
          unsigned cm8 = 0;
          // the tilings of the segment's (up to eight) cells: ALL table look-ups first, independent of one another -- a loop
          // over the set bits with the look-up inside was a chain of dependent global loads, one per cell of the fullest lane
          // (the section's 3.0 of a batch's 9.4 us, tools/mc_wave_times.py)
          // case index of voxel bb: bit q = bit bb of corner mask q (mc_case: corner q = dx | dy << 1 | dz << 2) -- the
          // transpose of an 8 x 8 bit matrix (rows = the masks' bytes), three swap steps on a 64-bit word instead of 64
          // single-bit moves
          u64 tm = 0ull;
#pragma unroll
          for (int q = 0; q < 8; ++q) tm |= (u64)m8[q] << (8 * q);
          u64 tt = (tm ^ (tm >> 7)) & 0x00AA00AA00AA00AAull;
          tm ^= tt ^ (tt << 7);
          tt = (tm ^ (tm >> 14)) & 0x0000CCCC0000CCCCull;
          tm ^= tt ^ (tt << 14);
          tt = (tm ^ (tm >> 28)) & 0x00000000F0F0F0F0ull;
          tm ^= tt ^ (tt << 28);
          unsigned selv[8];
#pragma unroll
          for (int bb = 0; bb < 8; ++bb) {
            const unsigned cs = (unsigned)(tm >> (8 * bb)) & 255u;
            selv[bb] = ((ac8 >> bb) & 1u) ? s_lwc[cs] : 0u;
          }
#pragma unroll
          for (int bb = 0; bb < 8; ++bb) {
            if (!((ac8 >> bb) & 1u)) continue;
            unsigned sel = selv[bb];
            if (sel == 0xFFFFFFFFu)  // an ambiguous case: the tiling k_mc_amb filed under the cell's voxel index
              sel = amb_lookup(A, (unsigned)((s_xyz[k][0] * D.ny + s_xyz[k][1]) * D.nz + s_xyz[k][2] * 64 + 8 * seg + bb));
            const unsigned nt = LW_NT(sel);
            s_cl[c] = (unsigned)k | ((unsigned)(8 * seg + bb) << 4) | (nt << 10) | (LW_C(sel) << 14) | (LW_OFF(sel) << 15);
            cm8 |= LW_C(sel) << bb;
            ntp |= nt << sh;
            tsum += (int)nt;
            ++c; sh += 4;
          }
          if (cm8) atomicOr((unsigned long long*)&s_ccm[k], (unsigned long long)cm8 << (8 * seg));
        }
      }
      int inc = tsum;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int q = __shfl_up(inc, o, 64);
        if (lane >= o) inc += q;
      }
      int t0 = run + inc - tsum;
      int c = c0;
      for (unsigned a8 = ac8; a8; a8 &= a8 - 1u, ++c, ntp >>= 4) {
        s_ct[c] = (unsigned short)t0;
        t0 += (int)(ntp & 15u);
      }
      run += __shfl(inc, 63, 64);
    }
  }
  LT_MC_WSYNC();
  LT_MC_SECTION(5);  // cell masks, cell list, triangle offsets
#if defined(LT_MC_STOP) && LT_MC_STOP == 4  // ... + cell list and scan
  continue;
#endif
  // ---- vertices
  // Listing: a lane takes one (word, SEGMENT of 8 voxels) pair -- 8 K pairs, one per lane for K = 8 --, finds where its
  // segment's vertices start in the batch's output range (the word's base + three popcounts below the segment: no search,
  // no scan) and walks its own voxels.  A street scene's segment holds 0.45 vertices on average and 24 at most: the walk
  // is short where the batch is sparse and costs what a lane-per-voxel pass costs where it is dense (walls along z), so one
  // path serves both -- the lane-per-ITEM search it replaces (word by the bases, voxel by a 6-step binary search over
  // three 64-bit popcounts) was 277 vector instructions per batch.
  for (int vb = 0; vb < nvt; vb += LT_MC_VCAP) {
#pragma unroll
    for (int i = 0; i < NPV; ++i) {
      const int p = lane + 64 * i, k = p >> 3, seg = p & 7;
      if (k < nw) {
        const mc_rec R = s_rec[k];
        const unsigned ex8 = (unsigned)(R.ex >> (8 * seg)) & 255u, ey8 = (unsigned)(R.ey >> (8 * seg)) & 255u,
                       ez8 = (unsigned)(R.ez >> (8 * seg)) & 255u;
        unsigned any = ex8 | ey8 | ez8;
        if (any) {
          const u64 below = (1ull << (8 * seg)) - 1ull;
          int j = R.vbase - vbase0 - vb + __popcll(R.ex & below) + __popcll(R.ey & below) + __popcll(R.ez & below);
          const unsigned e0 = (unsigned)k | ((unsigned)(8 * seg) << 4);
          for (; any; any &= any - 1u) {
            const int bb = __ffs((int)any) - 1;
            const unsigned e = e0 + ((unsigned)bb << 4);
            if ((ex8 >> bb) & 1u) { if ((unsigned)j < LT_MC_VCAP) s_vl[j] = (unsigned short)e; ++j; }
            if ((ey8 >> bb) & 1u) { if ((unsigned)j < LT_MC_VCAP) s_vl[j] = (unsigned short)(e | (1u << 10)); ++j; }
            if ((ez8 >> bb) & 1u) { if ((unsigned)j < LT_MC_VCAP) s_vl[j] = (unsigned short)(e | (2u << 10)); ++j; }
          }
        }
        // the centre vertices of the segment's cells: after all edge vertices of the word, in cell order
        const u64 ccm = s_ccm[k];
        unsigned cm8 = (unsigned)(ccm >> (8 * seg)) & 255u;
        if (cm8) {
          int j = R.vbase - vbase0 - vb + __popcll(R.ex) + __popcll(R.ey) + __popcll(R.ez) + __popcll(ccm & ((1ull << (8 * seg)) - 1ull));
          for (; cm8; cm8 &= cm8 - 1u, ++j) {
            const int bb = __ffs((int)cm8) - 1;
            if ((unsigned)j < LT_MC_VCAP) s_vl[j] = (unsigned short)((unsigned)k | ((unsigned)(8 * seg + bb) << 4) | (3u << 10));
          }
        }
      }
    }
    LT_MC_WSYNC();
    LT_MC_SECTION(3);  // vertex list
#if defined(LT_MC_STOP) && LT_MC_STOP == 2  // ... + vertex list
    continue;
#endif
    const int nwin = min(LT_MC_VCAP, nvt - vb);
    for (int j = lane; j < nwin; j += 64) {
      const unsigned e = s_vl[j];
      const int k = e & 15, b = (e >> 4) & 63, a = (e >> 10) & 3;
      const int x = s_xyz[k][0], y = s_xyz[k][1], z = s_xyz[k][2] * 64 + b;
      const size_t i = s_base[k] + (size_t)b;  // (64-bit multiplications are quarter rate: once per word, not per vertex)
      float p0 = (float)x, p1 = (float)y, p2 = (float)z;
      constexpr size_t VS = AOS ? 4 : 1;
      float rgb, rm;
      if (a == 3) {
        // the cell's centre vertex (Cell.calculate_center_vertex): the centre of mass of the eight corners with weights
        // 1 / (eps + |v|), summed in Lewiner's corner order (0 .. 7) in double, stored as float32
        const float* c = tsdf + i * VS;
        double ff = 0.0, f0 = 0.0, f1 = 0.0, f2 = 0.0;
#pragma unroll 1  // (rare: a rolled loop keeps the kernel's registers where they were without the centre vertex)
        for (int q = 0; q < 8; ++q) {  // Lewiner's corner q sits at (a0, a1, a2) = (q >> 2, q >> 1 & 1, (q ^ q >> 1) & 1)
          const int a0 = q >> 2, a1 = (q >> 1) & 1, a2 = (q ^ (q >> 1)) & 1;
          const double w = 1.0 / (LT_MC_EPS + fabs((double)c[((size_t)a0 * sx + (size_t)a1 * sy + (size_t)a2) * VS]));
          ff += w;
          if (a2) f2 += w;
          if (a1) f1 += w;
          if (a0) f0 += w;
        }
        p0 = (float)((double)x + f0 / ff); p1 = (float)((double)y + f1 / ff); p2 = (float)((double)z + f2 / ff);
        // verts_ind = np.round(verts).astype(int) on the float32 coordinates (fusion_lidar.py:409).
        // The vertex lies inside a cell (below: on an edge) that starts at voxel (x, y, z): every rounded coordinate is
        // the voxel's or the next one's, and the next one exists where the cell / the edge does.  (Anything else -- a NaN
        // field value, on which numpy would raise -- counts as "the next one": never a wild address.)
        const size_t jj = i + ((int)rintf(p0) == x ? (size_t)0 : sx) + ((int)rintf(p1) == y ? (size_t)0 : sy) +
                          ((int)rintf(p2) == z ? (size_t)0 : (size_t)1);
        rgb = color_vol[jj * VS];
        rm = rem_vol[jj * VS];
      } else {
        const size_t i1 = i + (a == 0 ? sx : (a == 1 ? sy : (size_t)1));
        const int ce = a == 0 ? x : (a == 1 ? y : z);
        if (AOS) {
          const float4 q0 = reinterpret_cast<const float4*>(tsdf)[i], q1 = reinterpret_cast<const float4*>(tsdf)[i1];
          const float pe = mc_edge_coord(ce, q0.x, q1.x);
          if (a == 0) p0 = pe; else if (a == 1) p1 = pe; else p2 = pe;
          const bool far_end = (int)rintf(pe) != ce;
          rgb = far_end ? q1.z : q0.z;
          rm = far_end ? q1.w : q0.w;
        } else {
          const float pe = mc_edge_coord(ce, tsdf[i], tsdf[i1]);
          if (a == 0) p0 = pe; else if (a == 1) p1 = pe; else p2 = pe;
          const size_t jj = (int)rintf(pe) != ce ? i1 : i;
          rgb = color_vol[jj];
          rm = rem_vol[jj];
        }
      }
      const int vid = vbase0 + vb + j;
      if (vid < cap_v) {
        verts[3 * (size_t)vid] = p0 * voxel_size + ox;  // verts * voxel_size + vol_origin in float32 (:412)
        verts[3 * (size_t)vid + 1] = p1 * voxel_size + oy;
        verts[3 * (size_t)vid + 2] = p2 * voxel_size + oz;
        // colour unfolding (:419-423) in float32, .astype(np.uint8) = truncation to 8 bits
        const float cb = floorf(rgb / (float)(256 * 256));
        const float cg = floorf((rgb - cb * 256.0f * 256.0f) / 256.0f);
        const float cr = rgb - cb * 256.0f * 256.0f - cg * 256.0f;
        colors[3 * (size_t)vid] = (int)floorf(cr) & 255;
        colors[3 * (size_t)vid + 1] = (int)floorf(cg) & 255;
        colors[3 * (size_t)vid + 2] = (int)floorf(cb) & 255;
        rem[vid] = rm;
      }
    }
    LT_MC_WSYNC();
  }
  LT_MC_SECTION(4);  // vertex pass (field samples, attributes, stores drained)
#if defined(LT_MC_STOP) && LT_MC_STOP == 3  // ... + vertex pass
  continue;
#endif
  // ---- triangles (cells exist where x + 1 < nx, y + 1 < ny, z + 1 < nz): their cells were listed above
  for (int tb = 0; tb < ntt; tb += LT_MC_TCAP) {
    for (int c = lane; c < ncell; c += 64) {  // a cell's (up to twelve) triangles into the window's list
      const int nt = (int)((s_cl[c] >> 10) & 15u);
      const int j0 = (int)s_ct[c] - tb;
      for (int t = 0; t < nt; ++t)
        if ((unsigned)(j0 + t) < LT_MC_TCAP) s_tl[j0 + t] = (unsigned short)((unsigned)c | ((unsigned)t << 10));
    }
    LT_MC_WSYNC();
    const int nwin = min(LT_MC_TCAP, ntt - tb);
    for (int j = lane; j < nwin; j += 64) {
      const unsigned tl = s_tl[j];
      const unsigned e = s_cl[tl & 1023u];
      const int k = e & 15, b = (e >> 4) & 63, t = (int)(tl >> 10);
      // the triangle's three codes in one 16-bit word (already in the reference's order: np.fliplr(faces)); three byte loads
      // with a branch on each were three dependent round trips per triangle window
      const unsigned codes = LT_LWF3[(e >> 15) / 3u + (unsigned)t];
      const int tid = tbase0 + tb + j;
      int id[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int code = (int)((codes >> (5 * q)) & 31u);
        if (code == 31) {  // the cell's centre vertex: behind the word's edge vertices, ranked among the word's centre cells
          const mc_rec& Rk = s_rec[k];
          id[q] = Rk.vbase + __popcll(Rk.ex) + __popcll(Rk.ey) + __popcll(Rk.ez) + __popcll(s_ccm[k] & ((1ull << b) - 1ull));
          continue;
        }
        const int c0 = code & 7, ax = code >> 3;
        int b2 = b + ((c0 >> 2) & 1), slot = c0 & 3;
        if (b2 == 64) { b2 = 0; slot |= 4; }
        const mc_nb nbe = s_nb[k][slot];
        const u64 l2 = (1ull << b2) - 1ull;
        int v = nbe.vbase + __popcll(nbe.ex & l2) + __popcll(nbe.ey & l2) + __popcll(nbe.ez & l2);
        if (ax > 0) v += (int)((nbe.ex >> b2) & 1ull);
        if (ax > 1) v += (int)((nbe.ey >> b2) & 1ull);
        id[q] = v;
      }
      if (tid < cap_f) {
        faces[3 * (size_t)tid] = id[0];
        faces[3 * (size_t)tid + 1] = id[1];
        faces[3 * (size_t)tid + 2] = id[2];
      }
    }
    LT_MC_WSYNC();
  }
    LT_MC_SECTION(6);  // triangle list + pass (stores drained)
    LT_MC_WSYNC();  // the batch's arrays are reused
  }
}

// ---- host ------------------------------------------------------------------------------------------------------------
extern "C" int lt_mesh_create(lt_mesh** out, int device) {
  if (!out) {
    lt_set_error("lt_mesh_create: NULL out pointer");
    return LT_ERR_INVALID_ARG;
  }
  *out = nullptr;
  if (device < 0) LT_HIP(hipGetDevice(&device));
  LT_HIP(hipSetDevice(device));
  lt_mesh* m = (lt_mesh*)calloc(1, sizeof(lt_mesh));
  if (!m) return LT_ERR_NO_MEMORY;
  m->device = device;
  if (hipHostMalloc((void**)&m->totals_host, 4 * sizeof(int), hipHostMallocDefault) != hipSuccess) {
    lt_set_error("lt_mesh_create: hipHostMalloc failed");
    free(m);
    return LT_ERR_NO_MEMORY;
  }
  for (int k = 0; k < 3; ++k) (void)hipEventCreate(&m->ev[k]);
  *out = m;
  return LT_OK;
}

extern "C" int lt_mesh_destroy(lt_mesh* m) {
  if (!m) return LT_OK;
  (void)hipSetDevice(m->device);
  (void)hipDeviceSynchronize();
  void* ps[] = {m->verts, m->faces, m->colors, m->rem, m->bits, m->cnt, m->cmap, m->blk, m->rec, m->wave_na, m->amb.queue, m->amb.table, m->amb.counter,
                m->verts2, m->colors2, m->rem2, m->rn_first, m->rn_newid, m->rn_bsum};
  for (void* p : ps)
    if (p) (void)hipFree(p);
  if (m->totals_host) (void)hipHostFree(m->totals_host);
  for (int k = 0; k < 3; ++k)
    if (m->ev[k]) (void)hipEventDestroy(m->ev[k]);
  free(m);
  return LT_OK;
}

extern "C" int lt_mesh_get(lt_mesh* m, int* n_verts, int* n_faces, float** verts, int** faces, int** colors,
                           float** rem) {
  if (!m) {
    lt_set_error("lt_mesh_get: NULL mesh");
    return LT_ERR_INVALID_ARG;
  }
  if (n_verts) *n_verts = m->n_verts;
  if (n_faces) *n_faces = m->n_faces;
  if (verts) *verts = m->verts;
  if (faces) *faces = m->faces;
  if (colors) *colors = m->colors;
  if (rem) *rem = m->rem;
  return LT_OK;
}

template <class T>
static int mc_grow(T** p, size_t* cap, size_t need) {
  if (need <= *cap && *p) return LT_OK;
  if (*p) {
    LT_HIP(hipDeviceSynchronize());
    (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
  }
  const size_t n = need + need / 4 + 1024;
  LT_HIP(hipMalloc((void**)p, n * sizeof(T)));
  *cap = n;
  return LT_OK;
}

static int mc_extract(const float* tsdf, const float* color_vol, const float* rem_vol, int vs, int nx, int ny, int nz,
                      float voxel_size, const float* origin, lt_mesh* m, void* stream_, float* ms,
                      const unsigned* col_epoch, unsigned epoch, const u64* ext_bits, const unsigned* chunk_epoch) {
  if (!tsdf || !color_vol || !rem_vol || !origin || !m || nx <= 0 || ny <= 0 || nz <= 0) {
    lt_set_error("lt_marching_cubes_dev: invalid argument");
    return LT_ERR_INVALID_ARG;
  }
  if ((double)nx * ny * nz >= 2147483647.0) {
    lt_set_error("lt_marching_cubes_dev: more than 2^31 - 1 voxels");
    return LT_ERR_TOO_LARGE;
  }
  hipStream_t stream = (hipStream_t)stream_;
  LT_HIP(hipSetDevice(m->device));
  if (nx < 2 || ny < 2 || nz < 2) {  // no cell (scikit-image: "Input array must be at least 2x2x2"): the empty mesh
    m->n_verts = 0;
    m->n_faces = 0;
    if (ms) ms[0] = ms[1] = 0.f;
    return LT_OK;
  }
  mc_dims D;
  D.nx = nx; D.ny = ny; D.nz = nz; D.vs = vs;
  D.wz = (nz + 63) / 64;
  mc_magic((unsigned)D.wz, &D.wz_m, &D.wz_s);
  mc_magic((unsigned)D.ny, &D.ny_m, &D.ny_s);
  const size_t n_words = (size_t)nx * ny * D.wz;
  if (n_words >= 2147483647ull) {
    lt_set_error("lt_marching_cubes_dev: volume too large");
    return LT_ERR_TOO_LARGE;
  }
  D.n_words = (int)n_words;
  const int n_rows = nx * ny;
  const int n_blocks = ((n_rows + 255) / 256) * 4;  // one block of the scan per wave of 64 rows
  if (D.wz > 21) {  // (the packed per-block totals hold 20 bits a field: 64 rows x wz words x 768 triangles)
    lt_set_error("lt_marching_cubes_dev: nz = %d exceeds 1344", nz);
    return LT_ERR_TOO_LARGE;
  }
  if (n_words > m->cap_words || !m->bits) {
    if (m->bits) {
      LT_HIP(hipDeviceSynchronize());
      (void)hipFree(m->bits); (void)hipFree(m->cnt); (void)hipFree(m->cmap);
      m->bits = nullptr; m->cnt = nullptr; m->cmap = nullptr;
      m->cap_words = 0;
    }
    LT_HIP(hipMalloc((void**)&m->bits, (n_words + 1) * sizeof(u64)));
    LT_HIP(hipMalloc((void**)&m->cnt, n_words * sizeof(unsigned)));
    LT_HIP(hipMalloc((void**)&m->cmap, n_words * sizeof(int)));
    m->cap_words = n_words;
    m->state_dirty = 1;  // (fresh buffers: swept below)
  }
  // the ambiguous cells' queue and tiling table: grown when an extraction queued more cells than fit (below)
  if (!m->amb.queue) {
    const size_t cap = m->amb_cap ? m->amb_cap : (size_t)1 << 10;   // per shard
    size_t tcap = 1;
    while (tcap < 4 * LT_MC_SHARDS * cap) tcap <<= 1;
    const size_t n_ctr = LT_MC_SHARDS * LT_MC_CTR_STRIDE + LT_MC_SHARDS;
    if (hipMalloc((void**)&m->amb.queue, LT_MC_SHARDS * cap * sizeof(uint2)) != hipSuccess || hipMalloc((void**)&m->amb.table, tcap * sizeof(uint2)) != hipSuccess ||
        (!m->amb.counter && hipMalloc((void**)&m->amb.counter, n_ctr * sizeof(int)) != hipSuccess)) {
      (void)hipGetLastError();
      (void)hipFree(m->amb.queue); (void)hipFree(m->amb.table);
      m->amb.queue = nullptr; m->amb.table = nullptr;
      lt_set_error("lt_marching_cubes_dev: hipMalloc of the ambiguous cells' buffers (%zu cells) failed", cap);
      return LT_ERR_NO_MEMORY;
    }
    m->amb_cap = cap;
    m->amb.cap = (unsigned)cap;
    m->amb.tmask = (unsigned)(tcap - 1);
    LT_HIP(hipMemsetAsync(m->amb.counter, 0, n_ctr * sizeof(int), stream));
    LT_HIP(hipMemsetAsync(m->amb.table, 0, tcap * sizeof(uint2), stream));
    m->n_amb_prev = 0;
  }
  if (m->state_dirty) {  // new buffers, or an extraction that did not finish: cnt = 0, cmap = -1 everywhere, once
    LT_HIP(hipMemsetAsync(m->cnt, 0, m->cap_words * sizeof(unsigned), stream));
    LT_HIP(hipMemsetAsync(m->cmap, 0xFF, m->cap_words * sizeof(int), stream));
    LT_HIP(hipMemsetAsync(m->amb.counter, 0, (LT_MC_SHARDS * LT_MC_CTR_STRIDE + LT_MC_SHARDS) * sizeof(int), stream));
    LT_HIP(hipMemsetAsync(m->amb.table, 0, ((size_t)m->amb.tmask + 1) * sizeof(uint2), stream));
    m->n_prev = 0;
    m->n_amb_prev = 0;
  } else if (m->n_prev > 0) {
    hipLaunchKernelGGL(k_mc_clear, dim3((max(m->n_prev, LT_MC_SHARDS * m->n_amb_prev) + 255) / 256), dim3(256), 0, stream, m->rec,
                       m->n_prev, m->cnt, m->cmap, m->amb, m->n_amb_prev);
  }
  m->state_dirty = 1;  // until this extraction has finished
  m->n_prev = 0;
  m->n_amb_prev = 0;
  LT_CHECK(mc_grow(&m->wave_na, &m->cap_wave_na, (size_t)n_blocks));
  const int n_seg = (n_blocks + 255) / 256;
  if (n_seg > 1024) {
    lt_set_error("lt_marching_cubes_dev: volume too large (%d words)", D.n_words);
    return LT_ERR_TOO_LARGE;
  }
  LT_CHECK(mc_grow(&m->blk, &m->cap_blocks, (size_t)3 * n_blocks + 3 * (size_t)n_seg + 4));
  int* seg_dev = m->blk + 3 * (size_t)n_blocks;
  int* totals_dev = seg_dev + 3 * (size_t)n_seg;
  if (ms) LT_HIP(hipEventRecord(m->ev[0], stream));
  // sign bits: the volume's own (kept current by integrate / reset, lt_tsdf.hip) or one pass over the float field
  const u64* bits = ext_bits ? ext_bits : m->bits;
  if (!ext_bits)
    hipLaunchKernelGGL(k_mc_signs, dim3((unsigned)min((size_t)16384, ((size_t)nx * ny + 255) / 256)), dim3(256), 0, stream,
                       tsdf, D, m->bits, col_epoch, epoch);
  if (ms) LT_HIP(hipEventRecord(m->ev[1], stream));
  // (a wave takes LT_MC_BLOCKS_PER_WAVE blocks, see k_mc_words)
  const int sweep_wgs = lt_deal_count(n_blocks, LT_MC_BLOCKS_PER_WAVE, ny, 4) / 4;
  hipLaunchKernelGGL(k_mc_words, dim3(sweep_wgs), dim3(256), 0, stream, m->amb, bits, D, m->cnt, m->blk, ext_bits ? col_epoch : nullptr,
                     epoch, ext_bits ? chunk_epoch : nullptr, m->wave_na, n_blocks);
  // Lewiner's ambiguous cells (1-2 % of a street scene's): tests on their eight values, counts added, tilings filed
  hipLaunchKernelGGL(k_mc_amb, dim3(2 * LT_MC_SHARDS), dim3(256), 0, stream, tsdf, D, m->amb, m->cnt, m->blk, m->wave_na);
  hipLaunchKernelGGL(k_mc_scan1, dim3(n_seg), dim3(256), 0, stream, m->blk, n_blocks, seg_dev);
  hipLaunchKernelGGL(k_mc_scan2, dim3(1), dim3(1024), 0, stream, seg_dev, n_seg, totals_dev, m->amb.counter, (int)m->amb.cap);
  LT_HIP(hipMemcpyAsync(m->totals_host, totals_dev, 4 * sizeof(int), hipMemcpyDeviceToHost, stream));
  LT_HIP(hipStreamSynchronize(stream));  // the one synchronisation: the sizes of the mesh
  if ((size_t)(unsigned)m->totals_host[3] > m->amb_cap) {
    // more ambiguous cells in a shard than its segment holds (white noise: a third of all cells): the counts above are incomplete --
    // larger buffers, and the extraction once more from the start (cnt / cmap are swept: state_dirty is still set)
    const size_t need = (size_t)(unsigned)m->totals_host[3];
    (void)hipFree(m->amb.queue); (void)hipFree(m->amb.table);
    m->amb.queue = nullptr; m->amb.table = nullptr;
    m->amb_cap = need + need / 4 + 1024;
    return mc_extract(tsdf, color_vol, rem_vol, vs, nx, ny, nz, voxel_size, origin, m, stream_, ms, col_epoch, epoch, ext_bits, chunk_epoch);
  }
  const int n_active = m->totals_host[0], nv = m->totals_host[1], nf = m->totals_host[2];
  static const bool dbg_mc = getenv("LIDARHIP_DEBUG_MC") != nullptr;
  if (dbg_mc) fprintf(stderr, "marching cubes: %d active words, %d vertices, %d triangles\n", n_active, nv, nf);
  if (nv == 2147483647 || nf == 2147483647) {
    lt_set_error("lt_marching_cubes_dev: mesh exceeds 2^31 - 1 vertices / faces");
    return LT_ERR_TOO_LARGE;
  }
  if (nf >= LT_MAX_FACES) {
    lt_set_error("lt_marching_cubes_dev: %d faces exceed LT_MAX_FACES", nf);
    return LT_ERR_TOO_LARGE;
  }
  LT_CHECK(mc_grow(&m->rec, &m->cap_rec, (size_t)n_active));
  if (nv > m->cap_v || !m->verts) {
    if (m->verts) {
      LT_HIP(hipDeviceSynchronize());
      (void)hipFree(m->verts); (void)hipFree(m->colors); (void)hipFree(m->rem);
      m->verts = nullptr; m->colors = nullptr; m->rem = nullptr;
    }
    const size_t cap = (size_t)nv + nv / 4 + 1024;
    m->cap_v = 0;  // (a failure below must not leave the old capacity standing over freed / missing buffers)
    if (hipMalloc((void**)&m->verts, cap * 12) != hipSuccess || hipMalloc((void**)&m->colors, cap * 12) != hipSuccess ||
        hipMalloc((void**)&m->rem, cap * 4) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipFree(m->verts); (void)hipFree(m->colors); (void)hipFree(m->rem);
      m->verts = nullptr; m->colors = nullptr; m->rem = nullptr;
      lt_set_error("lt_marching_cubes_dev: hipMalloc of the vertex arrays (%zu vertices) failed", cap);
      return LT_ERR_NO_MEMORY;
    }
    m->cap_v = (int)min(cap, (size_t)2147483647);
  }
  if (nf > m->cap_f || !m->faces) {
    if (m->faces) {
      LT_HIP(hipDeviceSynchronize());
      (void)hipFree(m->faces);
      m->faces = nullptr;
    }
    const size_t cap = (size_t)nf + nf / 4 + 1024;
    m->cap_f = 0;
    LT_HIP(hipMalloc((void**)&m->faces, cap * 12));
    m->cap_f = (int)min(cap, (size_t)2147483647);
  }
  hipLaunchKernelGGL(k_mc_compact, dim3(sweep_wgs), dim3(256), 0, stream, bits, D, m->cnt, m->blk, seg_dev, m->cmap, m->rec,
                     (int)min(m->cap_rec, (size_t)2147483647), m->wave_na, n_blocks);
#define LT_MC_EMIT_ARGS tsdf, color_vol, rem_vol, bits, D, m->cmap, m->rec, n_active, voxel_size, origin[0], origin[1], origin[2], \
                        m->verts, m->faces, m->colors, m->rem, m->cap_v, m->cap_f, m->amb, xcd_map
  if (n_active > 0) {
    static const int kk = []() { const char* e = getenv("LIDARHIP_MC_EMIT_K"); return e ? atoi(e) : 8; }();
    // The grid: persistent waves taking batches in turn (bi += gridDim), 36 per CU -- 1.8 x the resident capacity (20 waves
    // per CU, see LT_MC_VCAP; swept again with four-wave workgroups on the six-workgroup variant: 24 / 36 / 48 / 72 / 96 per
    // CU = 118 / 90 / 100 / 92 / 102 us).  On the default volume's street scene (31 500 batches; a batch lives 8.3 us on
    // average and up to 42, tools/mc_wave_times.py): a wave per batch 96 us -- a wave's start and drain 31 500 times --,
    // 4 608 / 9 216 / 18 432 persistent waves 107 / 87 / 86 us (static dealing is at the mercy of the heavy batches: with
    // one round of waves the slowest wave is the kernel), and persistent waves DRAWING their batches from eight counters
    // (returning atomics, the draw for the next batch issued under the current one) 117-128 us: a returning atomic is a
    // memory-side round trip that the batch's first barrier waits for.  LIDARHIP_MC_EMIT_WAVES=n: n waves; =0: a wave per batch.
    static const int env_waves = []() { const char* e = getenv("LIDARHIP_MC_EMIT_WAVES"); return e ? atoi(e) : -1; }();
    const int dflt_waves = lt_cu_count(m->device) * 36;
    const int max_waves = env_waves > 0 ? env_waves : (env_waves == 0 ? (1 << 24) : dflt_waves);
    // LIDARHIP_MC_EMIT_XCD=0: batches dealt to the waves round-robin over the whole list (rounds 3-4)
    static const int env_xcd = []() { const char* e = getenv("LIDARHIP_MC_EMIT_XCD"); return e ? atoi(e) : 64; }();
    auto grid = [&](int k) { return dim3((unsigned)((min((n_active + k - 1) / k, max_waves) + LT_MC_EW - 1) / LT_MC_EW)); };
    const int xcd_map = (env_xcd > 0 && (int)grid(vs == 4 ? 8 : kk >= 16 ? 16 : kk >= 8 ? 8 : kk >= 4 ? 4 : 2).x * LT_MC_EW >= 64) ? env_xcd : 0;
    if (vs == 4) hipLaunchKernelGGL((k_mc_emit_batch<8, true>), grid(8), dim3(64 * LT_MC_EW), 0, stream, LT_MC_EMIT_ARGS);
    else if (kk >= 16) hipLaunchKernelGGL((k_mc_emit_batch<16, false>), grid(16), dim3(64 * LT_MC_EW), 0, stream, LT_MC_EMIT_ARGS);
    else if (kk >= 8) hipLaunchKernelGGL((k_mc_emit_batch<8, false>), grid(8), dim3(64 * LT_MC_EW), 0, stream, LT_MC_EMIT_ARGS);
    else if (kk >= 4) hipLaunchKernelGGL((k_mc_emit_batch<4, false>), grid(4), dim3(64 * LT_MC_EW), 0, stream, LT_MC_EMIT_ARGS);
    else hipLaunchKernelGGL((k_mc_emit_batch<2, false>), grid(2), dim3(64 * LT_MC_EW), 0, stream, LT_MC_EMIT_ARGS);
  }
#undef LT_MC_EMIT_ARGS
  LT_HIP(hipGetLastError());
  m->n_verts = nv;
  m->n_faces = nf;
  m->n_prev = n_active;  // (the records k_mc_clear will undo before the next extraction)
  m->n_amb_prev = m->totals_host[3];  // (the fullest shard's count: k_mc_clear's extent per shard)
  m->state_dirty = 0;
  if (ms) {
    LT_HIP(hipEventRecord(m->ev[2], stream));
    LT_HIP(hipStreamSynchronize(stream));
    LT_HIP(hipEventElapsedTime(&m->ms_signs, m->ev[0], m->ev[1]));
    LT_HIP(hipEventElapsedTime(&m->ms_rest, m->ev[1], m->ev[2]));
    ms[0] = m->ms_signs;
    ms[1] = m->ms_rest;
  }
  return LT_OK;
}

// ---- lt_mesh_renumber_dev: the vertices numbered as scikit-image numbers them ----------------------------------------
// The extraction above emits scikit-image's FACE STREAM with its own vertex numbers (by word / owner voxel / edge axis).
// scikit-image creates a vertex when a face first references it, so its number is the rank of its first use in ITS face
// stream -- which lists a face's corners in the tiling's order; the reversal (np.fliplr, gradient_direction="descent")
// happens afterwards, so position q = 3 face + k of that stream is corner 2 - k of the face array here (rn_at).
// first[v] = min q (atomicMin), flag(q) = "q is a first use", new number = exclusive prefix sum of the flags at first[v]
// (two-level scan, 1024 positions per block), vertex arrays permuted, faces rewritten.
// For the host-facing get_mesh (the arrays then EQUAL the reference's, golden F10); the in-HBM chain has no use for it.
__device__ __forceinline__ int rn_at(int q) { return q + 2 - 2 * (q % 3); }  // index in the face array of stream position q
__global__ __launch_bounds__(256) void k_rn_first(const int* __restrict__ faces, int n3, int* __restrict__ first) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q < n3) atomicMin(&first[faces[rn_at(q)]], q);
}
__device__ __forceinline__ int rn_block_excl_scan(int v, int* total) {  // 256 threads; returns the exclusive prefix of v
  __shared__ int ws[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int q = __shfl_up(inc, o, 64);
    if (lane >= o) inc += q;
  }
  __syncthreads();  // (ws may still be read by a previous call)
  if (lane == 63) ws[wv] = inc;
  __syncthreads();
  int off = 0;
  for (int w = 0; w < wv; ++w) off += ws[w];
  *total = ws[0] + ws[1] + ws[2] + ws[3];
  return off + inc - v;
}
// stream positions [1024 b, 1024 b + 1024): thread t holds the four consecutive positions 1024 b + 4 t ..
template <bool WRITE>
__global__ __launch_bounds__(256) void k_rn_rank(const int* __restrict__ faces, int n3, const int* __restrict__ first,
                                                 int* __restrict__ bsum, int* __restrict__ newid) {
  const int p0 = blockIdx.x * 1024 + threadIdx.x * 4;
  int v[4], f[4], s = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[j] = p0 + j < n3 ? faces[rn_at(p0 + j)] : -1;
    f[j] = (v[j] >= 0 && first[v[j]] == p0 + j) ? 1 : 0;
    s += f[j];
  }
  int total;
  int ex = rn_block_excl_scan(s, &total);
  if (!WRITE) {
    if (threadIdx.x == 0) bsum[blockIdx.x] = total;
    return;
  }
  ex += bsum[blockIdx.x];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (f[j]) newid[v[j]] = ex;
    ex += f[j];
  }
}
__global__ __launch_bounds__(1024) void k_rn_scan_blocks(int* __restrict__ bsum, int nb) {  // exclusive, in place; total -> bsum[nb]
  __shared__ int carry, ws[16];
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int base = 0; base < nb; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < nb ? bsum[i] : 0;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int q = __shfl_up(inc, o, 64);
      if (lane >= o) inc += q;
    }
    if (lane == 63) ws[wv] = inc;
    __syncthreads();
    int off = carry;
    for (int w = 0; w < wv; ++w) off += ws[w];
    if (i < nb) bsum[i] = off + inc - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry = off + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) bsum[nb] = carry;
}
__global__ __launch_bounds__(256) void k_rn_move(const int* __restrict__ newid, int nv, const float* __restrict__ verts,
                                                 const int* __restrict__ colors, const float* __restrict__ rem,
                                                 float* __restrict__ verts2, int* __restrict__ colors2, float* __restrict__ rem2) {
  const int v = blockIdx.x * 256 + threadIdx.x;
  if (v >= nv) return;
  const int j = newid[v];
  if (j < 0) return;  // (an unreferenced vertex: reported by the host)
#pragma unroll
  for (int k = 0; k < 3; ++k) { verts2[3 * (size_t)j + k] = verts[3 * (size_t)v + k]; colors2[3 * (size_t)j + k] = colors[3 * (size_t)v + k]; }
  rem2[j] = rem[v];
}
__global__ __launch_bounds__(256) void k_rn_faces(int* __restrict__ faces, int n3, const int* __restrict__ newid) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p < n3) faces[p] = newid[faces[p]];
}

// see include/lidarhip.h
extern "C" int lt_mesh_renumber_dev(lt_mesh* m, void* stream_) {
  if (!m) {
    lt_set_error("lt_mesh_renumber_dev: NULL mesh");
    return LT_ERR_INVALID_ARG;
  }
  const int nv = m->n_verts, n3 = 3 * m->n_faces;
  if (nv == 0 || n3 == 0) return LT_OK;
  if ((long long)m->n_faces * 3 > 2147483647ll) {
    lt_set_error("lt_mesh_renumber_dev: more than 2^31 face corners");
    return LT_ERR_TOO_LARGE;
  }
  hipStream_t stream = (hipStream_t)stream_;
  LT_HIP(hipSetDevice(m->device));
  const int nb = (n3 + 1023) / 1024;
  if ((size_t)nv > m->cap_rn || !m->rn_first) {
    (void)hipFree(m->rn_first); (void)hipFree(m->rn_newid);
    m->rn_first = nullptr; m->rn_newid = nullptr; m->cap_rn = 0;
    const size_t cap = (size_t)nv + nv / 4 + 1024;
    LT_HIP(hipMalloc((void**)&m->rn_first, cap * sizeof(int)));
    LT_HIP(hipMalloc((void**)&m->rn_newid, cap * sizeof(int)));
    m->cap_rn = cap;
  }
  if ((size_t)nb + 1 > m->cap_bsum || !m->rn_bsum) {
    (void)hipFree(m->rn_bsum);
    m->rn_bsum = nullptr; m->cap_bsum = 0;
    const size_t cap = (size_t)nb + nb / 4 + 1024;
    LT_HIP(hipMalloc((void**)&m->rn_bsum, cap * sizeof(int)));
    m->cap_bsum = cap;
  }
  if (m->cap_v2 < m->cap_v || !m->verts2) {
    (void)hipFree(m->verts2); (void)hipFree(m->colors2); (void)hipFree(m->rem2);
    m->verts2 = nullptr; m->colors2 = nullptr; m->rem2 = nullptr; m->cap_v2 = 0;
    LT_HIP(hipMalloc((void**)&m->verts2, (size_t)m->cap_v * 12));
    LT_HIP(hipMalloc((void**)&m->colors2, (size_t)m->cap_v * 12));
    LT_HIP(hipMalloc((void**)&m->rem2, (size_t)m->cap_v * 4));
    m->cap_v2 = m->cap_v;
  }
  LT_HIP(hipMemsetAsync(m->rn_first, 0x7F, (size_t)nv * sizeof(int), stream));   // 0x7F7F7F7F: beyond any position
  LT_HIP(hipMemsetAsync(m->rn_newid, 0xFF, (size_t)nv * sizeof(int), stream));   // -1
  hipLaunchKernelGGL(k_rn_first, dim3((n3 + 255) / 256), dim3(256), 0, stream, m->faces, n3, m->rn_first);
  hipLaunchKernelGGL(k_rn_rank<false>, dim3(nb), dim3(256), 0, stream, m->faces, n3, m->rn_first, m->rn_bsum, m->rn_newid);
  hipLaunchKernelGGL(k_rn_scan_blocks, dim3(1), dim3(1024), 0, stream, m->rn_bsum, nb);
  hipLaunchKernelGGL(k_rn_rank<true>, dim3(nb), dim3(256), 0, stream, m->faces, n3, m->rn_first, m->rn_bsum, m->rn_newid);
  hipLaunchKernelGGL(k_rn_move, dim3((nv + 255) / 256), dim3(256), 0, stream, m->rn_newid, nv, m->verts, m->colors, m->rem,
                     m->verts2, m->colors2, m->rem2);
  int referenced = 0;
  LT_HIP(hipMemcpyAsync(&referenced, m->rn_bsum + nb, sizeof(int), hipMemcpyDeviceToHost, stream));
  LT_HIP(hipStreamSynchronize(stream));
  LT_HIP(hipGetLastError());
  if (referenced != nv) {  // (cannot happen for a mesh of lt_marching_cubes_dev: every vertex lies on an edge some tiling uses)
    // nothing of the mesh has been touched yet: the faces are rewritten only once the new numbering is known to be complete
    lt_set_error("lt_mesh_renumber_dev: %d of %d vertices are referenced by no face", nv - referenced, nv);
    return LT_ERR_BAD_INDEX;
  }
  hipLaunchKernelGGL(k_rn_faces, dim3((n3 + 255) / 256), dim3(256), 0, stream, m->faces, n3, m->rn_newid);
  LT_HIP(hipGetLastError());
  float* tv = m->verts; m->verts = m->verts2; m->verts2 = tv;
  int* tc = m->colors; m->colors = m->colors2; m->colors2 = tc;
  float* tr = m->rem; m->rem = m->rem2; m->rem2 = tr;
  const int tcap = m->cap_v; m->cap_v = m->cap_v2; m->cap_v2 = tcap;
  return LT_OK;
}

// Marching cubes over a TSDF volume's current state (level 0): replaces TSDFVolume.get_mesh (fusion_lidar.py:403-424).
extern "C" int lt_tsdf_extract_mesh_dev(lt_tsdf* t, lt_mesh* m, void* stream, float* ms) {
  if (!t || !m) {
    lt_set_error("lt_tsdf_extract_mesh_dev: NULL volume / mesh");
    return LT_ERR_INVALID_ARG;
  }
  if (t->device != m->device) {
    lt_set_error("lt_tsdf_extract_mesh_dev: volume on device %d, mesh on device %d", t->device, m->device);
    return LT_ERR_INVALID_ARG;
  }
  // the volume knows which columns were written since its last reset: the others are not read
  return mc_extract(t->tsdf, t->color, t->rem, 4 /* LT_VOX records */, t->dim[0], t->dim[1], t->dim[2], t->voxel_size, t->origin, m, stream, ms,
                    t->all_dirty ? nullptr : t->col_epoch, t->epoch, t->all_dirty ? nullptr : t->bits,
                    t->all_dirty ? nullptr : t->chunk_epoch);
}

extern "C" int lt_marching_cubes_dev(const float* tsdf, const float* color_vol, const float* rem_vol, int nx, int ny,
                                     int nz, float voxel_size, const float* origin, lt_mesh* m, void* stream_,
                                     float* ms) {
  return mc_extract(tsdf, color_vol, rem_vol, 1, nx, ny, nz, voxel_size, origin, m, stream_, ms, nullptr, 0u, nullptr, nullptr);
}

extern "C" int lt_scene_set_mesh(lt_scene* s, lt_mesh* m) {
  if (!s || !m) {
    lt_set_error("lt_scene_set_mesh: NULL scene / mesh");
    return LT_ERR_INVALID_ARG;
  }
  if (s->device != m->device) {
    lt_set_error("lt_scene_set_mesh: scene on device %d, mesh on device %d", s->device, m->device);
    return LT_ERR_INVALID_ARG;
  }
  return lt_scene_set_mesh_dev(s, m->verts, m->faces, m->colors, m->rem, m->n_verts, m->n_faces);
}

// One output scan of the `mesh` adaption (laserscan.py:874-914) in one call: see include/lidarhip.h.
extern "C" int lt_fusion_scan_dev(lt_tsdf* vol, lt_mesh* mesh, lt_scene* scene, lt_rayset* rayset, int n_obs,
                                  const float* const* color_ims, const float* const* depth_ims,
                                  const float* const* rem_ims, int im_h, int im_w, float obs_weight, unsigned tsdf_flags,
                                  const float* origin, float* endpoints, int* endcolors, float* range, float* endrem,
                                  int* tri, unsigned trace_flags, void* stream, int sync) {
  if (!vol || !mesh || !scene || !rayset || !origin || n_obs < 0 || (n_obs > 0 && (!color_ims || !depth_ims || !rem_ims))) {
    lt_set_error("lt_fusion_scan_dev: invalid argument");
    return LT_ERR_INVALID_ARG;
  }
  LT_CHECK(lt_tsdf_reset(vol, stream));
  // (all observations of the fresh volume in one pass where the class-aware update allows it: lt_tsdf.hip)
  LT_CHECK(lt_tsdf_integrate_multi_dev(vol, n_obs, color_ims, depth_ims, rem_ims, im_h, im_w, obs_weight, tsdf_flags, stream));
  LT_CHECK(lt_tsdf_extract_mesh_dev(vol, mesh, stream, nullptr));
  LT_CHECK(lt_scene_set_mesh(scene, mesh));
  LT_CHECK(lt_scene_render_dev(scene, rayset, origin, endpoints, endcolors, range, endrem, tri, trace_flags, stream, nullptr));
  if (sync) LT_HIP(hipStreamSynchronize((hipStream_t)stream));
  return LT_OK;
}
