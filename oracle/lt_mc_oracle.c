/*
 * lt_mc_oracle.c -- TEST INFRASTRUCTURE ONLY: scalar CPU restatement of the mesh extraction step
 * `TSDFVolume.get_mesh` of the reference (/root/reference auxiliary/fusion_lidar.py:403-424):
 *
 *     verts, faces, norms, vals = measure.marching_cubes_lewiner(tsdf_vol, level=0)       (:407)
 *     verts_ind = np.round(verts).astype(int)                                             (:409)
 *     verts = verts * self._voxel_size + self._vol_origin                                 (:412)
 *     rgb_vals = color_vol[verts_ind...]; rem = rem_vol[verts_ind...]                     (:415-418)
 *     colors_b/g/r = floor(...) ; colors = floor([r, g, b]).T.astype(np.uint8)            (:419-423)
 *
 * PARITY: pinned to scikit-image 0.18.3 (the last series with `marching_cubes_lewiner` under that name; the reference's
 * pip dependency, README.md:18).  An interpreter that has it (/opt/conda/bin/python3.9 in the build image) runs the
 * reference's own `get_mesh` for the golden fixtures F10 (tests/golden/make_golden_mc.py) and the live fuzz
 * tools/mc_lewiner_fuzz.py; this file reproduces its OUTPUT ARRAYS -- vertices and faces, values and order -- exactly.
 * What is restated is the published algorithm (Lewiner, Lopes, Vieira, Tavares, JGT 8(2) 2003: MarchingCubes.cpp
 * `process_cube`, `test_face`, `test_interior`, `add_c_vertex`) as scikit-image's Cython port runs it
 * (skimage/measure/_marching_cubes_lewiner_cy.pyx: `the_big_switch`, `test_face`, `test_internal`, `Cell`):
 *   - volume[a0][a1][a2]: scikit-image's internal x is the LAST array axis; corner p of a cell = Lewiner's numbering
 *     0:(0,0,0) 1:(1,0,0) 2:(1,1,0) 3:(0,1,0) 4:(0,0,1) .. 7:(0,1,1) in (x, y, z) = (a2, a1, a0); index bit p = value > level;
 *   - CASES[index] -> (case, config); faces tests (asymptotic decider A C - B D with eps = np.spacing(1.0), in double) and
 *     the interior test select the tiling; tilings 6.1.2, 7.3, 10.2, 12.2, 13.3, 13.4 use a centre vertex (edge code 12);
 *   - one vertex per lattice edge, shared by the cells around it, created at its first use in the face stream (cells in
 *     a0, a1, a2 order, a tiling's triangles in table order), placed by the centre-of-mass rule
 *         w_i = 1 / (eps + |v_i|)  (double),   p = (p_1 w_1 + p_2 w_2) / (w_1 + w_2),   stored as float32;
 *     the centre vertex likewise over the eight corners; output in array-axis order (np.fliplr of x, y, z);
 *   - `gradient_direction="descent"` (the default the reference takes): every face's vertex order reversed.
 * The tables are Lewiner's, decoded from scikit-image's LUT file by tools/gen_mc_lewiner.py
 * (lidar_transfer_amd/csrc/lt_mc_lewiner_table.h).
 * Lines :409-:423 (index rounding, world transform, colour unfolding, uint8 wrap) are restated operation by
 * operation in float32 as numpy evaluates them.
 *
 * Independent of the HIP implementation in everything but the tables: plain loops over cells in scikit-image's order, a
 * dense edge -> vertex map, the case switch written out; the kernels use sign bitmasks, word-parallel classification,
 * prefix sums and a flattened tiling directory.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "lt_mc_lewiner_table.h"

/* scikit-image's `FLT_EPSILON` is np.spacing(1.0) = 2^-52 (a double), not C's FLT_EPSILON */
#define LT_MC_EPS 2.220446049250313e-16

/* test_face (MarchingCubes.cpp / _marching_cubes_lewiner_cy.pyx): is the face's ambiguity resolved towards joining
 * the positive corners?  v[] = the cell's eight values minus the level */
static int lw_test_face(const double* v, int face) {
  double A, B, C, D;
  switch (face < 0 ? -face : face) {
    case 1: A = v[0]; B = v[4]; C = v[5]; D = v[1]; break;
    case 2: A = v[1]; B = v[5]; C = v[6]; D = v[2]; break;
    case 3: A = v[2]; B = v[6]; C = v[7]; D = v[3]; break;
    case 4: A = v[3]; B = v[7]; C = v[4]; D = v[0]; break;
    case 5: A = v[0]; B = v[3]; C = v[2]; D = v[1]; break;
    default: A = v[4]; B = v[7]; C = v[6]; D = v[5]; break; /* 6 */
  }
  const double acbd = A * C - B * D;
  if (acbd > -LT_MC_EPS && acbd < LT_MC_EPS) return face >= 0;
  return (double)face * A * acbd >= 0;
}

/* test_interior / test_internal: does the interior of the cell connect the two face-ambiguous sheets? */
static int lw_test_interior(const double* v, int mc_case, int config, int subconfig, int s) {
  double t, At = 0, Bt = 0, Ct = 0, Dt = 0, a, b;
  int edge = -1;
  if (mc_case == 4 || mc_case == 10) {
    a = (v[4] - v[0]) * (v[6] - v[2]) - (v[7] - v[3]) * (v[5] - v[1]);
    b = v[2] * (v[4] - v[0]) + v[0] * (v[6] - v[2]) - v[1] * (v[7] - v[3]) - v[3] * (v[5] - v[1]);
    t = -b / (2 * a + LT_MC_EPS); /* (scikit-image guards its divisions with + eps; it decides ties) */
    if (t < 0 || t > 1) return s > 0;
    At = v[0] + (v[4] - v[0]) * t;
    Bt = v[3] + (v[7] - v[3]) * t;
    Ct = v[2] + (v[6] - v[2]) * t;
    Dt = v[1] + (v[5] - v[1]) * t;
  } else {
    switch (mc_case) {
      case 6: edge = LT_LW_TEST6[config][2]; break;
      case 7: edge = LT_LW_TEST7[config][4]; break;
      case 12: edge = LT_LW_TEST12[config][3]; break;
      default: edge = LT_LW_TILING13_5_1[config][subconfig][0]; break; /* 13 */
    }
    /* the reference edge (p, q) and the three edges parallel to it, walking round the cell: B, C, D */
    static const signed char E[12][8] = {
        {0, 1, 3, 2, 7, 6, 4, 5}, {1, 2, 0, 3, 4, 7, 5, 6}, {2, 3, 1, 0, 5, 4, 6, 7}, {3, 0, 2, 1, 6, 5, 7, 4},
        {4, 5, 7, 6, 3, 2, 0, 1}, {5, 6, 4, 7, 0, 3, 1, 2}, {6, 7, 5, 4, 1, 0, 2, 3}, {7, 4, 6, 5, 2, 1, 3, 0},
        {0, 4, 3, 7, 2, 6, 1, 5}, {1, 5, 0, 4, 3, 7, 2, 6}, {2, 6, 1, 5, 0, 4, 3, 7}, {3, 7, 2, 6, 1, 5, 0, 4}};
    const signed char* e = E[edge];
    t = v[e[0]] / (v[e[0]] - v[e[1]] + LT_MC_EPS);
    At = 0;
    Bt = v[e[2]] + (v[e[3]] - v[e[2]]) * t;
    Ct = v[e[4]] + (v[e[5]] - v[e[4]]) * t;
    Dt = v[e[6]] + (v[e[7]] - v[e[6]]) * t;
  }
  int test = 0;
  if (At >= 0) test += 1;
  if (Bt >= 0) test += 2;
  if (Ct >= 0) test += 4;
  if (Dt >= 0) test += 8;
  switch (test) {
    case 0: case 1: case 2: case 3: case 4: case 6: case 8: case 9: case 12: return s > 0;
    case 5: if (At * Ct - Bt * Dt < LT_MC_EPS) return s > 0; break;
    case 10: if (At * Ct - Bt * Dt >= LT_MC_EPS) return s > 0; break;
    default: return s < 0; /* 7, 11, 13, 14, 15 */
  }
  /* Lewiner's C++ ends in `return s < 0` here; scikit-image's port falls off the end of `test_internal` and returns 0
   * (measured, tools/mc_lewiner_fuzz.py: case 4 with a negative TEST4 entry takes tiling 4.1.2 for tests 5 and 10
   * whatever the sign of At Ct - Bt Dt) -- the reference runs scikit-image, so that is the behaviour restated */
  return 0;
}

/* process_cube / the_big_switch: the tiling (edge codes, 3 per triangle) and its triangle count for a cell */
static const signed char* lw_tiling(const double* v, int index, int* nt) {
  const int c = LT_LW_CASES[index][0], g = LT_LW_CASES[index][1];
  int sub = 0;
#define LW(T, n) do { *nt = (n); return (const signed char*)(T); } while (0)
  switch (c) {
    case 1: LW(LT_LW_TILING1[g], 1);
    case 2: LW(LT_LW_TILING2[g], 2);
    case 3:
      if (lw_test_face(v, LT_LW_TEST3[g])) LW(LT_LW_TILING3_2[g], 4);
      LW(LT_LW_TILING3_1[g], 2);
    case 4:
      if (lw_test_interior(v, c, g, 0, LT_LW_TEST4[g])) LW(LT_LW_TILING4_1[g], 2);
      LW(LT_LW_TILING4_2[g], 6);
    case 5: LW(LT_LW_TILING5[g], 3);
    case 6:
      if (lw_test_face(v, LT_LW_TEST6[g][0])) LW(LT_LW_TILING6_2[g], 5);
      if (lw_test_interior(v, c, g, 0, LT_LW_TEST6[g][1])) LW(LT_LW_TILING6_1_1[g], 3);
      LW(LT_LW_TILING6_1_2[g], 9);
    case 7:
      if (lw_test_face(v, LT_LW_TEST7[g][0])) sub += 1;
      if (lw_test_face(v, LT_LW_TEST7[g][1])) sub += 2;
      if (lw_test_face(v, LT_LW_TEST7[g][2])) sub += 4;
      switch (sub) {
        case 0: LW(LT_LW_TILING7_1[g], 3);
        case 1: LW(LT_LW_TILING7_2[g][0], 5);
        case 2: LW(LT_LW_TILING7_2[g][1], 5);
        case 3: LW(LT_LW_TILING7_3[g][0], 9);
        case 4: LW(LT_LW_TILING7_2[g][2], 5);
        case 5: LW(LT_LW_TILING7_3[g][1], 9);
        case 6: LW(LT_LW_TILING7_3[g][2], 9);
        default:
          if (lw_test_interior(v, c, g, 0, LT_LW_TEST7[g][3])) LW(LT_LW_TILING7_4_2[g], 9);
          LW(LT_LW_TILING7_4_1[g], 5);
      }
    case 8: LW(LT_LW_TILING8[g], 2);
    case 9: LW(LT_LW_TILING9[g], 4);
    case 10:
      if (lw_test_face(v, LT_LW_TEST10[g][0])) {
        if (lw_test_face(v, LT_LW_TEST10[g][1])) LW(LT_LW_TILING10_1_1_[g], 4);
        LW(LT_LW_TILING10_2[g], 8);
      }
      if (lw_test_face(v, LT_LW_TEST10[g][1])) LW(LT_LW_TILING10_2_[g], 8);
      if (lw_test_interior(v, c, g, 0, LT_LW_TEST10[g][2])) LW(LT_LW_TILING10_1_1[g], 4);
      LW(LT_LW_TILING10_1_2[g], 8);
    case 11: LW(LT_LW_TILING11[g], 4);
    case 12:
      if (lw_test_face(v, LT_LW_TEST12[g][0])) {
        if (lw_test_face(v, LT_LW_TEST12[g][1])) LW(LT_LW_TILING12_1_1_[g], 4);
        LW(LT_LW_TILING12_2[g], 8);
      }
      if (lw_test_face(v, LT_LW_TEST12[g][1])) LW(LT_LW_TILING12_2_[g], 8);
      if (lw_test_interior(v, c, g, 0, LT_LW_TEST12[g][2])) LW(LT_LW_TILING12_1_1[g], 4);
      LW(LT_LW_TILING12_1_2[g], 8);
    case 13: {
      for (int k = 0; k < 6; ++k)
        if (lw_test_face(v, LT_LW_TEST13[g][k])) sub += 1 << k;
      const int sc = LT_LW_SUBCONFIG13[sub];
      if (sc < 0) { *nt = 0; return 0; } /* "Impossible case 13?": nothing is added */
      if (sc == 0) LW(LT_LW_TILING13_1[g], 4);
      if (sc <= 6) LW(LT_LW_TILING13_2[g][sc - 1], 6);
      if (sc <= 18) LW(LT_LW_TILING13_3[g][sc - 7], 10);
      if (sc <= 22) LW(LT_LW_TILING13_4[g][sc - 19], 12);
      if (sc <= 26) {
        if (lw_test_interior(v, c, g, sc - 23, LT_LW_TEST13[g][6])) LW(LT_LW_TILING13_5_1[g][sc - 23], 6);
        LW(LT_LW_TILING13_5_2[g][sc - 23], 10);
      }
      if (sc <= 38) LW(LT_LW_TILING13_3_[g][sc - 27], 10);
      if (sc <= 44) LW(LT_LW_TILING13_2_[g][sc - 39], 6);
      if (sc == 45) LW(LT_LW_TILING13_1_[g], 4);
      *nt = 0;
      return 0; /* "impossible case 13" */
    }
    case 14: LW(LT_LW_TILING14[g], 4);
    default: *nt = 0; return 0;
  }
#undef LW
}

/* corner p of a cell -> offsets along (a0, a1, a2); edge code -> (corner 1, corner 2) */
static const signed char LW_CORNER[8][3] = {{0, 0, 0}, {0, 0, 1}, {0, 1, 1}, {0, 1, 0}, {1, 0, 0}, {1, 0, 1}, {1, 1, 1}, {1, 1, 0}};
static const signed char LW_EDGE[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6}, {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};

/*
 * Returns 0 and the counts in *n_verts / *n_faces.  Output arrays may be NULL (count only); otherwise they must
 * hold cap_v vertices / cap_f faces -- if the mesh is larger, nothing is written beyond the capacity and the
 * return value is 1.
 */
int lto_marching_cubes(const float* tsdf, const float* color_vol, const float* rem_vol, int nx, int ny, int nz,
                       float voxel_size, const float* origin, float* verts, int* faces, int* colors, float* rem,
                       int cap_v, int cap_f, int* n_verts, int* n_faces) {
  const size_t n = (size_t)nx * ny * nz;
  const size_t sx = (size_t)ny * nz, sy = (size_t)nz;
  const int dim[3] = {nx, ny, nz};
  int* vid = (int*)malloc(4 * n * sizeof(int)); /* vertex id of (voxel, slot): slots 0..2 = the edge along a2 / a1 / a0, 3 = centre */
  if (!vid) return -1;
  for (size_t k = 0; k < 4 * n; ++k) vid[k] = -1;
  int nv = 0, nf = 0, overflow = 0;
  for (int x = 0; x + 1 < nx; ++x)
    for (int y = 0; y + 1 < ny; ++y)
      for (int z = 0; z + 1 < nz; ++z) {
        const size_t i = x * sx + y * sy + z;
        double v[8];
        int index = 0;
        for (int p = 0; p < 8; ++p) {
          v[p] = (double)tsdf[i + LW_CORNER[p][0] * sx + LW_CORNER[p][1] * sy + LW_CORNER[p][2]]; /* level = 0 */
          if (v[p] > 0.0) index |= 1 << p;
        }
        if (LT_LW_CASES[index][0] == 0) continue;
        int nt = 0;
        const signed char* T = lw_tiling(v, index, &nt);
        for (int t = 0; t < nt; ++t) {
          int f3[3];
          for (int k = 0; k < 3; ++k) {
            const int code = T[3 * t + k];
            size_t slot;
            double p[3] = {(double)x, (double)y, (double)z}; /* (a0, a1, a2) */
            if (code == 12) {
              slot = 4 * i + 3;
              if (vid[slot] < 0) {
                double f[3] = {0, 0, 0}, ff = 0;
                for (int q = 0; q < 8; ++q) {
                  const double w = 1.0 / (LT_MC_EPS + fabs(v[q]));
                  for (int a = 0; a < 3; ++a) f[a] += (double)LW_CORNER[q][a] * w;
                  ff += w;
                }
                for (int a = 0; a < 3; ++a) p[a] += f[a] / ff;
              }
            } else {
              const int c1 = LW_EDGE[code][0], c2 = LW_EDGE[code][1];
              /* the lattice edge's owner voxel = the lower corner; its axis */
              int lo[3], axis = 0;
              for (int a = 0; a < 3; ++a) {
                lo[a] = LW_CORNER[c1][a] < LW_CORNER[c2][a] ? LW_CORNER[c1][a] : LW_CORNER[c2][a];
                if (LW_CORNER[c1][a] != LW_CORNER[c2][a]) axis = a;
              }
              const size_t j = i + lo[0] * sx + lo[1] * sy + lo[2];
              slot = 4 * j + (2 - axis);
              if (vid[slot] < 0) {
                const double w1 = 1.0 / (LT_MC_EPS + fabs(v[c1])), w2 = 1.0 / (LT_MC_EPS + fabs(v[c2]));
                double f[3] = {0, 0, 0}, ff = 0;
                for (int a = 0; a < 3; ++a) f[a] += (double)LW_CORNER[c1][a] * w1;
                ff += w1;
                for (int a = 0; a < 3; ++a) f[a] += (double)LW_CORNER[c2][a] * w2;
                ff += w2;
                for (int a = 0; a < 3; ++a) p[a] += f[a] / ff;
              }
            }
            if (vid[slot] < 0) {
              vid[slot] = nv;
              if (verts && nv < cap_v) {
                int ind[3];
                for (int a = 0; a < 3; ++a) {
                  const float pf = (float)p[a];                   /* the float32 vertex array of scikit-image */
                  ind[a] = (int)rintf(pf);                        /* np.round: half to even, on the float32 value */
                  if (!(ind[a] >= 0)) ind[a] = 0;                 /* (a NaN field value: numpy would raise) */
                  if (ind[a] > dim[a] - 1) ind[a] = dim[a] - 1;
                  verts[3 * (size_t)nv + a] = pf * voxel_size + origin[a]; /* float32 multiply, then float32 add */
                }
                const size_t j = ind[0] * sx + ind[1] * sy + ind[2];
                const float rgb = color_vol[j];
                const float cb = floorf(rgb / (float)(256 * 256));
                const float cg = floorf((rgb - cb * 256.0f * 256.0f) / 256.0f);
                const float cr = rgb - cb * 256.0f * 256.0f - cg * 256.0f;
                /* .astype(np.uint8): truncation to 8 bits (labels 256..259 wrap to 0..3) */
                colors[3 * (size_t)nv] = (int)(uint8_t)(int)floorf(cr);
                colors[3 * (size_t)nv + 1] = (int)(uint8_t)(int)floorf(cg);
                colors[3 * (size_t)nv + 2] = (int)(uint8_t)(int)floorf(cb);
                rem[nv] = rem_vol[j];
              } else if (verts) {
                overflow = 1;
              }
              ++nv;
            }
            f3[k] = vid[slot];
          }
          if (faces && nf < cap_f) { /* gradient_direction == "descent": np.fliplr(faces) */
            faces[3 * (size_t)nf] = f3[2]; faces[3 * (size_t)nf + 1] = f3[1]; faces[3 * (size_t)nf + 2] = f3[0];
          } else if (faces) {
            overflow = 1;
          }
          ++nf;
        }
      }
  free(vid);
  if (n_verts) *n_verts = nv;
  if (n_faces) *n_faces = nf;
  return overflow;
}

