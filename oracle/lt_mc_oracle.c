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
 * PARITY UNPINNED for the marching-cubes call itself: scikit-image (the reference's pip dependency, no version
 * pinned, README.md:18) is not importable in this image, so neither its Lewiner look-up tables nor an output
 * of it can be obtained.  What is restated from its published algorithm (skimage/measure/
 * _marching_cubes_lewiner_cy.pyx, Cell._add_face_from_edge_index): one vertex per sign-changing lattice edge,
 * shared by all cells around the edge, placed by the centre-of-mass rule
 *     w_i = 1 / (eps + |v_i - level|)   (double, eps = np.spacing(1.0)),   p = (p_1 w_1 + p_2 w_2) / (w_1 + w_2),
 * stored as float32.  The triangulation comes from the generated table oracle/lt_mc_table.h (classic marching
 * cubes with a face-consistent, watertight disambiguation -- tools/gen_mc_table.py); on ambiguous cells
 * scikit-image's Lewiner variant may connect differently, and the ORDER of vertices / faces is this
 * implementation's (owner voxel x, y, z ascending, then edge axis; cells ascending, then table order).
 * Lines :409-:423 (index rounding, world transform, colour unfolding, uint8 wrap) are restated operation by
 * operation in float32 as numpy evaluates them.
 *
 * Independent of the HIP implementation in everything but the table: plain loops over voxels and cells, a dense
 * edge -> vertex map; the kernels use sign bitmasks, word-parallel classification and prefix sums.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "lt_mc_table.h"

static inline int mc_inside(float v) { return v < 0.0f; } /* level = 0; NaN is outside */

/* float32 vertex coordinate along the edge from voxel coordinate c (value v1) to c + 1 (value v2) */
/* scikit-image's `FLT_EPSILON` is np.spacing(1.0) = 2^-52 (a double), not C's FLT_EPSILON */
#define LT_MC_EPS 2.220446049250313e-16
static inline float mc_edge_coord(int c, float v1, float v2) {
  const double w1 = 1.0 / (LT_MC_EPS + fabs((double)v1));
  const double w2 = 1.0 / (LT_MC_EPS + fabs((double)v2));
  const double ff = w1 + w2;
  return (float)((double)c + w2 / ff);
}

/*
 * Returns 0 and the counts in *n_verts / *n_faces.  Output arrays may be NULL (count only); otherwise they must
 * hold cap_v vertices / cap_f faces -- if the mesh is larger, nothing is written beyond the capacity and the
 * return value is 1.
 */
int lto_marching_cubes(const float* tsdf, const float* color_vol, const float* rem_vol, int nx, int ny, int nz,
                       float voxel_size, const float* origin, float* verts, int* faces, int* colors, float* rem,
                       int cap_v, int cap_f, int* n_verts, int* n_faces) {
  const size_t n = (size_t)nx * ny * nz;
  const size_t sx = (size_t)ny * nz, sy = (size_t)nz, sz = 1;
  const size_t stride[3] = {sx, sy, sz};
  const int dim[3] = {nx, ny, nz};
  int* vid = (int*)malloc(3 * n * sizeof(int)); /* vertex id of edge (voxel, axis) or -1 */
  if (!vid) return -1;
  int nv = 0, nf = 0, overflow = 0;
  for (int x = 0; x < nx; ++x)
    for (int y = 0; y < ny; ++y)
      for (int z = 0; z < nz; ++z) {
        const size_t i = x * sx + y * sy + z;
        const int c[3] = {x, y, z};
        for (int a = 0; a < 3; ++a) {
          vid[3 * i + a] = -1;
          if (c[a] + 1 >= dim[a]) continue;
          const float v1 = tsdf[i], v2 = tsdf[i + stride[a]];
          if (mc_inside(v1) == mc_inside(v2)) continue;
          vid[3 * i + a] = nv;
          if (verts && nv < cap_v) {
            float p[3] = {(float)x, (float)y, (float)z};
            p[a] = mc_edge_coord(c[a], v1, v2);
            int ind[3];
            for (int k = 0; k < 3; ++k) {
              ind[k] = (int)rintf(p[k]);                        /* np.round: half to even, on the float32 value */
              if (!(ind[k] >= 0)) ind[k] = 0;                   /* (a NaN field value: numpy would raise) */
              if (ind[k] > dim[k] - 1) ind[k] = dim[k] - 1;
              verts[3 * (size_t)nv + k] = p[k] * voxel_size + origin[k]; /* float32 multiply, then float32 add */
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
      }
  for (int x = 0; x + 1 < nx; ++x)
    for (int y = 0; y + 1 < ny; ++y)
      for (int z = 0; z + 1 < nz; ++z) {
        const size_t i = x * sx + y * sy + z;
        int cs = 0;
        for (int k = 0; k < 8; ++k)
          if (mc_inside(tsdf[i + (k & 1) * sx + ((k >> 1) & 1) * sy + ((k >> 2) & 1) * sz])) cs |= 1 << k;
        for (int t = LT_MC_FIRST[cs]; t < LT_MC_FIRST[cs + 1]; t += 3) {
          if (faces && nf < cap_f) {
            for (int k = 0; k < 3; ++k) {
              const int code = LT_MC_TRIS[t + k], c0 = code & 7, a = code >> 3;
              const size_t j = i + (c0 & 1) * sx + ((c0 >> 1) & 1) * sy + ((c0 >> 2) & 1) * sz;
              faces[3 * (size_t)nf + k] = vid[3 * j + a];
            }
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
