// lt_tsdf.hip -- class-aware TSDF integration of a spherical range image.
//
// Replaces the pycuda kernel `integrate` of the reference's TSDFVolume (auxiliary/fusion_lidar.py:66-229,
// launch :252-287, volumes :46-63) with device-resident volumes: one thread per voxel projects the voxel
// centre into the (H x W) spherical image, reads the observed depth there and updates tsdf / weight /
// colour(label) / remission.  Both branches of the kernel are kept:
//   LT_TSDF_MERGE (what the reference runs: `bool merge = true`, :177)  class-aware: same label -> running
//        average of tsdf and remission; different label -> the closer observation replaces the voxel
//        (comparing the new distance with the WEIGHT volume, sic, :195, :212)
//   0            (`merge == false`, :178-205) plain running average incl. per-channel colour average
// The arithmetic follows the CUDA source expression by expression (float unless the source promotes to
// double through the PI / 1.0 literals); `a + b * c` patterns are written as fused multiply-adds because
// nvcc contracts them by default (-fmad=true).  norm3df / atan2f / asinf are the device library's -- exactly what
// the reference's own kernel source gets when hipcc compiles it for gfx950 (oracle/build_ref_tsdf.py ->
// oracle/_ref/libref_tsdf_integrate.so): tests/test_tsdf_ref_kernel_gpu.py runs that build next to these kernels and
// finds all four volumes bit-identical.  (CUDA's own last-ulp behaviour of the three functions exists on no other
// platform; against a CUDA run voxels within an ulp of a pixel / field-of-view / truncation boundary may land differently.)
//
// The kernel is the reference's arithmetic, voxel by voxel; what is NOT the reference's is the amount of work spent
// on voxels that cannot change -- 800 M voxels at the default volume, of which one observation updates a few per cent:
//   * the image column px of a voxel depends on its (x, y) only: k_tsdf_columns computes it once per COLUMN of
//     dim_z voxels with the very expressions of the kernel (atan2f and the double-precision proj_x: the costly part),
//     and marks a column DEAD when even its nearest possible depth sqrt(x^2 + y^2) lies more than the truncation
//     margin behind the largest depth of that image column -- every voxel of it takes the kernel's
//     `depth_diff < -trunc_margin` exit (float subtraction is monotonic and the test leaves four ulp of room for the
//     rounding of norm3df, so the implication holds);
//   * a conservative sine test (|margin| 1e-5, far above asinf's error) drops voxels clearly outside the vertical
//     field of view before asinf; the band around the limits takes the exact path;
//   * columns written since the last reset are stamped with the volume's epoch: reset re-initialises those only, and
//     marching cubes (lt_mc.hip) does not read the clean ones.
// The one-thread-per-voxel restatement of the reference kernel is test infrastructure (oracle/lt_tsdf_dense.hip,
// not in this library): the A/B partner of the kernels below -- bit-identical volumes, tests/test_tsdf_gpu.py.
#include "lt_internal.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define LT_PI_D 3.14159265358979323846
#ifndef LT_TSDF_WAVES_ATTR
#define LT_TSDF_WAVES_ATTR
#endif
#define LT_TSDF_DBG_WAVES (1 << 18)  // debug stamps: 65 536 workgroups (the default volume has 62 500 chunks)
// col_zw: the z range written in a column since the last reset, as TWO words that only ever grow -- [c] = 0x7fff - lo,
// [n_cols + c] = hi + 1; a clean column holds (0, 0) -- so that any number of writers can merge into it with two
// non-returning atomicMax (k_tsdf_integrate_pix: a wall's column is visited by dozens of image rows at once; a
// compare-and-swap loop on one packed word serialised them, 95 us per workgroup)
#define LT_ZW_LO(a) (0x7FFF - (int)(a))
#define LT_ZW_HI(b) ((int)(b) - 1)

// The volume is ONE array of float4 per voxel -- (tsdf, weight, colour, remission), [x][y][z] -- not the reference's four
// arrays: an update is one 16-byte read-modify-write instead of four 4-byte ones in four places (a z run of ~10 written
// voxels was 2-3 32-byte sectors in each of four arrays: 3.3 x the bytes of the voxels), and marching cubes finds a
// vertex's field samples and its attributes in the two lines it reads anyway.  The kernels keep their four pointer
// arguments (base, base + 1, base + 2, base + 3: what lt_tsdf_volumes hands out, stride 4); only LT_VOX touches memory.
#define LT_VOX(tsdf_base, idx) (reinterpret_cast<float4*>(tsdf_base) + (idx))
__global__ __launch_bounds__(256) void k_tsdf_fill(float4* __restrict__ vol, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    vol[i] = make_float4(1.0f, 0.0f, 0.0f, 0.0f);  // np.ones / np.zeros (fusion_lidar.py:47-51)
}

// the update of one voxel (fusion_lidar.py:178-228); returns what
// happened to the voxel's tsdf: 0 untouched, 1 written (above the level 0), 2 written NOT above it -- <= 0 or NaN -- (the sign bit marching cubes
// needs, lt_mc.hip)
// `fresh`: the voxel's column has not been written since the last reset, so the old values are the initial ones
// (tsdf 1, weight 0, colour 0, remission 0) and need not be loaded
template <bool MERGE>
__device__ __forceinline__ int tsdf_update(float* __restrict__ tsdf_vol, float* __restrict__ /*weight_vol*/,
                                            float* __restrict__ /*color_vol*/, float* __restrict__ /*rem_vol*/, int voxel_idx,
                                            float dist, float obs_weight, float new_color, float new_rem,
                                            bool fresh = false) {
  float4* const vox = LT_VOX(tsdf_vol, voxel_idx);
  const float4 o = fresh ? make_float4(1.0f, 0.0f, 0.0f, 0.0f) : *vox;  // (tsdf, weight, colour, remission)
  if (!MERGE) {
    const float w_old = o.y;
    const float w_new = w_old + obs_weight;
    const float tv = __fmaf_rn(o.x, w_old, dist) / w_new;
    const float old_color = o.z;
    const float old_b = floorf(old_color / (256 * 256));
    const float old_g = floorf((old_color - old_b * 256 * 256) / 256);
    const float old_r = old_color - old_b * 256 * 256 - old_g * 256;
    float new_b = floorf(new_color / (256 * 256));
    float new_g = floorf((new_color - new_b * 256 * 256) / 256);
    float new_r = new_color - new_b * 256 * 256 - new_g * 256;
    new_b = fminf(roundf(__fmaf_rn(old_b, w_old, new_b) / w_new), 255.0f);
    new_g = fminf(roundf(__fmaf_rn(old_g, w_old, new_g) / w_new), 255.0f);
    new_r = fminf(roundf(__fmaf_rn(old_r, w_old, new_r) / w_new), 255.0f);
    *vox = make_float4(tv, w_new, new_b * 256 * 256 + new_g * 256 + new_r, __fmaf_rn(o.w, w_old, new_rem) / w_new);
    return !(tv > 0.0f) ? 2 : 1;
  } else {
    const float dist_old = o.y;  // sic: the reference compares against the weight volume
    const float old_color = o.z;
    if (old_color == new_color) {  // same class: integrate (the colour stays)
      const float w_old = o.y;
      const float w_new = w_old + obs_weight;
      const float tv = __fmaf_rn(o.x, w_old, dist) / w_new;
      *vox = make_float4(tv, w_new, o.z, __fmaf_rn(o.w, w_old, new_rem) / w_new);
      return !(tv > 0.0f) ? 2 : 1;
    } else if (dist < dist_old) {  // other class: the closer observation wins (the weight stays)
      const float new_b = floorf(new_color / (256 * 256));
      const float new_g = floorf((new_color - new_b * 256 * 256) / 256);
      const float new_r = new_color - new_b * 256 * 256 - new_g * 256;
      *vox = make_float4(dist, o.y, new_b * 256 * 256 + new_g * 256 + new_r, new_rem);
      return !(dist > 0.0f) ? 2 : 1;
    }
    return 0;
  }
}

// largest depth of every image column: a workgroup takes 64 columns, its four waves a quarter of the rows each (rows of
// the image are contiguous: coalesced), partial maxima through LDS.  (One thread per column walking its rows: 8
// workgroups, 64 dependent-latency loads each, 18 us.)
// ... and the transposed, packed copy of the depth and colour images (lt_tsdf::dct).
__global__ __launch_bounds__(256) void k_tsdf_colmax(const float* __restrict__ depth_im,
                                                     const float* __restrict__ color_im, int im_h, int im_w,
                                                     float* __restrict__ colmax, float2* __restrict__ dct) {
  __shared__ float part[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int x = blockIdx.x * 64 + tx;
  // the maximum over the pixels the reference kernel does not leave at `depth_value == 0` -- NOT started at 0: the
  // reference's "no data" value is -1 (laserscan.py:38), and with a truncation margin above 1 m (voxel_size > 0.2) the
  // voxels within trunc_margin - 1 of the sensor ARE written through such a pixel (depth_diff = -1 - depth >= -trunc)
  float m = -INFINITY;  // no non-zero pixel: every voxel of the column leaves at `depth_value == 0`
  if (x < im_w)
    for (int y = ty; y < im_h; y += 4) {
      const float d = depth_im[y * im_w + x];
      // (a NaN pixel passes every depth test of the reference kernel: never "dead")
      m = d == d ? (d != 0.f ? fmaxf(m, d) : m) : INFINITY;
      dct[(size_t)x * im_h + y] = make_float2(d, color_im[y * im_w + x]);
    }
  part[ty][tx] = m;
  __syncthreads();
  if (ty == 0 && x < im_w) colmax[x] = fmaxf(fmaxf(part[0][tx], part[1][tx]), fmaxf(part[2][tx], part[3][tx]));
}

// the transposed, packed copy alone (the pixel-centric integrate has no use for the column maxima): one thread per pixel,
// reads coalesced along the image rows, 512 workgroups for a 64 x 2048 image (k_tsdf_colmax: 32, 18 us)
__global__ __launch_bounds__(256) void k_tsdf_dct(const float* __restrict__ depth_im, const float* __restrict__ color_im,
                                                  int im_h, int im_w, float2* __restrict__ dct) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= im_h * im_w) return;
  const int y = i / im_w, x = i - y * im_w;
  dct[(size_t)x * im_h + y] = make_float2(depth_im[i], color_im[i]);
}

// A walk over z in [z0, z1) of the table column (cx, cy) is PLAIN when the reference's float decomposition of every voxel
// index in it (:95-98) yields (cx, cy, z): (float)voxel_idx is monotone in voxel_idx and so is the quotient's floor, hence
// it is enough that both ends of the walk decompose to cx -- voxel_y and voxel_z then follow in exact integer / small-float
// arithmetic.  (Not plain: walks touching an x boundary of a volume of more than 2^24 voxels.)  For a plain walk the
// column's share of the per-voxel expressions is computed once: the two IEEE divisions, two fused multiply-adds, the
// product and the table look-up were a quarter of the vector instructions of the column walk
// (SQ_ACTIVE_INST_VALU: 71 % of its time; 439 -> 352 us on the default volume).
// Columns on the sensor's vertical axis: norm3df(x, y, z) < |z| needs x^2 + y^2 below the rounding of z^2 and of the
// 1-ulp square root, i.e. rho < 6e-4 |z|; columns with rho <= LT_AXIS_RHO * (largest |pt_z| of the volume) are taken out
// of every candidate shortcut (whole z range, every voxel through the exact expressions).  A handful of columns at most.
#define LT_AXIS_RHO 1e-3f
__device__ __forceinline__ float axis_rho_limit(float oz, float voxel_size, int dim_z) {
  return LT_AXIS_RHO * fmaxf(fabsf(oz), fabsf(__fmaf_rn((float)(dim_z - 1), voxel_size, oz)));
}

struct col_plain {
  bool plain;
  int px;      // colinfo[cx * dim_y + cy]
  float rho2;  // fma(pt_y, pt_y, pt_x * pt_x): the conservative candidate tests work on it
  int col;     // cx * dim_y + cy: the exact evaluation takes pt_x, pt_y from it (col_xy)
};
// (pt_x, pt_y) of table column `col` -- the kernel's `vol_origin + voxel * voxel_size` on voxel_x = cx, voxel_y = cy (:101-103)
__device__ __forceinline__ void col_xy(int col, int dim_y, float voxel_size, float ox, float oy, float& pt_x, float& pt_y) {
  int cx = (int)((float)col * __builtin_amdgcn_rcpf((float)dim_y));  // (within one of col / dim_y for dim_x < 2^21)
  int cy = col - cx * dim_y;
  if (cy < 0) { cx -= 1; cy += dim_y; }
  if (cy >= dim_y) { cx += 1; cy -= dim_y; }
  pt_x = __fmaf_rn((float)cx, voxel_size, ox);
  pt_y = __fmaf_rn((float)cy, voxel_size, oy);
}
__device__ __forceinline__ col_plain col_plain_of(int cx, int cy, int z0, int z1, int vol_dim_y, int vol_dim_z, float ox,
                                                  float oy, float voxel_size, const int* __restrict__ colinfo) {
  col_plain C;
  C.plain = false; C.px = -2; C.rho2 = 0.f; C.col = 0;
  if (z1 <= z0) return C;
  const int cc = cx * vol_dim_y + cy;
  const int i0 = cc * vol_dim_z + z0, i1 = cc * vol_dim_z + z1 - 1;
  const float dyz = (float)(vol_dim_y * vol_dim_z);
  const float x0 = floorf(((float)i0) / dyz), x1 = floorf(((float)i1) / dyz);
  if (x0 != (float)cx || x1 != (float)cx) return C;
  const int px = colinfo[cc];
  if (px < 0) return C;  // (a dead column walked because cy == dim_y - 1: the general path sorts its voxels out)
  const float pt_x = __fmaf_rn((float)cx, voxel_size, ox), pt_y = __fmaf_rn((float)cy, voxel_size, oy);
  C.plain = true;
  C.px = px;
  C.rho2 = __fmaf_rn(pt_y, pt_y, pt_x * pt_x);
  C.col = cc;
  return C;
}

template <bool MERGE>
__device__ __forceinline__ int tsdf_voxel(
    int voxel_idx, float* __restrict__ tsdf_vol, float* __restrict__ weight_vol, float* __restrict__ color_vol,
    float* __restrict__ rem_vol, int vol_dim_x, int vol_dim_y, int vol_dim_z, float ox, float oy, float oz,
    float voxel_size, int im_h, int im_w, float trunc_margin, float obs_weight, float fov_up, float fov_down,
    float sin_up_hi, float sin_down_lo, const float* __restrict__ color_im, const float* __restrict__ depth_im,
    const float* __restrict__ rem_im, const int* __restrict__ colinfo, unsigned* __restrict__ col_epoch,
    unsigned epoch, bool fresh, const col_plain& C, int z_plain, const float2* __restrict__ dct, int want_py = -1) {
  int px = -2;
  float pt_x, pt_y, pt_z;
  if (C.plain) {
    // the reference's float decomposition of every voxel index of this walk gives (cx, cy, z) (see col_plain): the
    // column's share of the expressions below comes from the caller
    px = C.px;
    col_xy(C.col, vol_dim_y, voxel_size, ox, oy, pt_x, pt_y);
    pt_z = __fmaf_rn((float)z_plain, voxel_size, oz);
  } else {
    // voxel grid coordinates -- float division exactly as the reference (:95-98); beyond 2^24 voxels (float)voxel_idx
    // is rounded, which moves a few voxels next to an x boundary to (x + 1, -1, z): those are not a column of the table
    const float voxel_x = floorf(((float)voxel_idx) / ((float)(vol_dim_y * vol_dim_z)));
    const float voxel_y = floorf(((float)(voxel_idx - ((int)voxel_x) * vol_dim_y * vol_dim_z)) / ((float)vol_dim_z));
    const float voxel_z = (float)(voxel_idx - ((int)voxel_x) * vol_dim_y * vol_dim_z - ((int)voxel_y) * vol_dim_z);
    const int ix = (int)voxel_x, iy = (int)voxel_y;
    const bool in_table = ix >= 0 && ix < vol_dim_x && iy >= 0 && iy < vol_dim_y;
    if (in_table) {
      px = colinfo[ix * vol_dim_y + iy];
      if (px == -1) return 0;
      if (px >= 0) px &= 0x3FFFFFFF;  // (the wedge table's per-column image column carries a flag bit: LT_WD_QUIRK_FLAG)
    }
    pt_x = __fmaf_rn(voxel_x, voxel_size, ox);
    pt_y = __fmaf_rn(voxel_y, voxel_size, oy);
    pt_z = __fmaf_rn(voxel_z, voxel_size, oz);
    if (px < 0) {
      const float yaw = -atan2f(pt_y, pt_x);
      float proj_x = (float)(0.5 * ((double)yaw / LT_PI_D + 1.0));
      proj_x *= (float)im_w;
      px = (int)floorf(proj_x);
      px = min(im_w - 1, px);
      px = max(0, px);
    }
  }
  const float fov = fabsf(fov_up) + fabsf(fov_down);
  const float depth = norm3df(pt_x, pt_y, pt_z);  // the device library's, as the reference's source gets it (header note)
  const float s = pt_z / depth;
  // clearly outside the vertical field of view (NaN passes on -- and so does |s| > 1: on the sensor's axis the device
  // library's norm3df, a 1-ulp square root, can return less than |pt_z|; asinf is then NaN, no comparison of the
  // reference holds and the voxel IS written through pixel row 0 -- LT_AXIS_RHO)
  if ((s > sin_up_hi || s < sin_down_lo) && fabsf(s) <= 1.0f) return 0;
  const float pitch = asinf(s);
  if (pitch > fov_up || pitch < fov_down) return 0;
  float proj_y = (float)(1.0 - (double)((pitch + fabsf(fov_down)) / fov));
  proj_y *= (float)im_h;
  int py = (int)floorf(proj_y);
  py = min(im_h - 1, py);
  py = max(0, py);
  // (the pixel-centric integrate, k_tsdf_integrate_pix: a voxel is evaluated by the visitor of ITS OWN pixel only -- the
  // conservative candidate sets of neighbouring rows overlap, and the update is not idempotent)
  if (want_py >= 0 && py != want_py) return 0;
  const float2 dc = dct[px * im_h + py];  // (depth_im, color_im)[py * im_w + px], transposed copy (lt_tsdf::dct)
  const float new_rem = rem_im[py * im_w + px];  // (issued with it: one round trip, not two)
  const float depth_value = dc.x;
  if (depth_value == 0.f) return 0;
  const float depth_diff = depth_value - depth;
  if (depth_diff < -trunc_margin) return 0;
  const float dist = fminf(1.0f, depth_diff / trunc_margin);
  return tsdf_update<MERGE>(tsdf_vol, weight_vol, color_vol, rem_vol, voxel_idx, dist, obs_weight, dc.y, new_rem, fresh);
}

// Can voxel z of a FRESH plain column be written by the class-aware update?  On a fresh volume (the reference builds one
// per output scan, laserscan.py:886-887, and fuses number_of_scans = 1 into it by default) a voxel is written only inside
// the truncation band behind a surface -- depth in (D, D + trunc] of its pixel -- or, where the pixel carries colour 0 (the
// fresh volume's own colour: "same class"), anywhere in front of that band.  That is 1 voxel in 4 of those inside the
// field of view, and the exact evaluation costs ~150 vector instructions per 64 voxels against ~35 for this test:
// approximate depth (rsq), approximate pitch (odd polynomial, |s| <= 0.5: error < 1e-6 rad) -> the one or two image rows the
// exact projection can choose (+-0.25 row against an error below 0.01) -> their depth / colour with a margin far above
// the approximation errors.  CONSERVATIVE: it may only say yes too often; every candidate then runs the reference's
// expressions (tsdf_voxel), which decide.
__device__ __forceinline__ bool tsdf_band_candidate(int z, const col_plain& C, float oz, float voxel_size, int im_h,
                                                    int im_w, float trunc_margin, float kA, float kB, float sin_up_hi,
                                                    float sin_down_lo, const float2* __restrict__ dct) {
  // (branch-free, loads unconditional at clamped indices: the caller evaluates three voxels per lane at once and wants
  // their loads in flight together)
  const float pt_z = __fmaf_rn((float)z, voxel_size, oz);
  const float d2 = __fmaf_rn(pt_z, pt_z, C.rho2);
  // the voxel at the sensor, NaN, overflow, a voxel on the sensor's axis (rho < 2e-3 |z|, LT_AXIS_RHO): the exact path decides
  const bool odd = !(d2 > 0.f && d2 < 1e30f) || C.rho2 <= 4e-6f * d2;
  const float rinv = __builtin_amdgcn_rsqf(d2);
  const float s = pt_z * rinv, depth = d2 * rinv;
  const bool in_fov = !(s > sin_up_hi + 1e-4f || s < sin_down_lo - 1e-4f);  // (with margin)
  const float q = s * s;
  const float pitch = s * (1.0f + q * (0.16666667f + q * (0.075f + q * (0.044642857f + q * 0.030381944f))));
  const float py = __fmaf_rn(pitch, kA, kB);  // ~ proj_y * im_h
  const int r0 = min(max((int)floorf(py - 0.25f), 0), im_h - 1), r1 = min(max((int)floorf(py + 0.25f), 0), im_h - 1);
  const float eps = __fmaf_rn(4e-6f, depth, 1e-6f);
  // (depth, colour) pairs of the two rows from the transposed copy: the lanes of a column walk read neighbouring rows of
  // one image column -- a few lines per wave instead of one per lane
  const float2 p0 = dct[C.px * im_h + r0], p1 = dct[C.px * im_h + r1];
  const float D0 = p0.x, D1 = p1.x, c0 = p0.y, c1 = p1.y;
  const float f0 = D0 - depth, f1 = D1 - depth;
  // (written so that a NaN depth pixel -- which the reference's comparisons let through to the same-class test --
  // stays a candidate)
  const bool k0 = D0 != 0.f && !(f0 < -trunc_margin - eps) && (!(f0 >= eps) || c0 == 0.0f);
  const bool k1 = D1 != 0.f && !(f1 < -trunc_margin - eps) && (!(f1 >= eps) || c1 == 0.0f);
  const bool cand = odd || (in_fov && (k0 || k1));
  return cand;
}

// z range [z0, z1) of the voxel column (cx, cy) that can lie inside the vertical field of view -- conservative: the
// voxels left out would take tsdf_voxel's sine exit -- pt_z in [rho tan(fov_down) - pad, rho tan(fov_up) + pad].  The
// whole column for y = dim_y - 1 (its voxels may belong to another (x, y) by the reference's float index) and when the
// field of view is too steep for tangents.
struct col_geom {
  int dim_y, dim_z;
  float ox, oy, oz, voxel_size, tan_up, tan_down;
  int tan_ok;
};
__device__ __forceinline__ void col_zrange(const col_geom& G, int cx, int cy, int& z0, int& z1) {
  z0 = 0;
  z1 = G.dim_z;
  const float pt_x = __fmaf_rn((float)cx, G.voxel_size, G.ox), pt_y = __fmaf_rn((float)cy, G.voxel_size, G.oy);
  const float rho = sqrtf(pt_x * pt_x + pt_y * pt_y);
  if (cy != G.dim_y - 1 && G.tan_ok && rho > axis_rho_limit(G.oz, G.voxel_size, G.dim_z)) {  // (LT_AXIS_RHO: whole column)
    const float pad = 2.0f * G.voxel_size + 1e-3f * rho;
    const float zl = (rho * G.tan_down - pad - G.oz) / G.voxel_size, zh = (rho * G.tan_up + pad - G.oz) / G.voxel_size;
    z0 = max(0, (int)floorf(fminf(fmaxf(zl, -1.0f), (float)G.dim_z)));
    z1 = min(G.dim_z, (int)ceilf(fminf(fmaxf(zh, -1.0f), (float)G.dim_z)) + 1);
  }
}

// the n-th (0-based) set bit of m, or -1
__device__ __forceinline__ int nth_set_bit(unsigned long long m, int n) {
  for (int k = 0; k < n; ++k) m &= m - 1;
  return m ? __ffsll((long long)m) - 1 : -1;
}

// per voxel column (x, y): its image column px -- the kernel's own expressions on voxel_x = x, voxel_y = y -- or -1
// when no voxel of the column can pass the truncation test
__global__ __launch_bounds__(256) void k_tsdf_columns(int vol_dim_x, int vol_dim_y, float ox, float oy, float voxel_size,
                                                      int im_w, float trunc_margin, const float* __restrict__ colmax,
                                                      int* __restrict__ colinfo, int* __restrict__ chunk_live, col_geom G,
                                                      int vol_dim_z, unsigned* __restrict__ colz,
                                                      float* __restrict__ colrho2) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= vol_dim_x * vol_dim_y) return;
  const int x = c / vol_dim_y, y = c - x * vol_dim_y;
  const float pt_x = __fmaf_rn((float)x, voxel_size, ox);
  const float pt_y = __fmaf_rn((float)y, voxel_size, oy);
  const float yaw = -atan2f(pt_y, pt_x);
  float proj_x = (float)(0.5 * ((double)yaw / LT_PI_D + 1.0));
  proj_x *= (float)im_w;
  int px = (int)floorf(proj_x);
  px = min(im_w - 1, px);
  px = max(0, px);
  // every voxel of the column has depth = norm3df(x, y, z) >= rho_lo: the true norm is >= the true sqrt(x^2 + y^2), the
  // device library's norm3df is within an ulp of the one and rho within an ulp of the other, and rho_lo sits four ulp
  // below rho.  Float subtraction is monotonic, so depth_value - depth <= colmax - rho_lo for every non-zero pixel: if that
  // is already < -trunc_margin the column is dead.
  // A column without a non-zero pixel has colmax = -inf (every voxel leaves at `depth_value == 0`): dead by the same test.
  const float rho = sqrtf(__fmaf_rn(pt_y, pt_y, pt_x * pt_x));
  const float rho_lo = rho * 0.9999995f;
  const float cm = colmax[px];
  const bool dead = (cm - rho_lo) < -trunc_margin;
  // -2: a dead column with y = dim_y - 1 -- walked all the same, voxel by voxel (the reference's float index can put
  // voxels of the (x + 1, -1) "column" there), each finding its own px
  const int info = dead ? (y == vol_dim_y - 1 ? -2 : -1) : px;
  colinfo[c] = info;
  // the walk the integrate kernel will do in this column -- its z range, whether it is plain (col_plain_of), the
  // column's fma(pt_y, pt_y, pt_x^2) -- computed HERE, one lane per column, instead of by the 16 lanes that walk the
  // column there (two square roots and four IEEE divisions per column: a sixth of that kernel's vector instructions)
  if (info != -1) {
    int z0, z1;
    col_zrange(G, x, y, z0, z1);
    colinfo[c] = info;  // (col_plain_of reads it)
    const col_plain C = col_plain_of(x, y, z0, z1, vol_dim_y, vol_dim_z, ox, oy, voxel_size, colinfo);
    colz[c] = (unsigned)z0 | ((unsigned)z1 << 15) | (C.plain ? 0x80000000u : 0u);  // z < 2^15 (lt_tsdf_create)
    colrho2[c] = C.rho2;
  }
  // one flag per chunk of 64 columns (= this wave): does the integrate kernel have anything to walk there?  Three chunks
  // in four have not, and a workgroup that reads its flag through the scalar cache is gone in a fraction of the 1.6 us
  // it took to load and ballot 64 table entries
  const unsigned long long any = __ballot(info != -1);
  if ((threadIdx.x & 63) == 0) chunk_live[c >> 6] = any ? 1 : 0;
}

// A wave looks at the table entries of 64 voxel columns at once and then walks only the columns that are not dead:
// FOUR columns at a time, 16 lanes each with z along the lanes (a column's walk is a chain of dependent loads -- table
// entry, depth pixel, the voxel's fields -- and only the z range inside the field of view, 20 - 60 voxels, has work:
// one column per wave iteration left the kernel latency-bound at 0.66 ms for the default volume).  Columns with
// y = dim_y - 1 are always walked, whole: they are where the reference's float voxel index can misplace a voxel into
// the (x + 1, -1) column, which the table does not describe -- tsdf_voxel() handles every voxel by the reference's own
// decomposition.  The signs of the values written update the column's sign bits (bit b of word k = voxel z = 64 k + b
// is NOT above the level 0), which marching cubes reads instead of the float field; a 16-aligned chunk of z lies inside one word and
// only this quarter wave works on this column, so the read-modify-write needs no atomic.
template <bool MERGE>
__global__ __launch_bounds__(256) LT_TSDF_WAVES_ATTR void k_tsdf_integrate_cols(
    float* __restrict__ tsdf_vol, float* __restrict__ weight_vol, float* __restrict__ color_vol,
    float* __restrict__ rem_vol, int vol_dim_x, int vol_dim_y, int vol_dim_z, float ox, float oy, float oz,
    float voxel_size, int im_h, int im_w, float trunc_margin, float obs_weight, float fov_up, float fov_down,
    float sin_up_hi, float sin_down_lo, const float* __restrict__ color_im, const float* __restrict__ depth_im,
    const float* __restrict__ rem_im, const int* __restrict__ colinfo, unsigned* __restrict__ col_epoch,
    unsigned epoch, col_geom G, unsigned long long* __restrict__ sign_bits, int words_z,
    unsigned* __restrict__ col_zw, const float2* __restrict__ dct, float kA, float kB,
    const int* __restrict__ chunk_live, const unsigned* __restrict__ colz, const float* __restrict__ colrho2,
    int all_written, unsigned long long* __restrict__ dbg, unsigned* __restrict__ chunk_epoch) {
  const unsigned long long t_dbg = dbg ? (unsigned long long)wall_clock64() : 0ull;  // (LIDARHIP_DEBUG_TSDF: per-wave stamps)
  // candidates of the band test (fresh plain columns, class-aware update) wait here, per wave, until 64 are together:
  // then lane j evaluates candidate j exactly
  __shared__ int q_col[4][128], q_z[4][128], q_px[4][128];
  __shared__ float q_rho2[4][128];
  const int lane = threadIdx.x & 63, grp = lane >> 4, gl = lane & 15, wv = threadIdx.x >> 6;
  const int n_cols = vol_dim_x * vol_dim_y;
  const int n_chunks = (n_cols + 63) / 64;
  const unsigned long long lanes_below = (1ull << lane) - 1ull;
  // (kA, kB: the band test's row = pitch * kA + kB, from the host: a quarter of a million waves need not divide for them)
  int qn = 0;  // (wave-uniform)
  auto flush = [&](int n) {  // the exact evaluation + update of the queue's last n (<= 64) candidates, one per lane
    __builtin_amdgcn_wave_barrier();
    if (lane < n) {
      const int e = qn - n + lane;
      const int col = q_col[wv][e], z = q_z[wv][e];
      col_plain Cq;
      Cq.plain = true; Cq.px = q_px[wv][e]; Cq.rho2 = q_rho2[wv][e]; Cq.col = col;
      const int code = tsdf_voxel<MERGE>(col * vol_dim_z + z, tsdf_vol, weight_vol, color_vol, rem_vol, vol_dim_x,
                                         vol_dim_y, vol_dim_z, ox, oy, oz, voxel_size, im_h, im_w, trunc_margin, obs_weight,
                                         fov_up, fov_down, sin_up_hi, sin_down_lo, color_im, depth_im, rem_im, colinfo,
                                         col_epoch, epoch, true, Cq, z, dct);
      if (code) {  // the sign of the value written -> the column's sign bit (other lanes may hold voxels of the same word)
        unsigned long long* w = sign_bits + (size_t)col * words_z + (z >> 6);
        const unsigned long long bit = 1ull << (z & 63);
        if (code == 2) atomicOr(w, bit);
        else atomicAnd(w, ~bit);
      }
    }
    qn -= n;
    __builtin_amdgcn_wave_barrier();
  };
  // A workgroup takes a chunk of 64 columns and its four waves share the chunk's live columns, every fourth quad each: the
  // cost of a chunk ranges from nothing to 200 us (a wall: every voxel of every column a candidate), and with a chunk per
  // wave the launch ended in a 110 us tail of a few such waves (per-wave stamps: tools/tsdf_wave_times.py)
  for (int chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    if (!chunk_live[chunk]) continue;  // (uniform: a scalar load)
    const int c = chunk * 64 + lane;
    bool live = false, written = false;
    if (c < n_cols) {
      live = colinfo[c] != -1;  // (-2: the dead columns with y = dim_y - 1, see k_tsdf_columns)
      // before this launch: its voxels hold something else than the initial values (all_written: the caller wrote into
      // the volumes through the raw pointers, lt_tsdf_touch, or a kernel without stamps did: nothing may be folded)
      written = all_written != 0 || col_epoch[c] == epoch;
    }
    unsigned long long m = __ballot(live);
    const unsigned long long wm = __ballot(written);
    for (int quad = 0; m; ++quad) {
      const int bit = nth_set_bit(m, grp);  // this quarter wave's column of the next four
      m &= m - 1; m &= m - 1; m &= m - 1; m &= m - 1;
      if ((quad & 3) != wv) continue;  // (another wave of the workgroup)
      int z0 = 0, z1 = 0, cc = 0;
      col_plain C;
      C.plain = false; C.px = -2; C.rho2 = 0.f; C.col = 0;
      const bool fresh = bit >= 0 && !((wm >> bit) & 1ull);
      if (bit >= 0) {
        cc = chunk * 64 + bit;
        const unsigned zz = colz[cc];  // (k_tsdf_columns)
        z0 = (int)(zz & 0x7FFFu);
        z1 = (int)((zz >> 15) & 0xFFFFu);
        C.plain = (zz >> 31) != 0u;
        C.px = colinfo[cc];
        C.rho2 = colrho2[cc];
        C.col = cc;
      }
      int wz_lo = 0x7fff, wz_hi = -1;  // z range this launch writes in the column (any field; uniform over the group)
      // (uniform trip count over the wave: the ballots below need every lane)
      int trips = (z1 - (z0 & ~15) + 15) >> 4;
      trips = max(trips, __shfl_xor(trips, 16, 64));
      trips = max(trips, __shfl_xor(trips, 32, 64));
#ifdef LT_TSDF_NO_BAND  // A/B (LIDARHIP_EXTRA_FLAGS=-DLT_TSDF_NO_BAND): every voxel through the exact evaluation
      const bool band = false;
#else
      const bool band = MERGE && fresh && C.plain && kA == kA;  // (per quarter wave; kA is NaN when the test is off)
#endif
      const bool any_direct = __ballot(bit >= 0 && !band) != 0ull;  // (wave-uniform: usually false on a fresh volume)
      for (int k0 = 0; k0 < trips; k0 += 3) {
        // band columns: the candidate tests of three chunks of z at once (their loads in flight together: a wave's life
        // is a chain of load round trips, and this makes it a third as long)
        bool cd[3] = {false, false, false};
        if (band) {
#pragma unroll
          for (int u = 0; u < 3; ++u) {
            const int z = (z0 & ~15) + 16 * (k0 + u) + gl;
            const bool c1 = tsdf_band_candidate(min(z, vol_dim_z - 1), C, oz, voxel_size, im_h, im_w, trunc_margin, kA, kB,
                                                sin_up_hi, sin_down_lo, dct);
            cd[u] = c1 && z >= z0 && z < z1;
          }
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const int k = k0 + u;
          if (k >= trips) break;  // (wave-uniform)
          const int zc = (z0 & ~15) + 16 * k, z = zc + gl;
          const bool cand = cd[u];
          unsigned long long wrote = 0ull, neg = 0ull;
          if (any_direct) {
            int code = 0;
            if (!band && z >= z0 && z < z1)
              code = tsdf_voxel<MERGE>(cc * vol_dim_z + z, tsdf_vol, weight_vol, color_vol, rem_vol, vol_dim_x, vol_dim_y,
                                       vol_dim_z, ox, oy, oz, voxel_size, im_h, im_w, trunc_margin, obs_weight, fov_up,
                                       fov_down, sin_up_hi, sin_down_lo, color_im, depth_im, rem_im, colinfo, col_epoch,
                                       epoch, fresh, C, z, dct);
            const unsigned long long wrote_w = __ballot(code != 0), neg_w = __ballot(code == 2);
            wrote = (wrote_w >> (16 * grp)) & 0xFFFFull;
            neg = (neg_w >> (16 * grp)) & 0xFFFFull;
          }
          const unsigned long long cand_w = __ballot(cand);
          const unsigned long long cands = (cand_w >> (16 * grp)) & 0xFFFFull;
          if (wrote | cands) {  // (a candidate counts as written for the column's stamp and z range: a superset is safe)
            wz_lo = min(wz_lo, zc + (__ffsll((long long)(wrote | cands)) - 1));
            wz_hi = max(wz_hi, zc + 63 - __clzll((long long)(wrote | cands)));
            if (wrote && gl == 0) {
              unsigned long long* w = sign_bits + (size_t)cc * words_z + (zc >> 6);
              const int sh = zc & 63;
              *w = (*w & ~(wrote << sh)) | (neg << sh);
            }
          }
          if (cand_w) {  // (wave-uniform)
            if (cand) {
              const int e = qn + __popcll(cand_w & lanes_below);
              q_col[wv][e] = cc; q_z[wv][e] = z; q_px[wv][e] = C.px; q_rho2[wv][e] = C.rho2;
            }
            qn += __popcll(cand_w);
            if (qn >= 64) flush(64);
          }
        }
      }
      // the column becomes dirty only if something was written: stamp it and widen its written z range (what the next
      // reset has to re-initialise); this quarter wave owns the column, no atomics
      if (bit >= 0 && wz_hi >= 0 && gl == 0) {
        if (!fresh) {
          wz_lo = min(wz_lo, LT_ZW_LO(col_zw[cc]));
          wz_hi = max(wz_hi, LT_ZW_HI(col_zw[n_cols + cc]));
        }
        col_zw[cc] = (unsigned)(0x7FFF - wz_lo);
        col_zw[n_cols + cc] = (unsigned)(wz_hi + 1);
        col_epoch[cc] = epoch;
        chunk_epoch[cc >> 6] = epoch;  // (what reset and marching cubes look at first)
      }
    }
  }
  if (qn > 0) flush(qn);
  if (dbg && (threadIdx.x & 63) == 0) {
    const size_t w = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) % LT_TSDF_DBG_WAVES;
    dbg[2 * w] = t_dbg;
    dbg[2 * w + 1] = (unsigned long long)wall_clock64() - t_dbg;
  }
}

// re-initialise the dirty columns -- the z range that was actually written (col_zw, kept by integrate) -- and their sign
// words: a wave reads 64 stamps at once and walks the dirty columns, four at a time (16 lanes each: a written range is
// typically the truncation band, ~10 voxels).
// Three chunks in four of a street scene are clean: a wave takes LT_RESET_CHUNKS_PER_WAVE chunks, a stride apart (written
// chunks come in clusters, lt_deal_count) -- its first lanes read one stamp each, a ballot finds the written ones.  (Not
// faster than a wave per chunk, 23 against 25 us: a written chunk is a chain of three dependent round trips -- chunk stamp,
// column stamps, column ranges -- and the kernel is as long as its slowest waves; but an eighth of the waves.)
#ifndef LT_RESET_CHUNKS_PER_WAVE
#define LT_RESET_CHUNKS_PER_WAVE 8
#endif
__global__ __launch_bounds__(256) void k_tsdf_reset_cols(float* __restrict__ tsdf, float* __restrict__ weight,
                                                         float* __restrict__ color, float* __restrict__ rem,
                                                         int n_cols, int dim_z, const unsigned* __restrict__ col_epoch,
                                                         unsigned epoch, unsigned long long* __restrict__ sign_bits,
                                                         unsigned* __restrict__ col_zw,
                                                         const unsigned* __restrict__ chunk_epoch) {
  const int lane = threadIdx.x & 63, grp = lane >> 4, gl = lane & 15;
  const int n_chunks = (n_cols + 63) / 64;
  const int words_z = (dim_z + 63) / 64;
  const int n_waves = gridDim.x * 4, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  unsigned long long live;
  {
    const int ch = lane * n_waves + wave;
    live = __ballot(lane < LT_RESET_CHUNKS_PER_WAVE && ch < n_chunks && chunk_epoch[min(ch, n_chunks - 1)] == epoch);
  }
  while (live) {
    const int chunk = (__ffsll((long long)live) - 1) * n_waves + wave;  // (wave-uniform)
    live &= live - 1;
    const int c = chunk * 64 + lane;
    const bool dirty = c < n_cols && col_epoch[c] == epoch;
    unsigned zw = 0x7FFFu;  // (lo | hi << 16; empty)
    if (dirty) {
      const int hi = LT_ZW_HI(col_zw[n_cols + c]);
      if (hi >= 0) zw = (unsigned)LT_ZW_LO(col_zw[c]) | ((unsigned)hi << 16);
      for (int k = 0; k < words_z; ++k) sign_bits[(size_t)c * words_z + k] = 0ull;
      col_zw[c] = 0u;  // (a clean column holds the empty range)
      col_zw[n_cols + c] = 0u;
    }
    unsigned long long m = __ballot(dirty);
    while (m) {
      const int bit = nth_set_bit(m, grp);
      m &= m - 1; m &= m - 1; m &= m - 1; m &= m - 1;
      const unsigned r = __shfl(zw, max(bit, 0), 64);  // (every lane takes part: the source lane must be active)
      if (bit < 0) continue;
      const int z0 = (int)(r & 0xFFFFu), z1 = min((int)(r >> 16), dim_z - 1);
      const size_t base = (size_t)(chunk * 64 + bit) * dim_z;
      for (int z = z0 + gl; z <= z1; z += 16) {
        *LT_VOX(tsdf, base + z) = make_float4(1.0f, 0.0f, 0.0f, 0.0f);
      }
    }
  }
}

// =========================================================================================================================
// The pixel-centric integrate of a FRESH volume (the reference builds a new TSDFVolume per output scan and, by default,
// fuses ONE observation into it: laserscan.py:886-899, config `number_of_scans: 1`).
//
// k_tsdf_integrate_cols asks, for every voxel inside the field of view of every live column (48 M on the default volume),
// whether its pixel's depth puts it into the truncation band; what is finally written is 0.8 M voxels.  On a fresh volume the
// class-aware update (fusion_lidar.py:177-213) writes a voxel only if its pixel carries a depth D != 0 and
//     the voxel lies in the band behind the surface,  D < depth <= D + trunc_margin                 (dist < 0 = dist_old), or
//     the pixel's colour is 0, the fresh volume's own ("same class"), and depth <= D + trunc_margin (anywhere in front).
// So the work can be driven by the 131 072 PIXELS instead: the voxel columns (x, y) that project into image column px --
// a property of the volume's geometry and the image width alone -- are kept sorted by their horizontal distance rho (the
// WEDGE TABLE, built once per volume and image width: k_wd_keys -> radix sort of lt_build.hip -> k_wd_finish); the visitor of
// pixel (row, px) finds, by two binary searches, the columns whose rho can hold a voxel of that row at a depth in the band
// [rho = depth cos(pitch)], and for each of them the short z interval (row's pitch range x band's depth range, both with
// margins far above the float error of the reference's expressions).  Every voxel of that SUPERSET runs the reference's own
// expressions (tsdf_voxel), which decide -- and which evaluate a voxel only for the visitor of its own exact row (want_py):
// neighbouring rows' supersets overlap and the update must happen exactly once.  Lanes own pixels (a wave = 64 consecutive
// rows of one image column: they walk the same wedge); colour-0 pixels, whose candidates are the whole ray, are worked off by
// the whole wave, columns across the lanes.  Columns whose voxels the reference's float index arithmetic can misplace
// (y = dim_y - 1 beyond 2^24 voxels, col_plain) are not in the table: k_tsdf_integrate_quirk evaluates them voxel by voxel.
// Bit-identical to the column walk and to the one-thread-per-voxel restatement (tests/test_tsdf_gpu.py).
#ifdef LT_PIX_STAMP  // (-DLT_PIX_STAMP; tools/tsdf_written_times.py --pix): per workgroup of k_tsdf_integrate_pix -- start, end of phase A,
// end of the first chunk's pairs + scan, end of the voxel rounds, end (plain stores: shared counters measure themselves)
__device__ unsigned long long g_pix_stamp[5 << 12];
extern "C" int lt_debug_pix_stamps(unsigned long long* out, int n_wgs) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pix_stamp), (size_t)min(n_wgs, 1 << 12) * 5 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#define LT_PIX_MARK(i) do { if (tid == 0 && blockIdx.x < (1 << 12)) g_pix_stamp[5 * blockIdx.x + (i)] = wall_clock64(); } while (0)
#else
#define LT_PIX_MARK(i) do { } while (0)
#endif
struct wd_geom {
  int dim_x, dim_y, dim_z;
  float ox, oy, oz, vs;
  int im_w, rho_bits;
  float qscale;
};
#define LT_WD_QUIRK_KEY 0x3FFFFFFFu
#define LT_WD_QUIRK_FLAG 0x40000000  // in wd_px[c]: the column is evaluated by k_tsdf_integrate_quirk, voxel by voxel

__global__ __launch_bounds__(256) void k_wd_keys(wd_geom G, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                 int* __restrict__ wd_px, int* __restrict__ n_quirk) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= G.dim_x * G.dim_y) return;
  const int x = c / G.dim_y, y = c - x * G.dim_y;
  const float pt_x = __fmaf_rn((float)x, G.vs, G.ox), pt_y = __fmaf_rn((float)y, G.vs, G.oy);
  const float yaw = -atan2f(pt_y, pt_x);  // the very expressions of the reference kernel (:132-141), as k_tsdf_columns
  float proj_x = (float)(0.5 * ((double)yaw / LT_PI_D + 1.0));
  proj_x *= (float)G.im_w;
  int px = (int)floorf(proj_x);
  px = min(G.im_w - 1, px);
  px = max(0, px);
  // every voxel index of the column decomposes, by the reference's float division (:95-98), to this column's x?
  const float dyz = (float)(G.dim_y * G.dim_z);
  const int i0 = c * G.dim_z, i1 = c * G.dim_z + G.dim_z - 1;
  const bool plain = floorf(((float)i0) / dyz) == (float)x && floorf(((float)i1) / dyz) == (float)x;
  const float rho = sqrtf(__fmaf_rn(pt_y, pt_y, pt_x * pt_x));
  const bool quirk = !plain || y == G.dim_y - 1 || rho <= axis_rho_limit(G.oz, G.vs, G.dim_z);
  wd_px[c] = px | (quirk ? LT_WD_QUIRK_FLAG : 0);
  const unsigned qmax = (1u << G.rho_bits) - 2u;  // (all-ones is left to LT_WD_QUIRK_KEY)
  const unsigned q = (unsigned)fminf(rho * G.qscale, (float)qmax);
  keys[c] = quirk ? LT_WD_QUIRK_KEY : (((unsigned)px << G.rho_bits) | q);
  vals[c] = (uint32_t)c;
  if (quirk) atomicAdd(n_quirk, 1);
}

// sorted (key, column) -> table entries (column, rho^2 bits; column -1 for the quirk tail), the rho quantum of every entry
// (what the binary searches compare), the first entry of every image column's wedge
__global__ __launch_bounds__(256) void k_wd_finish(wd_geom G, const uint32_t* __restrict__ keys,
                                                   const uint32_t* __restrict__ vals, int n, int2* __restrict__ ent,
                                                   uint32_t* __restrict__ wkey, int* __restrict__ wd_start) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p > n) return;
  auto bin = [&](uint32_t k) { return (int)min(k >> G.rho_bits, (uint32_t)G.im_w); };
  const int lo = p == 0 ? 0 : bin(keys[p - 1]) + 1;
  const int hi = p == n ? G.im_w : bin(keys[p]);
  for (int b = lo; b <= hi; ++b) wd_start[b] = p;
  if (p < n) {
    const uint32_t k = keys[p];
    const int c = (int)vals[p];
    const int x = c / G.dim_y, y = c - x * G.dim_y;
    const float pt_x = __fmaf_rn((float)x, G.vs, G.ox), pt_y = __fmaf_rn((float)y, G.vs, G.oy);
    ent[p] = make_int2(k == LT_WD_QUIRK_KEY ? -1 : c, __float_as_int(__fmaf_rn(pt_y, pt_y, pt_x * pt_x)));
    wkey[p] = k == LT_WD_QUIRK_KEY ? 0xFFFFFFFFu : (k & ((1u << G.rho_bits) - 1u));
  }
}

// merge z into the written range of column c (lo | hi << 16), stamp the column; any number of concurrent writers
__device__ __forceinline__ void col_mark_written(unsigned* __restrict__ col_zw, unsigned* __restrict__ col_epoch,
                                                 unsigned* __restrict__ chunk_epoch, unsigned epoch, int n_cols, int c,
                                                 int z, int z_last) {
  atomicMax(&col_zw[c], (unsigned)(0x7FFF - z));       // (results unused: fire and forget)
  atomicMax(&col_zw[n_cols + c], (unsigned)(z_last + 1));
  col_epoch[c] = epoch;
  chunk_epoch[c >> 6] = epoch;
}

// One workgroup = 64 consecutive pixels of the transposed image (rows of one image column: they walk the same wedge), its
// four waves share the work, which is flattened twice so that no lane waits for a far pixel's long lists:
//   A  wave 0, lane = pixel: the two binary searches -> the pixel's run of table entries [k0, k0 + n); prefix sum of n
//   B  per chunk of 512 (pixel, column) PAIRS, two per thread: the pair's z interval (z_range) -> prefix sum of the
//      interval lengths; then the chunk's VOXELS, one per thread and round: the reference's expressions (tsdf_voxel with
//      want_py), the sign bit, the column's written range
// (the first version gave every lane its own pixel from start to end: a pixel at 60 m has 30 columns x 10 voxels, one at
// 5 m four voxels altogether -- 368 us with two waves per SIMD, most lanes idle; colour-0 pixels, whose run is the wedge up
// to the band, are simply long runs here)
#define LT_PIX_CHUNK 512    // pairs per chunk (two per thread)
#ifndef LT_PIX_AGG
#define LT_PIX_AGG 1024
#endif
// LT_PIX_AGG: table entries whose written ranges a workgroup merges in LDS before it touches col_zw
#define LT_PIX_STAGE 3840   // rho quanta staged in LDS for the binary searches (a wedge of the default volume: 2000 - 3900);
                            // 15 KB: with the other arrays 19.5 KB per workgroup = EIGHT per CU -- the 2048 workgroups of a
                            // 64 x 2048 image are resident at once (at 4096 entries, seven per CU: a second round)
#ifndef LT_PIX_WPE
#define LT_PIX_WPE 5
#endif
template <bool MERGE, bool VCOUNT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LT_PIX_WPE, 8))) void k_tsdf_integrate_pix(
    float* __restrict__ tsdf_vol, float* __restrict__ weight_vol, float* __restrict__ color_vol,
    float* __restrict__ rem_vol, int vol_dim_x, int vol_dim_y, int vol_dim_z, float ox, float oy, float oz,
    float voxel_size, float inv_vs, int im_h, int im_w, float trunc_margin, float obs_weight, float fov_up,
    float fov_down, float sin_up_hi, float sin_down_lo, const float* __restrict__ color_im,
    const float* __restrict__ depth_im, const float* __restrict__ rem_im, const int* __restrict__ wd_px,
    unsigned* __restrict__ col_epoch, unsigned epoch, unsigned long long* __restrict__ sign_bits, int words_z,
    unsigned* __restrict__ col_zw, unsigned* __restrict__ chunk_epoch, const float2* __restrict__ dct,
    const float4* __restrict__ rowtab, const int* __restrict__ wd_start, const int2* __restrict__ wd_ent, const uint32_t* __restrict__ wd_key, int rho_bits,
    float qscale, const unsigned* __restrict__ zw_snap, unsigned long long* __restrict__ dbg) {
  // zw_snap (NULL on a fresh volume): the columns' written z ranges as they were BEFORE this observation.  The voxels inside
  // them may hold anything -- k_tsdf_integrate_written has evaluated every one of them -- and are skipped here; everything
  // else still holds the initial values, so this kernel's superset argument (and `fresh` = true) stands.
  const int n_cols_all = vol_dim_x * vol_dim_y;
  // per pixel of the workgroup
  __shared__ int p_k0[64], p_pre[65], p_r[64], p_px[64];
  __shared__ float p_tlo[64], p_thi[64], p_dlo[64], p_dhi[64];
  // per pair of the chunk (one block: phase A borrows it as the staging area of the wedge's rho quanta)
  __shared__ int c_buf[LT_PIX_STAGE > 4 * LT_PIX_CHUNK + 1 ? LT_PIX_STAGE : 4 * LT_PIX_CHUNK + 1];
  int* const c_col = c_buf;
  int* const c_z0 = c_buf + LT_PIX_CHUNK;
  int* const c_src = c_buf + 2 * LT_PIX_CHUNK;
  int* const c_pre = c_buf + 3 * LT_PIX_CHUNK;  // [LT_PIX_CHUNK + 1]
  __shared__ float c_rho2[LT_PIX_CHUNK];
  __shared__ int wsum[4];
  // the columns' written ranges, merged in LDS first: the pairs of a workgroup fall on a few dozen table entries of ONE
  // wedge (a wall's column is visited by dozens of rows), and two global atomics + two stores per PAIR were 43 of the
  // kernel's 92 us (tools/pix_sections.sh).  a_lo / a_hi are indexed by table entry - a_k0 and hold the col_zw encoding
  // (0x7fff - lo, hi + 1; 0 = nothing); a workgroup whose pairs span more than LT_PIX_AGG entries marks directly.
  __shared__ unsigned a_lo[LT_PIX_AGG], a_hi[LT_PIX_AGG];
  __shared__ int a_k0, a_span;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t qmax = (1u << rho_bits) - 2u;
  const int n_pix = im_h * im_w;
  const uint32_t* const skey = (const uint32_t*)c_buf;  // staged quanta of [stage0, stage0 + n_stage)
  int stage0 = 0, n_stage = 0;
  // first table index in [a, b) whose rho quantum is >= q (b if none): the wedges are sorted by it.  The 64 pixels of a
  // workgroup search the SAME wedge(s): 22 dependent global loads per pixel became one coalesced copy into LDS
  bool staged = false;  // (workgroup-uniform: the whole search range is in LDS)
  auto lower = [&](int a, int b, uint32_t q) {
    if (staged) {
      while (a < b) {
        const int m = (a + b) >> 1;
        if (skey[m - stage0] < q) a = m + 1;
        else b = m;
      }
    } else {
      while (a < b) {
        const int m = (a + b) >> 1;
        if (wd_key[m] < q) a = m + 1;
        else b = m;
      }
    }
    return a;
  };
  // (An item's chunks dealt to several workgroups, each repeating phase A, were measured: 2 parts 88 us, 4 parts 129 against
  // 65 -- an item is 6 us of phase A, 3 of its first chunk's pairs and 9 of voxel rounds: tools/tsdf_written_times.py --pix.)
  for (int p0 = blockIdx.x * 64; p0 < n_pix; p0 += gridDim.x * 64) {  // (workgroup-uniform)
    const unsigned long long tm0 = VCOUNT ? (unsigned long long)wall_clock64() : 0ull;  // (100 MHz; debug)
    LT_PIX_MARK(0);
    // ---- A: the pixels' runs of table entries ------------------------------------------------------------------------
    // (wave 0's own loads -- its pixel, its row's table entry, its wedge's extent -- are issued BEFORE the staging and its
    // barrier: one dependent round trip less per item)
    const int pA = p0 + lane;
    const bool inA = wave == 0 && pA < n_pix;
    const int pxA = inA ? pA / im_h : 0, rA = inA ? pA - pxA * im_h : 0;
    const float2 dcA = inA ? dct[pA] : make_float2(0.f, 1.f);
    const float4 rowA = rowtab[rA];
    const int s0A = wd_start[pxA], s1A = wd_start[pxA + 1];
    {  // stage the quanta of the wedges these 64 pixels search (one image column when im_h is a multiple of 64)
      const int px_a = p0 / im_h, px_b = min(p0 + 63, n_pix - 1) / im_h;
      stage0 = wd_start[px_a];
      n_stage = wd_start[px_b + 1] - stage0;
      staged = n_stage <= LT_PIX_STAGE;  // (a longer run of wedges is searched in global memory)
#ifdef LT_PIX_NO_STAGE  // (timing experiment)
      staged = false;
#endif
      if (staged)
      {  // all loads first, then the LDS stores: the rolled loop waited for every load before it issued the next
        // (up to 15 dependent round trips; phase A 6.6 -> 5.8 us of an item's ~20 together with the hoisted loads above)
        constexpr int NS = (LT_PIX_STAGE + 255) / 256;
        uint32_t sv[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) sv[k] = tid + 256 * k < n_stage ? wd_key[stage0 + tid + 256 * k] : 0u;
#pragma unroll
        for (int k = 0; k < NS; ++k)
          if (tid + 256 * k < n_stage) c_buf[tid + 256 * k] = (int)sv[k];
      }
      __syncthreads();
    }
    if (wave != 0) {
      for (int i = tid - 64; i < LT_PIX_AGG; i += 192) { a_lo[i] = 0u; a_hi[i] = 0u; }
    }
    if (wave == 0) {
      const bool in = inA;  // pixel (row r, column px) at dct[px * im_h + r]
      const int px = pxA, r = rA;
      const float2 dc = dcA;
      const float D = dc.x;
      const float4 row = rowA;  // (tan_lo, tan_hi, cos_min, cos_max); tan_lo > tan_hi: no voxel can take this row
      const bool row_ok = row.x <= row.y;
      const bool finite = D == D && fabsf(D) < 1e30f;
      // the reference leaves at depth_value == 0; with another colour than 0 only the band is written (and nothing at all
      // through a NaN / infinite depth: dist = 1); colour 0: everything in front of D + trunc (all of it for NaN / inf)
      const bool zero_class = in && row_ok && D != 0.f && dc.y == 0.0f;
      const bool normal = in && row_ok && D != 0.f && dc.y != 0.0f && finite;
      const float eps = __fmaf_rn(4e-6f, fabsf(D) + trunc_margin, 1e-6f);
      const float d_hi = finite ? D + trunc_margin + eps : 3e38f;
      const float d_lo = zero_class ? 0.f : D - eps;
      const int s0 = s0A, s1 = s1A;
      int k = 0, kend = 0;
      if ((normal || zero_class) && d_hi > 0.f) {
        const float rho1 = fmaxf(d_lo, 0.f) * row.z * 0.999999f;
        const float f1 = floorf(rho1 * qscale) - 2.f;
        const uint32_t q1 = (uint32_t)fminf(fmaxf(f1, 0.f), (float)qmax);
        k = lower(s0, s1, q1);
        kend = s1;
        if (d_hi < 3e38f) {
          const float f2 = floorf(d_hi * row.w * 1.000001f * qscale) + 2.f;
          const uint32_t q2 = (uint32_t)fminf(fmaxf(f2, 0.f), (float)qmax);
          kend = lower(k, s1, q2 + 1u);
        }
      }
      const int cnt = kend - k;
      int inc = cnt;  // inclusive wave scan
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
      }
      p_k0[lane] = k; p_pre[lane] = inc - cnt; p_r[lane] = r; p_px[lane] = px;
      p_tlo[lane] = row.x; p_thi[lane] = row.y; p_dlo[lane] = d_lo; p_dhi[lane] = d_hi;
      if (lane == 63) p_pre[64] = inc;
      // the span of table entries the workgroup's pairs fall on
      int kmin = cnt > 0 ? k : 0x7fffffff, kmax = cnt > 0 ? kend : 0;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        kmin = min(kmin, __shfl_xor(kmin, o, 64));
        kmax = max(kmax, __shfl_xor(kmax, o, 64));
      }
      if (lane == 0) { a_k0 = kmin; a_span = kmax > kmin ? kmax - kmin : 0; }
    }
    __syncthreads();
    const int T = p_pre[64];
    const int agg_k0 = a_k0, agg_span = a_span;
    const bool agg = agg_span <= LT_PIX_AGG;  // (workgroup-uniform)
    if (VCOUNT && tid == 0) atomicAdd(&dbg[4], (unsigned long long)wall_clock64() - tm0);  // phase A
    LT_PIX_MARK(1);
    // ---- B: chunks of pairs ---------------------------------------------------------------------------------------------
    for (int base = 0; base < T; base += LT_PIX_CHUNK) {
      constexpr int PPT = LT_PIX_CHUNK / 256;  // pairs per thread
      int len[PPT];
#pragma unroll
      for (int u = 0; u < PPT; ++u) {
        const int slot = u * 256 + tid, i = base + slot;
        len[u] = 0;
        if (i < T) {
          int sidx = 0;  // largest s with p_pre[s] <= i
#pragma unroll
          for (int st = 32; st >= 1; st >>= 1)
            if (p_pre[sidx + st] <= i) sidx += st;
          const int kk = p_k0[sidx] + (i - p_pre[sidx]);
          const int2 e = wd_ent[kk];
          int z = 1, zend = 0;
          if (e.x >= 0) {  // (the quirk tail of the last wedge carries column -1)
            const float rho2 = __int_as_float(e.y), rho = sqrtf(rho2);
            const float d_lo = p_dlo[sidx], d_hi = p_dhi[sidx];
            const float hi2 = d_hi * d_hi - rho2;
            if (hi2 >= 0.f) {  // (else the whole column lies beyond the band; false also for NaN)
              const float zmax = sqrtf(hi2) * 1.000001f + 1e-6f;
              const float lo2 = d_lo > 0.f ? d_lo * d_lo - rho2 : -1.f;
              const float zmin = lo2 > 0.f ? fmaxf(sqrtf(lo2) * 0.999999f - 1e-6f, 0.f) : 0.f;
              const float za = rho * p_tlo[sidx], zb = rho * p_thi[sidx];  // pt_z of the row in this column
              float lo, hi;
              if (za >= 0.f) { lo = fmaxf(za, zmin); hi = fminf(zb, zmax); }
              else if (zb <= 0.f) { lo = fmaxf(za, -zmax); hi = fminf(zb, -zmin); }
              else { lo = fmaxf(za, -zmax); hi = fminf(zb, zmax); }
              if (lo <= hi) {
                const float fz0 = ceilf((lo - oz) * inv_vs - 0.02f), fz1 = floorf((hi - oz) * inv_vs + 0.02f);
                z = (int)fmaxf(fz0, 0.f);
                zend = (int)fminf(fz1, (float)(vol_dim_z - 1));
              }
            }
          }
          int partial = 0;
          if (zw_snap && zend >= z) {  // the part of the interval the column pass has done already
            const int olo = LT_ZW_LO(zw_snap[e.x]), ohi = LT_ZW_HI(zw_snap[n_cols_all + e.x]);
            if (ohi >= olo) {
              if (z >= olo && zend <= ohi) zend = z - 1;      // all of it (the usual case of a repeated observation)
              else if (z >= olo && z <= ohi) z = ohi + 1;     // its lower end
              else if (zend >= olo && zend <= ohi) zend = olo - 1;  // its upper end
              else if (z < olo && zend > ohi) partial = 0x40000000;  // a hole in the middle: decided per voxel
            }
          }
          len[u] = max(zend - z + 1, 0);
          c_col[slot] = e.x; c_rho2[slot] = __int_as_float(e.y); c_z0[slot] = z; c_src[slot] = sidx | partial;
          // the column's written range and stamp once per PAIR, for the whole candidate interval (a superset is safe, as in
          // the column walk) -- per written voxel, the ten threads holding one column's band voxels fought over one word
#ifndef LT_PIX_NO_MARK  // (timing experiment)
          if (len[u] > 0) {
            if (agg) {
              atomicMax(&a_lo[kk - agg_k0], (unsigned)(0x7FFF - z));
              atomicMax(&a_hi[kk - agg_k0], (unsigned)(zend + 1));
            } else {
              col_mark_written(col_zw, col_epoch, chunk_epoch, epoch, vol_dim_x * vol_dim_y, e.x, z, zend);
            }
          }
#endif
        }
      }
      // exclusive prefix of the interval lengths over the chunk's slots (slot = u * 256 + tid: strided passes)
      int run = 0;  // voxels of the passes before
#pragma unroll
      for (int u = 0; u < PPT; ++u) {
        int inc = len[u];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int t = __shfl_up(inc, o, 64);
          if (lane >= o) inc += t;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        const int tot = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
        c_pre[u * 256 + tid] = run + woff + inc - len[u];
        run += tot;
        __syncthreads();
      }
      if (tid == 0) c_pre[LT_PIX_CHUNK] = run;
      __syncthreads();
      const int V = run, n_slots = min(T - base, LT_PIX_CHUNK);
      const unsigned long long tm1 = VCOUNT ? (unsigned long long)wall_clock64() : 0ull;
      if (VCOUNT && tid == 0 && base == 0) atomicAdd(&dbg[5], tm1 - tm0);  // ... + first chunk's pairs and scan
      if (base == 0) LT_PIX_MARK(2);
      // the chunk's voxels, one per thread and round
      for (int jb = 0; jb < V; jb += 256) {  // (workgroup-uniform trips: the run aggregation below shuffles)
        const int j = jb + tid;
        int code = 0, col = 0, z = 0;
        if (j < V) {
        int sl = 0;  // largest slot < n_slots with c_pre[slot] <= j
#pragma unroll
        for (int st = LT_PIX_CHUNK / 2; st >= 1; st >>= 1)
          if (sl + st < n_slots && c_pre[sl + st] <= j) sl += st;
        const int sflag = c_src[sl], sidx = sflag & 63;
        col = c_col[sl]; z = c_z0[sl] + (j - c_pre[sl]);
        bool mine = true;
        if (sflag & 0x40000000) mine = z < LT_ZW_LO(zw_snap[col]) || z > LT_ZW_HI(zw_snap[n_cols_all + col]);
        col_plain Cq;
        Cq.plain = true; Cq.px = p_px[sidx]; Cq.rho2 = c_rho2[sl]; Cq.col = col;
#ifdef LT_PIX_NO_EVAL  // timing experiment: everything but the evaluation (nothing is written)
        mine = false;
#endif
        if (mine)
        code = tsdf_voxel<MERGE>(col * vol_dim_z + z, tsdf_vol, weight_vol, color_vol, rem_vol, vol_dim_x,
                                           vol_dim_y, vol_dim_z, ox, oy, oz, voxel_size, im_h, im_w, trunc_margin,
                                           obs_weight, fov_up, fov_down, sin_up_hi, sin_down_lo, color_im, depth_im,
                                           rem_im, wd_px, col_epoch, epoch, true, Cq, z, dct, p_r[sidx]);
        }
        // sign bits: the volume is fresh, every bit is 0 -- only negative values need a write; the voxels of a pair sit in
        // neighbouring lanes and (mostly) in one 64-bit word: OR them together over the run, one atomic per run
#ifdef LT_PIX_NO_BITS  // timing experiment: no sign bits (marching cubes would see nothing)
        code = 0;
#endif
        const int wkey = code == 2 ? col * words_z + (z >> 6) : -1 - lane;  // (unique when there is nothing to write)
        unsigned long long bits = code == 2 ? 1ull << (z & 63) : 0ull;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {  // runs of up to 16 lanes (a run is one column's interval: ~10 voxels)
          const unsigned long long ob = __shfl_down(bits, o, 64);
          const int ok = __shfl_down(wkey, o, 64);
          if (lane + o < 64 && ok == wkey) bits |= ob;
        }
        const int prev = __shfl_up(wkey, 1, 64);
        if (code == 2 && (lane == 0 || prev != wkey || (lane & 15) == 0))  // (lane & 15: a run longer than 16 lanes)
          atomicOr(sign_bits + (size_t)wkey, bits);
        if (VCOUNT) {  // (one atomic per wave and round: a per-voxel atomic on one address would be the whole kernel)
          const unsigned long long wr = __ballot(code != 0);
          if (lane == 0 && wr) atomicAdd(&dbg[2], (unsigned long long)__popcll(wr));
        }
      }
      if (VCOUNT && tid == 0) {
        atomicAdd(&dbg[0], (unsigned long long)n_slots); atomicAdd(&dbg[1], (unsigned long long)V);
        atomicAdd(&dbg[6], (unsigned long long)wall_clock64() - tm1);  // the voxel rounds
        atomicAdd(&dbg[7], 1ull);
      }
      __syncthreads();  // the chunk's arrays are reused
    }
    LT_PIX_MARK(3);
    // the merged ranges -> col_zw, stamps: once per column (all pairs of all chunks have merged: the barrier above)
    if (agg)
      for (int i = tid; i < agg_span; i += 256) {
        const unsigned hi1 = a_hi[i];
        if (hi1) {
          const int c = wd_ent[agg_k0 + i].x;
          const int n_cols = vol_dim_x * vol_dim_y;
          atomicMax(&col_zw[c], a_lo[i]);  // (atomic: with an image height that does not divide 64 two workgroups share a wedge)
          atomicMax(&col_zw[n_cols + c], hi1);
          col_epoch[c] = epoch;
          chunk_epoch[c >> 6] = epoch;
        }
      }
    LT_PIX_MARK(4);
    __syncthreads();  // ... and the pixels'
  }
}

// ---- several observations of a FRESH volume in ONE pass (lt_tsdf_integrate_multi_dev) --------------------------------------
// The reference's `mesh` adaption fuses `number_of_scans` range images into one new volume, all of them re-projected into
// the primary pose (laserscan.py:874-897): every observation projects a voxel into the SAME pixel, and the update of a voxel
// depends on that voxel's own state and its pixel only.  So instead of n passes -- the second one onwards with a snapshot of
// the written ranges, a pass over every voxel inside them and the pixel pass beside it (0.11 ms each) -- ONE pixel pass over
// the union of the observations' candidate intervals: a voxel's geometry (depth, pitch, row: the costly part) is evaluated
// once, then the n updates run IN ORDER on the voxel's state in registers (the expressions of tsdf_update, operation by
// operation) and the four fields are stored once.  Bit-identical to n calls of lt_tsdf_integrate_dev
// (tests/test_tsdf_gpu.py).  Class-aware branch only (the plain average writes the whole frustum: the column walk).
#define LT_TSDF_MULTI_MAX 8
struct tsdf_obs_ptrs {  // the images of the observations, by value (kernel arguments)
  const float* color[LT_TSDF_MULTI_MAX];
  const float* depth[LT_TSDF_MULTI_MAX];
  const float* rem[LT_TSDF_MULTI_MAX];
};

// (depth, colour, remission, -) of pixel (row, px) of observation k at obs4[(k * im_w + px) * im_h + row]: the transposed,
// packed copy the kernels read (one 16-B load per observation and voxel; rows of one image column contiguous)
__global__ __launch_bounds__(256) void k_tsdf_dct4(tsdf_obs_ptrs O, int n_obs, int im_h, int im_w, float4* __restrict__ obs4) {
  const int n_pix = im_h * im_w;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_pix * n_obs) return;
  const int k = i / n_pix, p = i - k * n_pix;
  const int y = p / im_w, x = p - y * im_w;
  float d = 0.f, c = 0.f, r = 0.f;
#pragma unroll
  for (int q = 0; q < LT_TSDF_MULTI_MAX; ++q)  // (the pointer table lives in scalar registers: select, do not index)
    if (q == k) { d = O.depth[q][p]; c = O.color[q][p]; r = O.rem[q][p]; }
  obs4[(size_t)k * n_pix + (size_t)x * im_h + y] = make_float4(d, c, r, 0.f);
}

// the n class-aware updates of one voxel of a FRESH volume (fusion_lidar.py:177, :191-228; tsdf_update<true> above, on
// registers): returns 0 untouched, 1 written (> 0), 2 written not > 0 (the sign bit, as tsdf_update)
__device__ __forceinline__ int tsdf_updates_fresh(float* __restrict__ tsdf_vol, float* __restrict__ weight_vol,
                                                   float* __restrict__ color_vol, float* __restrict__ rem_vol, int voxel_idx,
                                                   float depth, float trunc_margin, float obs_weight,
                                                   const float4* __restrict__ obs4, size_t pix, size_t n_pix, int n_obs) {
  float tv = 1.0f, wv = 0.0f, cv = 0.0f, rv = 0.0f;  // the initial volume (fusion_lidar.py:47-63)
  bool written = false;
  for (int k = 0; k < n_obs; ++k) {
    const float4 o = obs4[(size_t)k * n_pix + pix];
    const float depth_value = o.x, new_color = o.y, new_rem = o.z;
    if (depth_value == 0.f) continue;
    const float depth_diff = depth_value - depth;
    if (depth_diff < -trunc_margin) continue;
    const float dist = fminf(1.0f, depth_diff / trunc_margin);
    const float dist_old = wv;  // sic: the reference compares against the weight volume
    if (cv == new_color) {      // same class: integrate
      const float w_old = wv;
      const float w_new = w_old + obs_weight;
      wv = w_new;
      tv = __fmaf_rn(tv, w_old, dist) / w_new;
      rv = __fmaf_rn(rv, w_old, new_rem) / w_new;
      written = true;
    } else if (dist < dist_old) {  // other class: the closer observation wins
      tv = dist;
      const float new_b = floorf(new_color / (256 * 256));
      const float new_g = floorf((new_color - new_b * 256 * 256) / 256);
      const float new_r = new_color - new_b * 256 * 256 - new_g * 256;
      cv = new_b * 256 * 256 + new_g * 256 + new_r;
      rv = new_rem;
      written = true;
    }
  }
  if (!written) return 0;
  *LT_VOX(tsdf_vol, voxel_idx) = make_float4(tv, wv, cv, rv);
  return !(tv > 0.0f) ? 2 : 1;
}

// tsdf_voxel's geometry (the reference's expressions, :95-146), then the n updates; want_py as there
__device__ __forceinline__ int tsdf_voxel_multi(
    int voxel_idx, float* __restrict__ tsdf_vol, float* __restrict__ weight_vol, float* __restrict__ color_vol,
    float* __restrict__ rem_vol, int vol_dim_x, int vol_dim_y, int vol_dim_z, float ox, float oy, float oz,
    float voxel_size, int im_h, int im_w, float trunc_margin, float obs_weight, float fov_up, float fov_down,
    float sin_up_hi, float sin_down_lo, const int* __restrict__ colinfo, const col_plain& C, int z_plain,
    const float4* __restrict__ obs4, int n_obs, int want_py = -1) {
  int px = -2;
  float pt_x, pt_y, pt_z;
  if (C.plain) {
    px = C.px;
    col_xy(C.col, vol_dim_y, voxel_size, ox, oy, pt_x, pt_y);
    pt_z = __fmaf_rn((float)z_plain, voxel_size, oz);
  } else {
    const float voxel_x = floorf(((float)voxel_idx) / ((float)(vol_dim_y * vol_dim_z)));
    const float voxel_y = floorf(((float)(voxel_idx - ((int)voxel_x) * vol_dim_y * vol_dim_z)) / ((float)vol_dim_z));
    const float voxel_z = (float)(voxel_idx - ((int)voxel_x) * vol_dim_y * vol_dim_z - ((int)voxel_y) * vol_dim_z);
    const int ix = (int)voxel_x, iy = (int)voxel_y;
    const bool in_table = ix >= 0 && ix < vol_dim_x && iy >= 0 && iy < vol_dim_y;
    if (in_table) {
      px = colinfo[ix * vol_dim_y + iy];
      if (px == -1) return 0;
      if (px >= 0) px &= 0x3FFFFFFF;
    }
    pt_x = __fmaf_rn(voxel_x, voxel_size, ox);
    pt_y = __fmaf_rn(voxel_y, voxel_size, oy);
    pt_z = __fmaf_rn(voxel_z, voxel_size, oz);
    if (px < 0) {
      const float yaw = -atan2f(pt_y, pt_x);
      float proj_x = (float)(0.5 * ((double)yaw / LT_PI_D + 1.0));
      proj_x *= (float)im_w;
      px = (int)floorf(proj_x);
      px = min(im_w - 1, px);
      px = max(0, px);
    }
  }
  const float fov = fabsf(fov_up) + fabsf(fov_down);
  const float depth = norm3df(pt_x, pt_y, pt_z);  // the device library's, as the reference's source gets it (header note)
  const float s = pt_z / depth;
  if ((s > sin_up_hi || s < sin_down_lo) && fabsf(s) <= 1.0f) return 0;  // (|s| > 1: see tsdf_voxel)
  const float pitch = asinf(s);
  if (pitch > fov_up || pitch < fov_down) return 0;
  float proj_y = (float)(1.0 - (double)((pitch + fabsf(fov_down)) / fov));
  proj_y *= (float)im_h;
  int py = (int)floorf(proj_y);
  py = min(im_h - 1, py);
  py = max(0, py);
  if (want_py >= 0 && py != want_py) return 0;
  return tsdf_updates_fresh(tsdf_vol, weight_vol, color_vol, rem_vol, voxel_idx, depth, trunc_margin, obs_weight, obs4,
                            (size_t)px * im_h + py, (size_t)im_h * im_w, n_obs);
}

template <bool VCOUNT>
__global__ __launch_bounds__(256) void k_tsdf_integrate_pix_multi(
    float* __restrict__ tsdf_vol, float* __restrict__ weight_vol, float* __restrict__ color_vol,
    float* __restrict__ rem_vol, int vol_dim_x, int vol_dim_y, int vol_dim_z, float ox, float oy, float oz,
    float voxel_size, float inv_vs, int im_h, int im_w, float trunc_margin, float obs_weight, float fov_up,
    float fov_down, float sin_up_hi, float sin_down_lo, const float4* __restrict__ obs4, int n_obs,
    const int* __restrict__ wd_px,
    unsigned* __restrict__ col_epoch, unsigned epoch, unsigned long long* __restrict__ sign_bits, int words_z,
    unsigned* __restrict__ col_zw, unsigned* __restrict__ chunk_epoch,
    const float4* __restrict__ rowtab, const int* __restrict__ wd_start, const int2* __restrict__ wd_ent, const uint32_t* __restrict__ wd_key, int rho_bits,
    float qscale, unsigned long long* __restrict__ dbg) {
  // k_tsdf_integrate_pix for n_obs observations of a FRESH volume at once (see above): a pixel's candidate interval is the
  // UNION of the observations' -- a voxel an observation can write lies in that observation's interval while it still holds
  // its initial values, and once written (by an earlier observation: inside an earlier interval) anywhere in front of the band.
  const size_t n_pix_all = (size_t)im_h * im_w;
  // per pixel of the workgroup
  __shared__ int p_k0[64], p_pre[65], p_r[64], p_px[64];
  __shared__ float p_tlo[64], p_thi[64], p_dlo[64], p_dhi[64];
  // per pair of the chunk (one block: phase A borrows it as the staging area of the wedge's rho quanta)
  __shared__ int c_buf[LT_PIX_STAGE > 4 * LT_PIX_CHUNK + 1 ? LT_PIX_STAGE : 4 * LT_PIX_CHUNK + 1];
  int* const c_col = c_buf;
  int* const c_z0 = c_buf + LT_PIX_CHUNK;
  int* const c_src = c_buf + 2 * LT_PIX_CHUNK;
  int* const c_pre = c_buf + 3 * LT_PIX_CHUNK;  // [LT_PIX_CHUNK + 1]
  __shared__ float c_rho2[LT_PIX_CHUNK];
  __shared__ int wsum[4];
  // the columns' written ranges, merged in LDS first: the pairs of a workgroup fall on a few dozen table entries of ONE
  // wedge (a wall's column is visited by dozens of rows), and two global atomics + two stores per PAIR were 43 of the
  // kernel's 92 us (tools/pix_sections.sh).  a_lo / a_hi are indexed by table entry - a_k0 and hold the col_zw encoding
  // (0x7fff - lo, hi + 1; 0 = nothing); a workgroup whose pairs span more than LT_PIX_AGG entries marks directly.
  __shared__ unsigned a_lo[LT_PIX_AGG], a_hi[LT_PIX_AGG];
  __shared__ int a_k0, a_span;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t qmax = (1u << rho_bits) - 2u;
  const int n_pix = im_h * im_w;
  const uint32_t* const skey = (const uint32_t*)c_buf;  // staged quanta of [stage0, stage0 + n_stage)
  int stage0 = 0, n_stage = 0;
  // first table index in [a, b) whose rho quantum is >= q (b if none): the wedges are sorted by it.  The 64 pixels of a
  // workgroup search the SAME wedge(s): 22 dependent global loads per pixel became one coalesced copy into LDS
  bool staged = false;  // (workgroup-uniform: the whole search range is in LDS)
  auto lower = [&](int a, int b, uint32_t q) {
    if (staged) {
      while (a < b) {
        const int m = (a + b) >> 1;
        if (skey[m - stage0] < q) a = m + 1;
        else b = m;
      }
    } else {
      while (a < b) {
        const int m = (a + b) >> 1;
        if (wd_key[m] < q) a = m + 1;
        else b = m;
      }
    }
    return a;
  };
  for (int p0 = blockIdx.x * 64; p0 < n_pix; p0 += gridDim.x * 64) {  // (workgroup-uniform)
    const unsigned long long tm0 = VCOUNT ? (unsigned long long)wall_clock64() : 0ull;  // (100 MHz; debug)
    LT_PIX_MARK(0);
    // ---- A: the pixels' runs of table entries ------------------------------------------------------------------------
    {  // stage the quanta of the wedges these 64 pixels search (one image column when im_h is a multiple of 64)
      const int px_a = p0 / im_h, px_b = min(p0 + 63, n_pix - 1) / im_h;
      stage0 = wd_start[px_a];
      n_stage = wd_start[px_b + 1] - stage0;
      staged = n_stage <= LT_PIX_STAGE;  // (a longer run of wedges is searched in global memory)
#ifdef LT_PIX_NO_STAGE  // (timing experiment)
      staged = false;
#endif
      if (staged)
      {  // all loads first, then the LDS stores: the rolled loop waited for every load before it issued the next
        // (up to 15 dependent round trips, as in k_tsdf_integrate_pix)
        constexpr int NS = (LT_PIX_STAGE + 255) / 256;
        uint32_t sv[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) sv[k] = tid + 256 * k < n_stage ? wd_key[stage0 + tid + 256 * k] : 0u;
#pragma unroll
        for (int k = 0; k < NS; ++k)
          if (tid + 256 * k < n_stage) c_buf[tid + 256 * k] = (int)sv[k];
      }
      __syncthreads();
    }
    if (wave != 0) {
      for (int i = tid - 64; i < LT_PIX_AGG; i += 192) { a_lo[i] = 0u; a_hi[i] = 0u; }
    }
    if (wave == 0) {
      const int p = p0 + lane;  // pixel (row r, column px) at [px * im_h + r] of the transposed images
      const bool in = p < n_pix;
      const int px = in ? p / im_h : 0, r = in ? p - px * im_h : 0;
      const float4 row = rowtab[r];  // (tan_lo, tan_hi, cos_min, cos_max); tan_lo > tan_hi: no voxel can take this row
      const bool row_ok = row.x <= row.y;
      // per observation as in k_tsdf_integrate_pix; the union: [min d_lo, max d_hi]
      bool any = false;
      float d_lo = 3e38f, d_hi = 0.f;
      for (int kk = 0; kk < n_obs; ++kk) {
        const float4 o4 = in ? obs4[(size_t)kk * n_pix_all + p] : make_float4(0.f, 1.f, 0.f, 0.f);
        const float D = o4.x;
        const bool finite = D == D && fabsf(D) < 1e30f;
        const bool zero_class_k = in && row_ok && D != 0.f && o4.y == 0.0f;
        const bool normal_k = in && row_ok && D != 0.f && o4.y != 0.0f && finite;
        if (normal_k || zero_class_k) {
          const float eps = __fmaf_rn(4e-6f, fabsf(D) + trunc_margin, 1e-6f);
          const float hi_k = finite ? D + trunc_margin + eps : 3e38f;
          // (once a voxel has been written -- by an earlier observation, inside an earlier interval -- a later one updates it
          // anywhere in front of ITS band: a later observation's band further out extends the union's upper end, and the
          // voxels in front of it that earlier observations wrote are inside the union already)
          const float lo_k = zero_class_k ? 0.f : D - eps;
          if (hi_k > 0.f) { any = true; d_lo = fminf(d_lo, lo_k); d_hi = fmaxf(d_hi, hi_k); }
        }
      }
      const bool normal = any, zero_class = false;
      const int s0 = wd_start[px], s1 = wd_start[px + 1];
      int k = 0, kend = 0;
      if ((normal || zero_class) && d_hi > 0.f) {
        const float rho1 = fmaxf(d_lo, 0.f) * row.z * 0.999999f;
        const float f1 = floorf(rho1 * qscale) - 2.f;
        const uint32_t q1 = (uint32_t)fminf(fmaxf(f1, 0.f), (float)qmax);
        k = lower(s0, s1, q1);
        kend = s1;
        if (d_hi < 3e38f) {
          const float f2 = floorf(d_hi * row.w * 1.000001f * qscale) + 2.f;
          const uint32_t q2 = (uint32_t)fminf(fmaxf(f2, 0.f), (float)qmax);
          kend = lower(k, s1, q2 + 1u);
        }
      }
      const int cnt = kend - k;
      int inc = cnt;  // inclusive wave scan
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
      }
      p_k0[lane] = k; p_pre[lane] = inc - cnt; p_r[lane] = r; p_px[lane] = px;
      p_tlo[lane] = row.x; p_thi[lane] = row.y; p_dlo[lane] = d_lo; p_dhi[lane] = d_hi;
      if (lane == 63) p_pre[64] = inc;
      // the span of table entries the workgroup's pairs fall on
      int kmin = cnt > 0 ? k : 0x7fffffff, kmax = cnt > 0 ? kend : 0;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        kmin = min(kmin, __shfl_xor(kmin, o, 64));
        kmax = max(kmax, __shfl_xor(kmax, o, 64));
      }
      if (lane == 0) { a_k0 = kmin; a_span = kmax > kmin ? kmax - kmin : 0; }
    }
    __syncthreads();
    const int T = p_pre[64];
    const int agg_k0 = a_k0, agg_span = a_span;
    const bool agg = agg_span <= LT_PIX_AGG;  // (workgroup-uniform)
    if (VCOUNT && tid == 0) atomicAdd(&dbg[4], (unsigned long long)wall_clock64() - tm0);  // phase A
    LT_PIX_MARK(1);
    // ---- B: chunks of pairs ---------------------------------------------------------------------------------------------
    for (int base = 0; base < T; base += LT_PIX_CHUNK) {
      constexpr int PPT = LT_PIX_CHUNK / 256;  // pairs per thread
      int len[PPT];
#pragma unroll
      for (int u = 0; u < PPT; ++u) {
        const int slot = u * 256 + tid, i = base + slot;
        len[u] = 0;
        if (i < T) {
          int sidx = 0;  // largest s with p_pre[s] <= i
#pragma unroll
          for (int st = 32; st >= 1; st >>= 1)
            if (p_pre[sidx + st] <= i) sidx += st;
          const int kk = p_k0[sidx] + (i - p_pre[sidx]);
          const int2 e = wd_ent[kk];
          int z = 1, zend = 0;
          if (e.x >= 0) {  // (the quirk tail of the last wedge carries column -1)
            const float rho2 = __int_as_float(e.y), rho = sqrtf(rho2);
            const float d_lo = p_dlo[sidx], d_hi = p_dhi[sidx];
            const float hi2 = d_hi * d_hi - rho2;
            if (hi2 >= 0.f) {  // (else the whole column lies beyond the band; false also for NaN)
              const float zmax = sqrtf(hi2) * 1.000001f + 1e-6f;
              const float lo2 = d_lo > 0.f ? d_lo * d_lo - rho2 : -1.f;
              const float zmin = lo2 > 0.f ? fmaxf(sqrtf(lo2) * 0.999999f - 1e-6f, 0.f) : 0.f;
              const float za = rho * p_tlo[sidx], zb = rho * p_thi[sidx];  // pt_z of the row in this column
              float lo, hi;
              if (za >= 0.f) { lo = fmaxf(za, zmin); hi = fminf(zb, zmax); }
              else if (zb <= 0.f) { lo = fmaxf(za, -zmax); hi = fminf(zb, -zmin); }
              else { lo = fmaxf(za, -zmax); hi = fminf(zb, zmax); }
              if (lo <= hi) {
                const float fz0 = ceilf((lo - oz) * inv_vs - 0.02f), fz1 = floorf((hi - oz) * inv_vs + 0.02f);
                z = (int)fmaxf(fz0, 0.f);
                zend = (int)fminf(fz1, (float)(vol_dim_z - 1));
              }
            }
          }
          const int partial = 0;
          len[u] = max(zend - z + 1, 0);
          c_col[slot] = e.x; c_rho2[slot] = __int_as_float(e.y); c_z0[slot] = z; c_src[slot] = sidx | partial;
          // the column's written range and stamp once per PAIR, for the whole candidate interval (a superset is safe, as in
          // the column walk) -- per written voxel, the ten threads holding one column's band voxels fought over one word
#ifndef LT_PIX_NO_MARK  // (timing experiment)
          if (len[u] > 0) {
            if (agg) {
              atomicMax(&a_lo[kk - agg_k0], (unsigned)(0x7FFF - z));
              atomicMax(&a_hi[kk - agg_k0], (unsigned)(zend + 1));
            } else {
              col_mark_written(col_zw, col_epoch, chunk_epoch, epoch, vol_dim_x * vol_dim_y, e.x, z, zend);
            }
          }
#endif
        }
      }
      // exclusive prefix of the interval lengths over the chunk's slots (slot = u * 256 + tid: strided passes)
      int run = 0;  // voxels of the passes before
#pragma unroll
      for (int u = 0; u < PPT; ++u) {
        int inc = len[u];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int t = __shfl_up(inc, o, 64);
          if (lane >= o) inc += t;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        const int tot = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
        c_pre[u * 256 + tid] = run + woff + inc - len[u];
        run += tot;
        __syncthreads();
      }
      if (tid == 0) c_pre[LT_PIX_CHUNK] = run;
      __syncthreads();
      const int V = run, n_slots = min(T - base, LT_PIX_CHUNK);
      const unsigned long long tm1 = VCOUNT ? (unsigned long long)wall_clock64() : 0ull;
      if (VCOUNT && tid == 0 && base == 0) atomicAdd(&dbg[5], tm1 - tm0);  // ... + first chunk's pairs and scan
      if (base == 0) LT_PIX_MARK(2);
      // the chunk's voxels, one per thread and round
      for (int jb = 0; jb < V; jb += 256) {  // (workgroup-uniform trips: the run aggregation below shuffles)
        const int j = jb + tid;
        int code = 0, col = 0, z = 0;
        if (j < V) {
        int sl = 0;  // largest slot < n_slots with c_pre[slot] <= j
#pragma unroll
        for (int st = LT_PIX_CHUNK / 2; st >= 1; st >>= 1)
          if (sl + st < n_slots && c_pre[sl + st] <= j) sl += st;
        const int sflag = c_src[sl], sidx = sflag & 63;
        col = c_col[sl]; z = c_z0[sl] + (j - c_pre[sl]);
        bool mine = true;
        col_plain Cq;
        Cq.plain = true; Cq.px = p_px[sidx]; Cq.rho2 = c_rho2[sl]; Cq.col = col;
#ifdef LT_PIX_NO_EVAL  // timing experiment: everything but the evaluation (nothing is written)
        mine = false;
#endif
        if (mine)
        code = tsdf_voxel_multi(col * vol_dim_z + z, tsdf_vol, weight_vol, color_vol, rem_vol, vol_dim_x, vol_dim_y, vol_dim_z,
                                ox, oy, oz, voxel_size, im_h, im_w, trunc_margin, obs_weight, fov_up, fov_down, sin_up_hi,
                                sin_down_lo, wd_px, Cq, z, obs4, n_obs, p_r[sidx]);
        }
        // sign bits: the volume is fresh, every bit is 0 -- only negative values need a write; the voxels of a pair sit in
        // neighbouring lanes and (mostly) in one 64-bit word: OR them together over the run, one atomic per run
#ifdef LT_PIX_NO_BITS  // timing experiment: no sign bits (marching cubes would see nothing)
        code = 0;
#endif
        const int wkey = code == 2 ? col * words_z + (z >> 6) : -1 - lane;  // (unique when there is nothing to write)
        unsigned long long bits = code == 2 ? 1ull << (z & 63) : 0ull;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {  // runs of up to 16 lanes (a run is one column's interval: ~10 voxels)
          const unsigned long long ob = __shfl_down(bits, o, 64);
          const int ok = __shfl_down(wkey, o, 64);
          if (lane + o < 64 && ok == wkey) bits |= ob;
        }
        const int prev = __shfl_up(wkey, 1, 64);
        if (code == 2 && (lane == 0 || prev != wkey || (lane & 15) == 0))  // (lane & 15: a run longer than 16 lanes)
          atomicOr(sign_bits + (size_t)wkey, bits);
        if (VCOUNT) {  // (one atomic per wave and round: a per-voxel atomic on one address would be the whole kernel)
          const unsigned long long wr = __ballot(code != 0);
          if (lane == 0 && wr) atomicAdd(&dbg[2], (unsigned long long)__popcll(wr));
        }
      }
      if (VCOUNT && tid == 0) {
        atomicAdd(&dbg[0], (unsigned long long)n_slots); atomicAdd(&dbg[1], (unsigned long long)V);
        atomicAdd(&dbg[6], (unsigned long long)wall_clock64() - tm1);  // the voxel rounds
        atomicAdd(&dbg[7], 1ull);
      }
      __syncthreads();  // the chunk's arrays are reused
    }
    LT_PIX_MARK(3);
    // the merged ranges -> col_zw, stamps: once per column (all pairs of all chunks have merged: the barrier above)
    if (agg)
      for (int i = tid; i < agg_span; i += 256) {
        const unsigned hi1 = a_hi[i];
        if (hi1) {
          const int c = wd_ent[agg_k0 + i].x;
          const int n_cols = vol_dim_x * vol_dim_y;
          atomicMax(&col_zw[c], a_lo[i]);  // (atomic: with an image height that does not divide 64 two workgroups share a wedge)
          atomicMax(&col_zw[n_cols + c], hi1);
          col_epoch[c] = epoch;
          chunk_epoch[c >> 6] = epoch;
        }
      }
    LT_PIX_MARK(4);
    __syncthreads();  // ... and the pixels'
  }
}

// the quirk columns (not in the wedge table) for the n observations at once, one thread per voxel
__global__ __launch_bounds__(256) void k_tsdf_integrate_quirk_multi(
    float* __restrict__ tsdf_vol, float* __restrict__ weight_vol, float* __restrict__ color_vol,
    float* __restrict__ rem_vol, int vol_dim_x, int vol_dim_y, int vol_dim_z, float ox, float oy, float oz,
    float voxel_size, int im_h, int im_w, float trunc_margin, float obs_weight, float fov_up, float fov_down,
    float sin_up_hi, float sin_down_lo, const float4* __restrict__ obs4, int n_obs, const int* __restrict__ wd_px,
    unsigned* __restrict__ col_epoch, unsigned epoch, unsigned long long* __restrict__ sign_bits, int words_z,
    unsigned* __restrict__ col_zw, unsigned* __restrict__ chunk_epoch, const uint32_t* __restrict__ qcols, int n_q) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)n_q * vol_dim_z) return;
  const int col = (int)qcols[i / vol_dim_z], z = (int)(i % vol_dim_z);
  col_plain C;
  C.plain = false; C.px = -2; C.rho2 = 0.f; C.col = 0;
  const int code = tsdf_voxel_multi(col * vol_dim_z + z, tsdf_vol, weight_vol, color_vol, rem_vol, vol_dim_x, vol_dim_y,
                                    vol_dim_z, ox, oy, oz, voxel_size, im_h, im_w, trunc_margin, obs_weight, fov_up, fov_down,
                                    sin_up_hi, sin_down_lo, wd_px, C, z, obs4, n_obs);
  if (code) {
    if (code == 2) atomicOr(sign_bits + (size_t)col * words_z + (z >> 6), 1ull << (z & 63));  // (fresh volume: the bit is 0)
    col_mark_written(col_zw, col_epoch, chunk_epoch, epoch, vol_dim_x * vol_dim_y, col, z, z);
  }
}

// A volume that already holds observations: every voxel inside a column's written range [lo, hi] -- as it stood before this
// observation (zw_snap) -- runs the reference's expressions with its stored values loaded (`fresh` = false): a voxel written
// earlier can be written again anywhere in front of the band (same class: the running average; another class: the closer
// observation wins, fusion_lidar.py:191-213), which the pixel pass, built for voxels in their initial state, does not cover.
// A workgroup per chunk of 64 columns; the chunk's voxels -- 64 ranges of ~10 -- are dealt one per thread and round
// (prefix sum of the range lengths in LDS), the signs of what was written go to the columns' sign words as one OR and
// one AND-NOT per run of lanes holding one column's word.  (First version: a wave per chunk, four columns per iteration
// with 16 lanes each -- a wall's chunk kept its wave for 16 rounds, 138 us; a wave per 16 columns: 86 us.)
// debug (-DLT_TSDF_STAMP; tools/tsdf_written_times.py): per workgroup of k_tsdf_integrate_written -- wall clock (100 MHz) at its
// start and end, time spent in the chunks' prologues (ranges -> LDS) and in their voxel rounds, written chunks walked
#ifdef LT_TSDF_STAMP
__device__ unsigned long long g_tsdf_stamp[5 << 12];
extern "C" int lt_debug_tsdf_stamps(unsigned long long* out, int n_wgs) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tsdf_stamp), (size_t)min(n_wgs, 1 << 12) * 5 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
#ifndef LT_WRITTEN_CHUNKS_PER_WG
#define LT_WRITTEN_CHUNKS_PER_WG 8  // (swept 4 .. 32 on the default volume: 62 / 58 / 70 / 71 us)
#endif
template <bool MERGE>
__global__ __launch_bounds__(256) void k_tsdf_integrate_written(
    float* __restrict__ tsdf_vol, float* __restrict__ weight_vol, float* __restrict__ color_vol,
    float* __restrict__ rem_vol, int vol_dim_x, int vol_dim_y, int vol_dim_z, float ox, float oy, float oz,
    float voxel_size, int im_h, int im_w, float trunc_margin, float obs_weight, float fov_up, float fov_down,
    float sin_up_hi, float sin_down_lo, const float* __restrict__ color_im, const float* __restrict__ depth_im,
    const float* __restrict__ rem_im, const int* __restrict__ wd_px, unsigned* __restrict__ col_epoch, unsigned epoch,
    unsigned long long* __restrict__ sign_bits, int words_z, const unsigned* __restrict__ zw_snap,
    const float2* __restrict__ dct, const unsigned* __restrict__ chunk_epoch) {
  __shared__ int w_lo[64], w_px[64], w_pre[65];
  __shared__ float w_rho2[64];
  __shared__ unsigned long long w_live;
  const int tid = threadIdx.x, lane = tid & 63;
  const int n_cols = vol_dim_x * vol_dim_y, n_chunks = (n_cols + 63) / 64;
  // a workgroup takes LT_WRITTEN_CHUNKS_PER_WG chunks, a stride apart (lt_deal_count); its first lanes read one stamp each
  // (a workgroup per chunk was 250 000 waves on the default volume, most of which read a stamp and left: 72 -> 58-64 us.
  // Per-workgroup stamps, tools/tsdf_written_times.py: a written chunk holds ~114 voxels in its columns' ranges and costs
  // 1.4 us of prologue + 3.3 us of voxel rounds; the launch is its slowest workgroups, not its work)
#ifdef LT_TSDF_STAMP
  const unsigned long long st_t0 = wall_clock64();
  unsigned long long st_pro = 0, st_ev = 0, st_n = 0;
#endif
  if (tid < 64) {
    const int ch = lane * (int)gridDim.x + (int)blockIdx.x;
    const unsigned long long m =
        __ballot(lane < LT_WRITTEN_CHUNKS_PER_WG && ch < n_chunks && chunk_epoch[min(ch, n_chunks - 1)] == epoch);
    if (lane == 0) w_live = m;
  }
  __syncthreads();
  for (unsigned long long live = w_live; live; live &= live - 1) {  // (workgroup-uniform)
    const int chunk = (__ffsll((long long)live) - 1) * (int)gridDim.x + (int)blockIdx.x;
#ifdef LT_TSDF_STAMP
    const unsigned long long st_a = wall_clock64();
#endif
    if (tid < 64) {
      const int c = chunk * 64 + lane;
      int lo = 0, len = 0, px = 0;
      float rho2 = 0.f;
      if (c < n_cols) {
        const int hi = LT_ZW_HI(zw_snap[n_cols + c]);
        if (hi >= 0) {
          const int pxq = wd_px[c];
          if (!(pxq & LT_WD_QUIRK_FLAG)) {  // (k_tsdf_integrate_quirk evaluates those columns whole)
            lo = LT_ZW_LO(zw_snap[c]);
            len = max(min(hi, vol_dim_z - 1) - lo + 1, 0);
            px = pxq;
            const int cx = c / vol_dim_y, cy = c - cx * vol_dim_y;
            const float pt_x = __fmaf_rn((float)cx, voxel_size, ox), pt_y = __fmaf_rn((float)cy, voxel_size, oy);
            rho2 = __fmaf_rn(pt_y, pt_y, pt_x * pt_x);
          }
        }
      }
      int inc = len;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
      }
      w_lo[lane] = lo; w_px[lane] = px; w_rho2[lane] = rho2; w_pre[lane] = inc - len;
      if (lane == 63) w_pre[64] = inc;
    }
    __syncthreads();
#ifdef LT_TSDF_STAMP
    const unsigned long long st_b = wall_clock64();
    st_pro += st_b - st_a; st_n += 1;
#endif
    const int V = w_pre[64];
    for (int jb = 0; jb < V; jb += 256) {  // (workgroup-uniform trips: the run aggregation below shuffles)
      const int j = jb + tid;
      int code = 0, cc = 0, z = 0;
      if (j < V) {
        int sidx = 0;  // largest column with w_pre <= j
#pragma unroll
        for (int st = 32; st >= 1; st >>= 1)
          if (w_pre[sidx + st] <= j) sidx += st;
        cc = chunk * 64 + sidx;
        z = w_lo[sidx] + (j - w_pre[sidx]);
        col_plain C;
        C.plain = true; C.px = w_px[sidx]; C.rho2 = w_rho2[sidx]; C.col = cc;
        code = tsdf_voxel<MERGE>(cc * vol_dim_z + z, tsdf_vol, weight_vol, color_vol, rem_vol, vol_dim_x, vol_dim_y, vol_dim_z,
                                 ox, oy, oz, voxel_size, im_h, im_w, trunc_margin, obs_weight, fov_up, fov_down, sin_up_hi,
                                 sin_down_lo, color_im, depth_im, rem_im, wd_px, col_epoch, epoch, false, C, z, dct);
      }
      // the sign of every value written: set the bit (negative) or clear it; one OR and one AND-NOT per run of lanes that
      // hold voxels of one column's word (a column's range is consecutive in j)
      const int wkey = code ? cc * words_z + (z >> 6) : -1 - lane;
      unsigned long long set = code == 2 ? 1ull << (z & 63) : 0ull, clr = code == 1 ? 1ull << (z & 63) : 0ull;
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        const unsigned long long os = __shfl_down(set, o, 64), oc = __shfl_down(clr, o, 64);
        const int ok = __shfl_down(wkey, o, 64);
        if (lane + o < 64 && ok == wkey) { set |= os; clr |= oc; }
      }
      const int prev = __shfl_up(wkey, 1, 64);
      if (code && (lane == 0 || prev != wkey || (lane & 15) == 0)) {
        if (set) atomicOr(sign_bits + (size_t)wkey, set);
        if (clr) atomicAnd(sign_bits + (size_t)wkey, ~clr);
      }
    }
    __syncthreads();  // the chunk's arrays are reused
#ifdef LT_TSDF_STAMP
    st_ev += wall_clock64() - st_b;
    st_n += (unsigned long long)V << 16;
#endif
  }
#ifdef LT_TSDF_STAMP
  if (tid == 0 && blockIdx.x < (1 << 12)) {
    unsigned long long* o = g_tsdf_stamp + 5 * blockIdx.x;
    o[0] = st_t0; o[1] = wall_clock64(); o[2] = st_pro; o[3] = st_ev; o[4] = st_n;
  }
#endif
}

// the columns that are not in the wedge table (k_wd_keys: quirk), one thread per voxel, by the reference's own
// decomposition of the voxel index (tsdf_voxel's general path; a decomposed (x, y) inside the table finds its image
// column in wd_px, one outside -- the "(x + 1, -1)" voxels -- computes it)
template <bool MERGE>
__global__ __launch_bounds__(256) void k_tsdf_integrate_quirk(
    float* __restrict__ tsdf_vol, float* __restrict__ weight_vol, float* __restrict__ color_vol,
    float* __restrict__ rem_vol, int vol_dim_x, int vol_dim_y, int vol_dim_z, float ox, float oy, float oz,
    float voxel_size, int im_h, int im_w, float trunc_margin, float obs_weight, float fov_up, float fov_down,
    float sin_up_hi, float sin_down_lo, const float* __restrict__ color_im, const float* __restrict__ depth_im,
    const float* __restrict__ rem_im, const int* __restrict__ wd_px, unsigned* __restrict__ col_epoch, unsigned epoch,
    unsigned long long* __restrict__ sign_bits, int words_z, unsigned* __restrict__ col_zw,
    unsigned* __restrict__ chunk_epoch, const float2* __restrict__ dct, const uint32_t* __restrict__ qcols, int n_q,
    int fresh) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)n_q * vol_dim_z) return;
  const int col = (int)qcols[i / vol_dim_z], z = (int)(i % vol_dim_z);
  col_plain C;
  C.plain = false; C.px = -2; C.rho2 = 0.f; C.col = 0;
  const int code = tsdf_voxel<MERGE>(col * vol_dim_z + z, tsdf_vol, weight_vol, color_vol, rem_vol, vol_dim_x, vol_dim_y,
                                     vol_dim_z, ox, oy, oz, voxel_size, im_h, im_w, trunc_margin, obs_weight, fov_up,
                                     fov_down, sin_up_hi, sin_down_lo, color_im, depth_im, rem_im, wd_px, col_epoch, epoch,
                                     fresh != 0, C, z, dct);
  if (code) {
    unsigned long long* w = sign_bits + (size_t)col * words_z + (z >> 6);
    const unsigned long long bit = 1ull << (z & 63);
    if (code == 2) atomicOr(w, bit);
    else atomicAnd(w, ~bit);
    col_mark_written(col_zw, col_epoch, chunk_epoch, epoch, vol_dim_x * vol_dim_y, col, z, z);
  }
}

extern "C" int lt_tsdf_destroy(lt_tsdf* t) {
  if (!t) return LT_OK;
  (void)hipSetDevice(t->device);
  (void)hipDeviceSynchronize();
  void* ps[] = {t->tsdf, t->col_epoch, t->colinfo, t->colmax, t->bits, t->col_zw, t->dct,
                t->wd_px, t->wd_start, t->wd_ent, t->wd_key, t->wd_qcols, t->rowtab, t->zw_snap, t->chunk_epoch, t->obs4};
  for (void* p : ps)
    if (p) (void)hipFree(p);
  free(t);
  return LT_OK;
}

// geometry of the per-column z range: slopes of the field of view with the 1e-5 margin of the sine test on the angles;
// off for fields of view beyond +-80 degrees
static unsigned long long* g_tsdf_dbg = nullptr;  // LIDARHIP_DEBUG_TSDF: per-wave stamps of k_tsdf_integrate_cols
static unsigned long long* g_pix_dbg = nullptr;   // ... and {pairs, candidate voxels, written voxels} of k_tsdf_integrate_pix

static col_geom tsdf_geom(const lt_tsdf* t) {
  const float fu = (float)((double)(float)t->fov_up_deg * LT_PI_D / 180.0);
  const float fd = (float)((double)(float)t->fov_down_deg * LT_PI_D / 180.0);
  col_geom G;
  G.dim_y = t->dim[1]; G.dim_z = t->dim[2];
  G.ox = t->origin[0]; G.oy = t->origin[1]; G.oz = t->origin[2];
  G.voxel_size = t->voxel_size;
  G.tan_ok = fabs((double)fu) < 1.39 && fabs((double)fd) < 1.39;
  G.tan_up = G.tan_ok ? (float)tan((double)fu + 1e-5) : 0.f;
  G.tan_down = G.tan_ok ? (float)tan((double)fd - 1e-5) : 0.f;
  return G;
}

static int tsdf_full_reset(lt_tsdf* t, hipStream_t stream) {
  hipLaunchKernelGGL(k_tsdf_fill, dim3(4096), dim3(256), 0, stream, reinterpret_cast<float4*>(t->tsdf), t->n);
  LT_HIP(hipMemsetAsync(t->col_epoch, 0, (size_t)t->dim[0] * t->dim[1] * sizeof(unsigned), stream));
  LT_HIP(hipMemsetAsync(t->bits, 0, (size_t)t->dim[0] * t->dim[1] * ((t->dim[2] + 63) / 64) * sizeof(unsigned long long),
                        stream));
  LT_HIP(hipMemsetAsync(t->col_zw, 0, (size_t)2 * t->dim[0] * t->dim[1] * sizeof(unsigned), stream));
  LT_HIP(hipMemsetAsync(t->chunk_epoch, 0, (((size_t)t->dim[0] * t->dim[1] + 63) / 64) * sizeof(unsigned), stream));
  LT_HIP(hipGetLastError());
  t->epoch = 1;
  t->all_dirty = 0;
  t->n_obs = 0;
  return LT_OK;
}

// Back to the initial volume (tsdf = 1, weight = colour = remission = 0): the reference builds a NEW TSDFVolume per
// output scan (laserscan.py:886-887, :968-969).  Only the columns written since the last reset are touched.
extern "C" int lt_tsdf_reset(lt_tsdf* t, void* stream) {
  if (!t) {
    lt_set_error("lt_tsdf_reset: NULL volume");
    return LT_ERR_INVALID_ARG;
  }
  LT_HIP(hipSetDevice(t->device));
  if (t->all_dirty || t->epoch == 0xFFFFFFFFu) return tsdf_full_reset(t, (hipStream_t)stream);
  const int n_cols = t->dim[0] * t->dim[1];
  const int n_chunks = (n_cols + 63) / 64;
  hipLaunchKernelGGL(k_tsdf_reset_cols, dim3((unsigned)(lt_deal_count(n_chunks, LT_RESET_CHUNKS_PER_WAVE, t->dim[1], 4) / 4)),
                     dim3(256), 0, (hipStream_t)stream,
                     t->tsdf, t->weight, t->color, t->rem, n_cols, t->dim[2], t->col_epoch, t->epoch, t->bits, t->col_zw,
                     t->chunk_epoch);
  LT_HIP(hipGetLastError());
  t->epoch += 1;  // every stamp is stale now: nothing to clear
  t->n_obs = 0;
  return LT_OK;
}

extern "C" int lt_tsdf_create(lt_tsdf** out, const double* vol_bnds, double voxel_size, double fov_up,
                              double fov_down, int device) {
  if (!out || !vol_bnds || !(voxel_size > 0)) {
    lt_set_error("lt_tsdf_create: invalid argument");
    return LT_ERR_INVALID_ARG;
  }
  *out = nullptr;
  if (device < 0) LT_HIP(hipGetDevice(&device));
  LT_HIP(hipSetDevice(device));
  lt_tsdf* t = (lt_tsdf*)calloc(1, sizeof(lt_tsdf));
  if (!t) return LT_ERR_NO_MEMORY;
  t->device = device;
  memcpy(t->bnds_given, vol_bnds, sizeof(t->bnds_given));
  double n = 1;
  for (int k = 0; k < 3; ++k) {  // fusion_lidar.py:33-36
    t->dim[k] = (int)ceil((vol_bnds[2 * k + 1] - vol_bnds[2 * k]) / voxel_size);
    t->origin[k] = (float)vol_bnds[2 * k];
    if (t->dim[k] <= 0) {
      lt_set_error("lt_tsdf_create: empty volume");
      free(t);
      return LT_ERR_INVALID_ARG;
    }
    n *= t->dim[k];
  }
  if (n >= 2147483647.0) {
    lt_set_error("lt_tsdf_create: %.0f voxels exceed the int32 voxel index of the reference kernel", n);
    free(t);
    return LT_ERR_TOO_LARGE;
  }
  if (t->dim[2] > 32767) {  // (the per-column z ranges are packed into 15 / 16 bits)
    lt_set_error("lt_tsdf_create: %d voxels along z exceed 32767", t->dim[2]);
    free(t);
    return LT_ERR_TOO_LARGE;
  }
  t->n = (size_t)n;
  t->voxel_size = (float)voxel_size;
  t->voxel_size_d = voxel_size;
  t->trunc_margin = (float)(voxel_size * 5);  // fusion_lidar.py:31
  t->fov_up_deg = fov_up;
  t->fov_down_deg = fov_down;
  if (hipMalloc((void**)&t->tsdf, t->n * sizeof(float4)) != hipSuccess) {  // one float4 per voxel (LT_VOX)
    lt_set_error("lt_tsdf_create: hipMalloc of %zu bytes failed", t->n * sizeof(float4));
    t->tsdf = nullptr;
    lt_tsdf_destroy(t);
    return LT_ERR_NO_MEMORY;
  }
  t->weight = t->tsdf + 1; t->color = t->tsdf + 2; t->rem = t->tsdf + 3;  // (the fields of voxel 0; stride 4)
  const size_t n_cols = (size_t)t->dim[0] * t->dim[1];
  if (hipMalloc((void**)&t->col_epoch, n_cols * sizeof(unsigned)) != hipSuccess ||
      hipMalloc((void**)&t->colinfo, (3 * n_cols + (n_cols + 63) / 64 + 64) * sizeof(int)) != hipSuccess ||  // + one flag per chunk, colz, colrho2
      hipMalloc((void**)&t->col_zw, 2 * n_cols * sizeof(unsigned)) != hipSuccess ||
      hipMalloc((void**)&t->chunk_epoch, ((n_cols + 63) / 64) * sizeof(unsigned)) != hipSuccess ||
      hipMalloc((void**)&t->bits, n_cols * ((t->dim[2] + 63) / 64) * sizeof(unsigned long long)) != hipSuccess) {
    lt_set_error("lt_tsdf_create: hipMalloc of the column tables failed");
    lt_tsdf_destroy(t);
    return LT_ERR_NO_MEMORY;
  }
  const int rc = tsdf_full_reset(t, nullptr);
  if (rc != LT_OK) {
    lt_tsdf_destroy(t);
    return rc;
  }
  LT_HIP(hipDeviceSynchronize());
  *out = t;
  return LT_OK;
}

// sort kernels of lt_build.hip
void lt_sort_pairs(uint32_t* keys[2], uint32_t* vals[2], uint32_t* hist, int n, int key_bits, hipStream_t stream,
                   int* out_buffer);

// the wedge table of (volume geometry, image width): built on first use, once (a few hundred microseconds + temporary
// sort buffers); rebuilt when an integrate comes with another image width
static int tsdf_wedge_build(lt_tsdf* t, int im_w, int rho_bits, hipStream_t stream) {
  const int n = t->dim[0] * t->dim[1];
  if (!t->wd_px || !t->wd_ent || !t->wd_key) {
    // all three or none: an allocation that fails half way must not leave a non-NULL wd_px behind (the next integrate
    // would skip this block and launch the table kernels on NULL tables)
    if (t->wd_px || t->wd_ent || t->wd_key) LT_HIP(hipStreamSynchronize(stream));
    if (t->wd_px) (void)hipFree(t->wd_px);
    if (t->wd_ent) (void)hipFree(t->wd_ent);
    if (t->wd_key) (void)hipFree(t->wd_key);
    t->wd_px = nullptr; t->wd_ent = nullptr; t->wd_key = nullptr;
    t->wd_w = 0;
    bool ok = hipMalloc((void**)&t->wd_px, (size_t)n * sizeof(int)) == hipSuccess &&
              hipMalloc((void**)&t->wd_ent, (size_t)n * sizeof(int2)) == hipSuccess &&
              hipMalloc((void**)&t->wd_key, (size_t)n * sizeof(unsigned)) == hipSuccess;
    if (!ok) {
      if (t->wd_px) (void)hipFree(t->wd_px);
      if (t->wd_ent) (void)hipFree(t->wd_ent);
      if (t->wd_key) (void)hipFree(t->wd_key);
      t->wd_px = nullptr; t->wd_ent = nullptr; t->wd_key = nullptr;
      (void)hipGetLastError();
      lt_set_error("lt_tsdf: out of device memory for the wedge table (%d columns)", n);
      return LT_ERR_NO_MEMORY;
    }
  }
  if (t->wd_start) { LT_HIP(hipStreamSynchronize(stream)); (void)hipFree(t->wd_start); t->wd_start = nullptr; }
  if (t->wd_qcols) { (void)hipFree(t->wd_qcols); t->wd_qcols = nullptr; }
  t->wd_w = 0;
  LT_HIP(hipMalloc((void**)&t->wd_start, (size_t)(im_w + 2) * sizeof(int)));
  // largest horizontal distance of a voxel column from the sensor (the world origin: the kernel's pt_x, pt_y)
  double rmax = 0.0;
  for (int ix = 0; ix < 2; ++ix)
    for (int iy = 0; iy < 2; ++iy) {
      const double x = (double)t->origin[0] + (ix ? t->dim[0] : 0) * (double)t->voxel_size;
      const double y = (double)t->origin[1] + (iy ? t->dim[1] : 0) * (double)t->voxel_size;
      rmax = fmax(rmax, sqrt(x * x + y * y));
    }
  wd_geom G;
  G.dim_x = t->dim[0]; G.dim_y = t->dim[1]; G.dim_z = t->dim[2];
  G.ox = t->origin[0]; G.oy = t->origin[1]; G.oz = t->origin[2]; G.vs = t->voxel_size;
  G.im_w = im_w; G.rho_bits = rho_bits;
  G.qscale = (float)(((double)((1u << rho_bits) - 2u)) / fmax(rmax * 1.001, 1e-6));
  uint32_t* keys[2] = {nullptr, nullptr};
  uint32_t* vals[2] = {nullptr, nullptr};
  uint32_t* hist = nullptr;
  int* n_quirk = nullptr;
  const int nb = (n + LT_SORT_TILE - 1) / LT_SORT_TILE;
  int rc = LT_OK;
  auto release = [&]() {
    for (int k = 0; k < 2; ++k) { if (keys[k]) (void)hipFree(keys[k]); if (vals[k]) (void)hipFree(vals[k]); }
    if (hist) (void)hipFree(hist);
    if (n_quirk) (void)hipFree(n_quirk);
  };
  for (int k = 0; k < 2 && rc == LT_OK; ++k)
    if (hipMalloc((void**)&keys[k], (size_t)n * 4) != hipSuccess || hipMalloc((void**)&vals[k], (size_t)n * 4) != hipSuccess)
      rc = LT_ERR_NO_MEMORY;
  if (rc == LT_OK && (hipMalloc((void**)&hist, ((size_t)1024 * nb + 1024) * 4) != hipSuccess ||
                      hipMalloc((void**)&n_quirk, sizeof(int)) != hipSuccess))
    rc = LT_ERR_NO_MEMORY;
  if (rc != LT_OK) {
    (void)hipGetLastError();
    release();
    lt_set_error("lt_tsdf_integrate_dev: hipMalloc of the wedge-table sort buffers failed");
    return rc;
  }
  int nq = 0, buf = 0;
  bool ok = hipMemsetAsync(n_quirk, 0, sizeof(int), stream) == hipSuccess;
  hipLaunchKernelGGL(k_wd_keys, dim3((n + 255) / 256), dim3(256), 0, stream, G, keys[0], vals[0], t->wd_px, n_quirk);
  lt_sort_pairs(keys, vals, hist, n, 30, stream, &buf);
  hipLaunchKernelGGL(k_wd_finish, dim3((n + 1 + 255) / 256), dim3(256), 0, stream, G, keys[buf], vals[buf], n, t->wd_ent,
                     t->wd_key, t->wd_start);
  ok = ok && hipGetLastError() == hipSuccess && hipStreamSynchronize(stream) == hipSuccess &&
       hipMemcpy(&nq, n_quirk, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
  if (ok && nq > 0) {  // the quirk columns are the tail of the sorted order
    ok = hipMalloc((void**)&t->wd_qcols, (size_t)nq * 4) == hipSuccess &&
         hipMemcpy(t->wd_qcols, vals[buf] + (n - nq), (size_t)nq * 4, hipMemcpyDeviceToDevice) == hipSuccess;
  }
  release();
  if (!ok) {
    lt_set_error("lt_tsdf_integrate_dev: building the wedge table failed: %s", hipGetErrorString(hipGetLastError()));
    return LT_ERR_HIP;
  }
  t->wd_n_quirk = nq;
  t->wd_rho_bits = rho_bits;
  t->wd_qscale = G.qscale;
  t->wd_w = im_w;
  return LT_OK;
}

// per image row r: the pitch range whose voxels the reference's projection (:143-146) puts into row r, cut to the field of
// view, +- 3e-5 rad (asinf, the float row arithmetic: < 1e-6) -> (tan_lo, tan_hi, cos_min, cos_max), rounded outwards
static int tsdf_rowtab(lt_tsdf* t, int im_h, float fu, float fd, hipStream_t stream) {
  std::vector<float4> tab((size_t)im_h);
  const double fov = (double)(fabsf(fu) + fabsf(fd)), afd = (double)fabsf(fd), m = 3e-5;
  for (int r = 0; r < im_h; ++r) {
    double hi = r == 0 ? 1.5 : fov * (1.0 - (double)r / im_h) - afd;             // (row 0 also takes proj_y < 0: clamped)
    double lo = r == im_h - 1 ? -1.5 : fov * (1.0 - (double)(r + 1) / im_h) - afd;  // (the last row proj_y >= im_h)
    hi = fmin(hi, (double)fu) + m;   // pitch > fov_up and pitch < fov_down leave the kernel (:128)
    lo = fmax(lo, (double)fd) - m;
    if (lo > hi) { tab[r] = make_float4(1.f, -1.f, 0.f, 0.f); continue; }
    const double ca = cos(lo), cb = cos(hi);
    const double cmax = (lo <= 0.0 && hi >= 0.0) ? 1.0 : fmax(ca, cb), cmin = fmax(fmin(ca, cb), 0.0);
    tab[r] = make_float4(nextafterf((float)tan(lo), -INFINITY), nextafterf((float)tan(hi), INFINITY),
                         nextafterf((float)cmin, 0.f), nextafterf((float)cmax, 2.f));
  }
  if (im_h > t->rowtab_h) {
    if (t->rowtab) { LT_HIP(hipStreamSynchronize(stream)); (void)hipFree(t->rowtab); t->rowtab = nullptr; }
    LT_HIP(hipMalloc((void**)&t->rowtab, (size_t)im_h * sizeof(float4)));
    t->rowtab_h = im_h;
  }
  // (a blocking copy from pageable memory: the vector may go out of scope when this returns)
  LT_HIP(hipMemcpyAsync(t->rowtab, tab.data(), (size_t)im_h * sizeof(float4), hipMemcpyHostToDevice, stream));
  LT_HIP(hipStreamSynchronize(stream));
  return LT_OK;
}

static int tsdf_integrate_pix(lt_tsdf* t, const float* color_im, const float* depth_im, const float* rem_im, int im_h,
                              int im_w, float obs_weight, float fu, float fd, int rho_bits, bool fresh_volume,
                              hipStream_t stream) {
  if (t->wd_w != im_w || t->wd_rho_bits != rho_bits) LT_CHECK(tsdf_wedge_build(t, im_w, rho_bits, stream));
  // the row table depends on (im_h, fov): both fixed for a sensor model; keyed by im_h and re-made when it changes
  if (t->rowtab_for_h != im_h || !t->rowtab) {
    t->rowtab_for_h = 0;
    LT_CHECK(tsdf_rowtab(t, im_h, fu, fd, stream));
    t->rowtab_for_h = im_h;
  }
  if ((size_t)im_w * im_h > t->cap_dct) {
    if (t->dct) { LT_HIP(hipDeviceSynchronize()); (void)hipFree(t->dct); t->dct = nullptr; t->cap_dct = 0; }
    LT_HIP(hipMalloc((void**)&t->dct, (size_t)im_w * im_h * sizeof(float2)));
    t->cap_dct = (size_t)im_w * im_h;
  }
  hipLaunchKernelGGL(k_tsdf_dct, dim3((im_h * im_w + 255) / 256), dim3(256), 0, stream, depth_im, color_im, im_h, im_w, t->dct);
  const float su = (float)(sin((double)fu) + 1e-5), sd = (float)(sin((double)fd) - 1e-5);
  const int words_z = (t->dim[2] + 63) / 64;
  const int n_pix = im_h * im_w;
  // a workgroup per 64 pixels -- unless that is more than the chip holds at once (5 per CU: 97 VGPRs, 27.5 KB of LDS): the
  // workgroups then walk the groups of 64 in turn, so that the launch is ONE round of resident workgroups
  // (LIDARHIP_PIX_WGS=n: n workgroups; =0: one per 64 pixels)
  static const int env_wgs = []() { const char* e = getenv("LIDARHIP_PIX_WGS"); return e ? atoi(e) : -1; }();
  const int res_wgs = lt_cu_count(t->device) * LT_PIX_WPE;
  const int groups = (n_pix + 63) / 64;
  const unsigned nb = (unsigned)min(groups, env_wgs > 0 ? env_wgs : (env_wgs == 0 ? (1 << 20) : res_wgs));
  const unsigned* zw_snap = nullptr;
  if (!fresh_volume) {
    // the written ranges as they stand before this observation (32 MB on the default volume: a device copy), then every
    // voxel inside them by the reference's expressions; the pixel pass below takes the rest
    const size_t n_cols = (size_t)t->dim[0] * t->dim[1];
    if (!t->zw_snap) LT_HIP(hipMalloc((void**)&t->zw_snap, 2 * n_cols * sizeof(unsigned)));
    LT_HIP(hipMemcpyAsync(t->zw_snap, t->col_zw, 2 * n_cols * sizeof(unsigned), hipMemcpyDeviceToDevice, stream));
    zw_snap = t->zw_snap;
    hipLaunchKernelGGL(k_tsdf_integrate_written<true>,
                       dim3((unsigned)lt_deal_count((int)((n_cols + 63) / 64), LT_WRITTEN_CHUNKS_PER_WG, t->dim[1], 1)), dim3(256), 0,
                       stream, t->tsdf, t->weight, t->color, t->rem, t->dim[0], t->dim[1], t->dim[2], t->origin[0], t->origin[1],
                       t->origin[2], t->voxel_size, im_h, im_w, t->trunc_margin, obs_weight, fu, fd, su, sd, color_im, depth_im,
                       rem_im, t->wd_px, t->col_epoch, t->epoch, t->bits, words_z, zw_snap, t->dct, t->chunk_epoch);
  }
  // LIDARHIP_DEBUG_TSDF=1: pairs / candidate voxels / written voxels of the launch (lt_debug_tsdf_pix_counts)
  static const bool want_cnt = getenv("LIDARHIP_DEBUG_TSDF") != nullptr;
  if (want_cnt && !g_pix_dbg) LT_HIP(hipMalloc((void**)&g_pix_dbg, 8 * sizeof(unsigned long long)));
  if (want_cnt) LT_HIP(hipMemsetAsync(g_pix_dbg, 0, 8 * sizeof(unsigned long long), stream));
#define LT_PIX_ARGS                                                                                                          \
  t->tsdf, t->weight, t->color, t->rem, t->dim[0], t->dim[1], t->dim[2], t->origin[0], t->origin[1], t->origin[2],           \
      t->voxel_size, 1.0f / t->voxel_size, im_h, im_w, t->trunc_margin, obs_weight, fu, fd, su, sd, color_im, depth_im, rem_im, \
      t->wd_px, t->col_epoch, t->epoch, t->bits, words_z, t->col_zw, t->chunk_epoch, t->dct, t->rowtab, t->wd_start, t->wd_ent, \
      t->wd_key,                                                                                                            \
      t->wd_rho_bits, t->wd_qscale, zw_snap, g_pix_dbg
  if (want_cnt) hipLaunchKernelGGL((k_tsdf_integrate_pix<true, true>), dim3(nb), dim3(256), 0, stream, LT_PIX_ARGS);
  else hipLaunchKernelGGL((k_tsdf_integrate_pix<true, false>), dim3(nb), dim3(256), 0, stream, LT_PIX_ARGS);
#undef LT_PIX_ARGS
  if (t->wd_n_quirk > 0) {
    const long long nv = (long long)t->wd_n_quirk * t->dim[2];
    hipLaunchKernelGGL(k_tsdf_integrate_quirk<true>, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, stream, t->tsdf, t->weight,
                       t->color, t->rem, t->dim[0], t->dim[1], t->dim[2], t->origin[0], t->origin[1], t->origin[2],
                       t->voxel_size, im_h, im_w, t->trunc_margin, obs_weight, fu, fd, su, sd, color_im, depth_im, rem_im,
                       t->wd_px, t->col_epoch, t->epoch, t->bits, words_z, t->col_zw, t->chunk_epoch, t->dct, t->wd_qcols, t->wd_n_quirk,
                       fresh_volume ? 1 : 0);
  }
  LT_HIP(hipGetLastError());
  return LT_OK;
}

// ---- LT_TSDF_HOST_MODE: the reference's numpy branch of `integrate` (FUSION_GPU_MODE == 0, fusion_lidar.py:290-388) --------
// What the reference runs wherever pycuda is absent -- and the only fusion mode of it that can be run next to this library
// without an NVIDIA GPU (goldens F8, F13, F14).  NOT the CUDA kernel's arithmetic: the voxel is projected in float64 (world
// coordinate = float32 origin + index * float64 voxel size, :299-300; norm / arctan2 / arcsin / the pixel in float64,
// :316-331), the field-of-view test is on the float64 pitch (:341-342), tsdf is the plain running average with the float32
// product `tsdf * w_old`, the float64 sum and quotient, rounded to float32 on the store (:352-366), the colour is averaged
// per channel in float32 with numpy's round-half-even (:372-388), remissions are not integrated (:390-392).  One thread per
// voxel, z fastest; expression by expression, no contraction (the file is compiled with -ffp-contract=off).
__global__ __launch_bounds__(256) void k_tsdf_integrate_host_mode(
    float* __restrict__ tsdf_vol, float* __restrict__ weight_vol, float* __restrict__ color_vol, int dim_x, int dim_y, int dim_z,
    double ox, double oy, double oz, double voxel_size, int im_h, int im_w, double trunc_margin, float obs_weight, double fov_up,
    double fov_down, const float* __restrict__ color_im, const float* __restrict__ depth_im) {
  const size_t n = (size_t)dim_x * dim_y * dim_z;
  const size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (v >= n) return;
  const int iz = (int)(v % dim_z), iy = (int)((v / dim_z) % dim_y), ix = (int)(v / ((size_t)dim_z * dim_y));
  const double x = ox + (double)ix * voxel_size, y = oy + (double)iy * voxel_size, z = oz + (double)iz * voxel_size;
  const double depth = sqrt(x * x + y * y + z * z);  // np.linalg.norm(cam_pts, 2, axis=0): sqrt(add.reduce(x * x))
  const double fov = fabs(fov_down) + fabs(fov_up);
  const double yaw = -atan2(y, x);
  const double pitch = asin(z / depth);
  double proj_x = 0.5 * (yaw / LT_PI_D + 1.0);
  double proj_y = 1.0 - (pitch + fabs(fov_down)) / fov;
  proj_x *= (double)im_w;
  proj_y *= (double)im_h;
  // np.minimum / np.maximum propagate NaN; NaN.astype(np.int32) is INT_MIN: such a voxel fails `pix >= 0` below
  const double fx = fmax(0.0, fmin((double)(im_w - 1), floor(proj_x)));
  const double fy = fmax(0.0, fmin((double)(im_h - 1), floor(proj_y)));
  if (proj_x != proj_x || proj_y != proj_y) return;
  const int pix_x = (int)fx, pix_y = (int)fy;
  if (!(pix_x >= 0 && pix_x < im_w && pix_y >= 0 && pix_y < im_h && pitch < fov_up && pitch > fov_down)) return;
  const double depth_val = (double)depth_im[(size_t)pix_y * im_w + pix_x];
  const double depth_diff = depth_val - depth;
  if (!(depth_val > 0.0 && depth_diff >= -trunc_margin)) return;
  const double dist = fmin(1.0, depth_diff / trunc_margin);
  float4* const vox = LT_VOX(tsdf_vol, v);
  const float4 o = *vox;  // (tsdf, weight, colour, remission -- which this branch of the reference never touches)
  const float w_old = o.y;
  const float w_new = w_old + obs_weight;
  const float tw = o.x * w_old;
  const float tv_new = (float)(((double)tw + dist) / (double)w_new);
  const float old_color = o.z;
  const float old_b = floorf(old_color / 65536.0f);
  const float old_g = floorf((old_color - old_b * 256.0f * 256.0f) / 256.0f);
  const float old_r = old_color - old_b * 256.0f * 256.0f - old_g * 256.0f;
  const float new_color = color_im[(size_t)pix_y * im_w + pix_x];
  float new_b = floorf(new_color / 65536.0f);
  float new_g = floorf((new_color - new_b * 256.0f * 256.0f) / 256.0f);
  float new_r = new_color - new_b * 256.0f * 256.0f - new_g * 256.0f;
  new_b = fminf(rintf((old_b * w_old + new_b) / w_new), 255.0f);
  new_g = fminf(rintf((old_g * w_old + new_g) / w_new), 255.0f);
  new_r = fminf(rintf((old_r * w_old + new_r) / w_new), 255.0f);
  *vox = make_float4(tv_new, w_new, new_b * 256.0f * 256.0f + new_g * 256.0f + new_r, o.w);
}

extern "C" int lt_tsdf_integrate_dev(lt_tsdf* t, const float* color_im, const float* depth_im, const float* rem_im,
                                     int im_h, int im_w, float obs_weight, unsigned flags, void* stream_) {
  if (!t || !color_im || !depth_im || !rem_im || im_h <= 0 || im_w <= 0) {
    lt_set_error("lt_tsdf_integrate_dev: invalid argument");
    return LT_ERR_INVALID_ARG;
  }
  if ((flags & LT_TSDF_HOST_MODE) && (flags & LT_TSDF_MERGE)) {  // (the numpy branch has no class-aware update: lidarhip.h)
    lt_set_error("lt_tsdf_integrate_dev: LT_TSDF_HOST_MODE excludes LT_TSDF_MERGE");
    return LT_ERR_INVALID_ARG;
  }
  hipStream_t stream = (hipStream_t)stream_;
  LT_HIP(hipSetDevice(t->device));
  // other_params[6] * PI / 180.0 in double, stored to float (fusion_lidar.py:124-125); the launch passes the
  // degrees as float32 (:278-280)
  if (flags & LT_TSDF_HOST_MODE) {
    // self.fov_up / 180.0 * np.pi (fusion_lidar.py:308-309); every voxel is visited: no column stamps -> all_dirty
    const double fu_d = t->fov_up_deg / 180.0 * LT_PI_D, fd_d = t->fov_down_deg / 180.0 * LT_PI_D;
    hipLaunchKernelGGL(k_tsdf_integrate_host_mode, dim3((unsigned)((t->n + 255) / 256)), dim3(256), 0, stream, t->tsdf, t->weight,
                       t->color, t->dim[0], t->dim[1], t->dim[2], (double)t->origin[0], (double)t->origin[1], (double)t->origin[2],
                       t->voxel_size_d, im_h, im_w, t->voxel_size_d * 5, obs_weight, fu_d, fd_d, color_im, depth_im);
    LT_HIP(hipGetLastError());
    t->all_dirty = 1;
    t->n_obs += 1;
    return LT_OK;
  }
  const float fu = (float)((double)(float)t->fov_up_deg * LT_PI_D / 180.0);
  const float fd = (float)((double)(float)t->fov_down_deg * LT_PI_D / 180.0);
  // ---- a fresh volume, the class-aware update: driven by the pixels (k_tsdf_integrate_pix) --------------------------------
  static const bool pix_off = []() { const char* e = getenv("LIDARHIP_TSDF_PIX"); return e && strcmp(e, "0") == 0; }();
  const bool tan_ok = fabs((double)fu) < 1.39 && fabs((double)fd) < 1.39;
  int px_bits = 1;
  while ((1 << px_bits) < im_w) ++px_bits;
  // (LIDARHIP_TSDF_PIX=1: only the first observation of a fresh volume; default: every observation)
  static const bool pix_fresh_only = []() { const char* e = getenv("LIDARHIP_TSDF_PIX"); return e && strcmp(e, "1") == 0; }();
  const bool use_pix = !pix_off && (flags & LT_TSDF_MERGE) && (t->n_obs == 0 || !pix_fresh_only) && !t->all_dirty && tan_ok &&
                       30 - px_bits >= 12 && fabsf(fu) + fabsf(fd) > 0.f;
  const bool fresh_volume = t->n_obs == 0;
  t->n_obs += 1;
  if (use_pix)
    return tsdf_integrate_pix(t, color_im, depth_im, rem_im, im_h, im_w, obs_weight, fu, fd, 30 - px_bits, fresh_volume, stream);
  if (im_w > t->cap_w) {
    if (t->colmax) {
      LT_HIP(hipDeviceSynchronize());
      (void)hipFree(t->colmax);
      t->colmax = nullptr;
    }
    LT_HIP(hipMalloc((void**)&t->colmax, (size_t)im_w * sizeof(float)));
    t->cap_w = im_w;
  }
  if ((size_t)im_w * im_h > t->cap_dct) {
    if (t->dct) {
      LT_HIP(hipDeviceSynchronize());
      (void)hipFree(t->dct);
      t->dct = nullptr;
      t->cap_dct = 0;
    }
    LT_HIP(hipMalloc((void**)&t->dct, (size_t)im_w * im_h * sizeof(float2)));
    t->cap_dct = (size_t)im_w * im_h;
  }
  const int n_cols = t->dim[0] * t->dim[1];
  const col_geom G = tsdf_geom(t);
  // behind the table: one flag per chunk of 64 columns, then the walks' z ranges and the columns' rho^2
  const size_t n_flags = ((size_t)n_cols + 63) / 64 + 64;
  unsigned* colz = (unsigned*)(t->colinfo + n_cols + n_flags);
  float* colrho2 = (float*)(t->colinfo + n_cols + n_flags + n_cols);
  hipLaunchKernelGGL(k_tsdf_colmax, dim3((im_w + 63) / 64), dim3(256), 0, stream, depth_im, color_im, im_h, im_w, t->colmax,
                     t->dct);
  hipLaunchKernelGGL(k_tsdf_columns, dim3((n_cols + 255) / 256), dim3(256), 0, stream, t->dim[0], t->dim[1], t->origin[0],
                     t->origin[1], t->voxel_size, im_w, t->trunc_margin, t->colmax, t->colinfo, t->colinfo + n_cols, G,
                     t->dim[2], colz, colrho2);
  // sine thresholds of the conservative field-of-view test: 1e-5 beyond the limits (asinf is good to ~1e-7)
  const float su = (float)(sin((double)fu) + 1e-5), sd = (float)(sin((double)fd) - 1e-5);
  static const int env_blocks = []() { const char* e = getenv("LIDARHIP_TSDF_BLOCKS"); return e ? atoi(e) : 0; }();
  const unsigned nbc = (unsigned)min((n_cols + 63) / 64, env_blocks > 0 ? env_blocks : (1 << 20));  // a chunk of 64 columns each
  // debug (LIDARHIP_DEBUG_TSDF=1, tools/tsdf_wave_times.py): start / duration of every wave at 100 MHz
  static const bool want_dbg = getenv("LIDARHIP_DEBUG_TSDF") != nullptr;
  if (want_dbg && !g_tsdf_dbg) LT_HIP(hipMalloc((void**)&g_tsdf_dbg, (size_t)LT_TSDF_DBG_WAVES * 2 * sizeof(unsigned long long)));
  unsigned long long* dbg = g_tsdf_dbg;
  const float fov_abs = fabsf(fu) + fabsf(fd);
  // the band test's pitch polynomial is good to 1e-5 rad for |sin| <= 0.5 and its row choice has +-0.25 row of slack: it is
  // used for fields of view inside +-30 degrees with rows at least 2e-4 rad apart (every spinning LiDAR); otherwise
  // kA = NaN switches it off and every voxel takes the exact evaluation
  const bool band_ok = fabsf(su) <= 0.5f && fabsf(sd) <= 0.5f && fov_abs / (float)im_h >= 2e-4f;
  const float kA = band_ok ? -(float)im_h / fov_abs : NAN, kB = (float)im_h * (1.0f - fabsf(fd) / fov_abs);
  if (flags & LT_TSDF_MERGE)
    hipLaunchKernelGGL(k_tsdf_integrate_cols<true>, dim3(nbc), dim3(256), 0, stream, t->tsdf, t->weight, t->color, t->rem,
                       t->dim[0], t->dim[1], t->dim[2], t->origin[0], t->origin[1], t->origin[2], t->voxel_size, im_h, im_w,
                       t->trunc_margin, obs_weight, fu, fd, su, sd, color_im, depth_im, rem_im, t->colinfo, t->col_epoch,
                       t->epoch, G, t->bits, (t->dim[2] + 63) / 64, t->col_zw, t->dct, kA, kB, t->colinfo + n_cols, colz, colrho2, t->all_dirty, dbg,
                       t->chunk_epoch);
  else
    hipLaunchKernelGGL(k_tsdf_integrate_cols<false>, dim3(nbc), dim3(256), 0, stream, t->tsdf, t->weight, t->color, t->rem,
                       t->dim[0], t->dim[1], t->dim[2], t->origin[0], t->origin[1], t->origin[2], t->voxel_size, im_h, im_w,
                       t->trunc_margin, obs_weight, fu, fd, su, sd, color_im, depth_im, rem_im, t->colinfo, t->col_epoch,
                       t->epoch, G, t->bits, (t->dim[2] + 63) / 64, t->col_zw, t->dct, kA, kB, t->colinfo + n_cols, colz, colrho2, t->all_dirty, dbg,
                       t->chunk_epoch);
  LT_HIP(hipGetLastError());
  return LT_OK;
}

// see include/lidarhip.h
extern "C" int lt_tsdf_integrate_multi_dev(lt_tsdf* t, int n_obs, const float* const* color_ims, const float* const* depth_ims,
                                           const float* const* rem_ims, int im_h, int im_w, float obs_weight, unsigned flags,
                                           void* stream_) {
  if (!t || n_obs < 0 || (n_obs > 0 && (!color_ims || !depth_ims || !rem_ims)) || im_h <= 0 || im_w <= 0) {
    lt_set_error("lt_tsdf_integrate_multi_dev: invalid argument");
    return LT_ERR_INVALID_ARG;
  }
  for (int k = 0; k < n_obs; ++k)
    if (!color_ims[k] || !depth_ims[k] || !rem_ims[k]) {
      lt_set_error("lt_tsdf_integrate_multi_dev: observation %d has a NULL image", k);
      return LT_ERR_INVALID_ARG;
    }
  hipStream_t stream = (hipStream_t)stream_;
  LT_HIP(hipSetDevice(t->device));
  const float fu = (float)((double)(float)t->fov_up_deg * LT_PI_D / 180.0);
  const float fd = (float)((double)(float)t->fov_down_deg * LT_PI_D / 180.0);
  static const bool pix_off = []() { const char* e = getenv("LIDARHIP_TSDF_PIX"); return e && e[0] != 0 && strcmp(e, "2") != 0; }();
  static const bool multi_off = []() { const char* e = getenv("LIDARHIP_TSDF_MULTI"); return e && strcmp(e, "0") == 0; }();
  const bool tan_ok = fabs((double)fu) < 1.39 && fabs((double)fd) < 1.39;
  int px_bits = 1;
  while ((1 << px_bits) < im_w) ++px_bits;
  // the fused pass: the class-aware update of a FRESH volume, under the conditions of the pixel-centric integrate
  const bool fuse = !pix_off && !multi_off && (flags & LT_TSDF_MERGE) && !(flags & LT_TSDF_HOST_MODE) && t->n_obs == 0 && !t->all_dirty && tan_ok &&
                    30 - px_bits >= 12 && fabsf(fu) + fabsf(fd) > 0.f && n_obs >= 2;
  int done = 0;
  if (fuse) {
    const int n = n_obs < LT_TSDF_MULTI_MAX ? n_obs : LT_TSDF_MULTI_MAX;
    const int rho_bits = 30 - px_bits;
    if (t->wd_w != im_w || t->wd_rho_bits != rho_bits) LT_CHECK(tsdf_wedge_build(t, im_w, rho_bits, stream));
    if (t->rowtab_for_h != im_h || !t->rowtab) {
      t->rowtab_for_h = 0;
      LT_CHECK(tsdf_rowtab(t, im_h, fu, fd, stream));
      t->rowtab_for_h = im_h;
    }
    const size_t n_pix = (size_t)im_h * im_w;
    if ((size_t)n * n_pix > t->cap_obs4) {
      if (t->obs4) { LT_HIP(hipStreamSynchronize(stream)); (void)hipFree(t->obs4); t->obs4 = nullptr; t->cap_obs4 = 0; }
      LT_HIP(hipMalloc((void**)&t->obs4, (size_t)LT_TSDF_MULTI_MAX * n_pix * sizeof(float4)));
      t->cap_obs4 = (size_t)LT_TSDF_MULTI_MAX * n_pix;
    }
    tsdf_obs_ptrs O;
    for (int k = 0; k < LT_TSDF_MULTI_MAX; ++k) {
      const int q = k < n ? k : 0;
      O.color[k] = color_ims[q]; O.depth[k] = depth_ims[q]; O.rem[k] = rem_ims[q];
    }
    hipLaunchKernelGGL(k_tsdf_dct4, dim3((unsigned)((n * n_pix + 255) / 256)), dim3(256), 0, stream, O, n, im_h, im_w, t->obs4);
    const float su = (float)(sin((double)fu) + 1e-5), sd = (float)(sin((double)fd) - 1e-5);
    const int words_z = (t->dim[2] + 63) / 64;
    static const int env_wgs = []() { const char* e = getenv("LIDARHIP_PIX_WGS"); return e ? atoi(e) : -1; }();
    const int res_wgs = lt_cu_count(t->device) * 5;
    const int groups = (int)((n_pix + 63) / 64);
    const unsigned nb = (unsigned)min(groups, env_wgs > 0 ? env_wgs : (env_wgs == 0 ? (1 << 20) : res_wgs));
    hipLaunchKernelGGL((k_tsdf_integrate_pix_multi<false>), dim3(nb), dim3(256), 0, stream, t->tsdf, t->weight, t->color, t->rem,
                       t->dim[0], t->dim[1], t->dim[2], t->origin[0], t->origin[1], t->origin[2], t->voxel_size,
                       1.0f / t->voxel_size, im_h, im_w, t->trunc_margin, obs_weight, fu, fd, su, sd, (const float4*)t->obs4, n,
                       t->wd_px, t->col_epoch, t->epoch, t->bits, words_z, t->col_zw, t->chunk_epoch, t->rowtab, t->wd_start,
                       t->wd_ent, t->wd_key, t->wd_rho_bits, t->wd_qscale, (unsigned long long*)nullptr);
    if (t->wd_n_quirk > 0) {
      const long long nv = (long long)t->wd_n_quirk * t->dim[2];
      hipLaunchKernelGGL(k_tsdf_integrate_quirk_multi, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, stream, t->tsdf,
                         t->weight, t->color, t->rem, t->dim[0], t->dim[1], t->dim[2], t->origin[0], t->origin[1], t->origin[2],
                         t->voxel_size, im_h, im_w, t->trunc_margin, obs_weight, fu, fd, su, sd, (const float4*)t->obs4, n,
                         t->wd_px, t->col_epoch, t->epoch, t->bits, words_z, t->col_zw, t->chunk_epoch, t->wd_qcols,
                         t->wd_n_quirk);
    }
    LT_HIP(hipGetLastError());
    t->n_obs += n;
    done = n;
  }
  for (int k = done; k < n_obs; ++k)
    LT_CHECK(lt_tsdf_integrate_dev(t, color_ims[k], depth_ims[k], rem_ims[k], im_h, im_w, obs_weight, flags, stream_));
  return LT_OK;
}

// The caller wrote into the volumes through the pointers of lt_tsdf_volumes: every column counts as written from
// now on (until the next reset) -- reset and marching cubes must not skip anything.
extern "C" int lt_tsdf_touch(lt_tsdf* t) {
  if (!t) {
    lt_set_error("lt_tsdf_touch: NULL volume");
    return LT_ERR_INVALID_ARG;
  }
  t->all_dirty = 1;
  return LT_OK;
}

extern "C" int lt_tsdf_volumes(lt_tsdf* t, int* dims, float* origin, float** tsdf, float** weight, float** color,
                               float** rem) {
  if (!t) {
    lt_set_error("lt_tsdf_volumes: NULL volume");
    return LT_ERR_INVALID_ARG;
  }
  for (int k = 0; k < 3; ++k) {
    if (dims) dims[k] = t->dim[k];
    if (origin) origin[k] = t->origin[k];
  }
  if (tsdf) *tsdf = t->tsdf;      // the fields of voxel 0: voxel v's are LT_VOLUME_STRIDE * v floats further (LT_VOX)
  if (weight) *weight = t->weight;
  if (color) *color = t->color;
  if (rem) *rem = t->rem;
  return LT_OK;
}

extern "C" int lt_tsdf_volume_stride(void) { return 4; }

// debug helper (not part of the documented ABI): {pairs, candidate voxels, written voxels} of the last pixel-centric
// integrate (LIDARHIP_DEBUG_TSDF=1)
extern "C" int lt_debug_tsdf_pix_counts(unsigned long long* out8) {
  if (!out8 || !g_pix_dbg) return LT_ERR_INVALID_ARG;
  LT_HIP(hipDeviceSynchronize());
  LT_HIP(hipMemcpy(out8, g_pix_dbg, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return LT_OK;
}

// debug helper (not part of the documented ABI): the per-wave stamps of the last column-aware integrate
// (LIDARHIP_DEBUG_TSDF=1); out: [n_waves][2] = {start, duration} at 100 MHz
extern "C" int lt_debug_tsdf_wave_times(unsigned long long* out, int n_waves) {
  if (!out || n_waves < 0 || n_waves > LT_TSDF_DBG_WAVES || !g_tsdf_dbg) return LT_ERR_INVALID_ARG;
  LT_HIP(hipDeviceSynchronize());
  LT_HIP(hipMemcpy(out, g_tsdf_dbg, (size_t)n_waves * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return LT_OK;
}
