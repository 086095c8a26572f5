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
// nvcc contracts them by default (-fmad=true).  CUDA's norm3df / atan2f / asinf are not available bit for bit
// on any other platform: voxels whose projection falls within an ulp of a pixel or field-of-view boundary
// may land differently -- tests bound that fraction.
#include "lt_internal.h"
#include <math.h>

#define LT_PI_D 3.14159265358979323846

__global__ __launch_bounds__(256) void k_tsdf_fill(float* __restrict__ tsdf, float* __restrict__ weight,
                                                   float* __restrict__ color, float* __restrict__ rem, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    tsdf[i] = 1.0f;  // np.ones (fusion_lidar.py:47)
    weight[i] = 0.0f;
    color[i] = 0.0f;
    rem[i] = 0.0f;
  }
}

template <bool MERGE>
__global__ __launch_bounds__(256) void k_tsdf_integrate(float* __restrict__ tsdf_vol, float* __restrict__ weight_vol,
                                                        float* __restrict__ color_vol, float* __restrict__ rem_vol,
                                                        int vol_dim_x, int vol_dim_y, int vol_dim_z, float ox, float oy,
                                                        float oz, float voxel_size, int im_h, int im_w,
                                                        float trunc_margin, float obs_weight, float fov_up,
                                                        float fov_down, const float* __restrict__ color_im,
                                                        const float* __restrict__ depth_im,
                                                        const float* __restrict__ rem_im) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long long)vol_dim_x * vol_dim_y * vol_dim_z) return;  // the reference tests `>` (one past the end)
  const int voxel_idx = (int)gid;
  // voxel grid coordinates -- float division exactly as the reference ("be careful when casting", :95-98)
  const float voxel_x = floorf(((float)voxel_idx) / ((float)(vol_dim_y * vol_dim_z)));
  const float voxel_y = floorf(((float)(voxel_idx - ((int)voxel_x) * vol_dim_y * vol_dim_z)) / ((float)vol_dim_z));
  const float voxel_z = (float)(voxel_idx - ((int)voxel_x) * vol_dim_y * vol_dim_z - ((int)voxel_y) * vol_dim_z);
  const float pt_x = __fmaf_rn(voxel_x, voxel_size, ox);
  const float pt_y = __fmaf_rn(voxel_y, voxel_size, oy);
  const float pt_z = __fmaf_rn(voxel_z, voxel_size, oz);
  // spherical projection (:120-146); cam_pose is not used by the reference kernel (:112-114)
  const float fov = fabsf(fov_up) + fabsf(fov_down);
  const float depth = sqrtf(__fmaf_rn(pt_z, pt_z, __fmaf_rn(pt_y, pt_y, pt_x * pt_x)));  // norm3df
  const float yaw = -atan2f(pt_y, pt_x);
  const float pitch = asinf(pt_z / depth);
  if (pitch > fov_up || pitch < fov_down) return;
  float proj_x = (float)(0.5 * ((double)yaw / LT_PI_D + 1.0));
  float proj_y = (float)(1.0 - (double)((pitch + fabsf(fov_down)) / fov));
  proj_x *= (float)im_w;
  proj_y *= (float)im_h;
  int px = (int)floorf(proj_x);
  px = min(im_w - 1, px);
  px = max(0, px);
  int py = (int)floorf(proj_y);
  py = min(im_h - 1, py);
  py = max(0, py);
  const float depth_value = depth_im[py * im_w + px];
  if (depth_value == 0.f) return;
  const float depth_diff = depth_value - depth;
  if (depth_diff < -trunc_margin) return;
  const float dist = fminf(1.0f, depth_diff / trunc_margin);
  if (!MERGE) {
    const float w_old = weight_vol[voxel_idx];
    const float w_new = w_old + obs_weight;
    weight_vol[voxel_idx] = w_new;
    tsdf_vol[voxel_idx] = __fmaf_rn(tsdf_vol[voxel_idx], w_old, dist) / w_new;
    const float old_color = color_vol[voxel_idx];
    const float old_b = floorf(old_color / (256 * 256));
    const float old_g = floorf((old_color - old_b * 256 * 256) / 256);
    const float old_r = old_color - old_b * 256 * 256 - old_g * 256;
    const float new_color = color_im[py * im_w + px];
    float new_b = floorf(new_color / (256 * 256));
    float new_g = floorf((new_color - new_b * 256 * 256) / 256);
    float new_r = new_color - new_b * 256 * 256 - new_g * 256;
    new_b = fminf(roundf(__fmaf_rn(old_b, w_old, new_b) / w_new), 255.0f);
    new_g = fminf(roundf(__fmaf_rn(old_g, w_old, new_g) / w_new), 255.0f);
    new_r = fminf(roundf(__fmaf_rn(old_r, w_old, new_r) / w_new), 255.0f);
    color_vol[voxel_idx] = new_b * 256 * 256 + new_g * 256 + new_r;
    rem_vol[voxel_idx] = __fmaf_rn(rem_vol[voxel_idx], w_old, rem_im[py * im_w + px]) / w_new;
  } else {
    const float dist_old = weight_vol[voxel_idx];  // sic: the reference compares against the weight volume
    const float old_color = color_vol[voxel_idx];
    const float new_color = color_im[py * im_w + px];
    if (old_color == new_color) {  // same class: integrate
      const float w_old = weight_vol[voxel_idx];
      const float w_new = w_old + obs_weight;
      weight_vol[voxel_idx] = w_new;
      tsdf_vol[voxel_idx] = __fmaf_rn(tsdf_vol[voxel_idx], w_old, dist) / w_new;
      rem_vol[voxel_idx] = __fmaf_rn(rem_vol[voxel_idx], w_old, rem_im[py * im_w + px]) / w_new;
    } else if (dist < dist_old) {  // other class: the closer observation wins
      tsdf_vol[voxel_idx] = dist;
      const float new_b = floorf(new_color / (256 * 256));
      const float new_g = floorf((new_color - new_b * 256 * 256) / 256);
      const float new_r = new_color - new_b * 256 * 256 - new_g * 256;
      color_vol[voxel_idx] = new_b * 256 * 256 + new_g * 256 + new_r;
      rem_vol[voxel_idx] = rem_im[py * im_w + px];
    }
  }
}

extern "C" int lt_tsdf_destroy(lt_tsdf* t) {
  if (!t) return LT_OK;
  (void)hipSetDevice(t->device);
  (void)hipDeviceSynchronize();
  float* ps[] = {t->tsdf, t->weight, t->color, t->rem};
  for (float* p : ps)
    if (p) (void)hipFree(p);
  free(t);
  return LT_OK;
}

extern "C" int lt_tsdf_reset(lt_tsdf* t, void* stream) {
  if (!t) {
    lt_set_error("lt_tsdf_reset: NULL volume");
    return LT_ERR_INVALID_ARG;
  }
  LT_HIP(hipSetDevice(t->device));
  hipLaunchKernelGGL(k_tsdf_fill, dim3(4096), dim3(256), 0, (hipStream_t)stream, t->tsdf, t->weight, t->color, t->rem,
                     t->n);
  LT_HIP(hipGetLastError());
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
  t->n = (size_t)n;
  t->voxel_size = (float)voxel_size;
  t->trunc_margin = (float)(voxel_size * 5);  // fusion_lidar.py:31
  t->fov_up_deg = fov_up;
  t->fov_down_deg = fov_down;
  float** ps[] = {&t->tsdf, &t->weight, &t->color, &t->rem};
  for (float** p : ps) {
    if (hipMalloc((void**)p, t->n * sizeof(float)) != hipSuccess) {
      lt_set_error("lt_tsdf_create: hipMalloc of %zu bytes failed", t->n * sizeof(float));
      lt_tsdf_destroy(t);
      return LT_ERR_NO_MEMORY;
    }
  }
  const int rc = lt_tsdf_reset(t, nullptr);
  if (rc != LT_OK) {
    lt_tsdf_destroy(t);
    return rc;
  }
  LT_HIP(hipDeviceSynchronize());
  *out = t;
  return LT_OK;
}

extern "C" int lt_tsdf_integrate_dev(lt_tsdf* t, const float* color_im, const float* depth_im, const float* rem_im,
                                     int im_h, int im_w, float obs_weight, unsigned flags, void* stream) {
  if (!t || !color_im || !depth_im || !rem_im || im_h <= 0 || im_w <= 0) {
    lt_set_error("lt_tsdf_integrate_dev: invalid argument");
    return LT_ERR_INVALID_ARG;
  }
  LT_HIP(hipSetDevice(t->device));
  // other_params[6] * PI / 180.0 in double, stored to float (fusion_lidar.py:124-125); the launch passes the
  // degrees as float32 (:278-280)
  const float fu = (float)((double)(float)t->fov_up_deg * LT_PI_D / 180.0);
  const float fd = (float)((double)(float)t->fov_down_deg * LT_PI_D / 180.0);
  const unsigned nb = (unsigned)((t->n + 255) / 256);
  if (flags & LT_TSDF_MERGE)
    hipLaunchKernelGGL(k_tsdf_integrate<true>, dim3(nb), dim3(256), 0, (hipStream_t)stream, t->tsdf, t->weight,
                       t->color, t->rem, t->dim[0], t->dim[1], t->dim[2], t->origin[0], t->origin[1], t->origin[2],
                       t->voxel_size, im_h, im_w, t->trunc_margin, obs_weight, fu, fd, color_im, depth_im, rem_im);
  else
    hipLaunchKernelGGL(k_tsdf_integrate<false>, dim3(nb), dim3(256), 0, (hipStream_t)stream, t->tsdf, t->weight,
                       t->color, t->rem, t->dim[0], t->dim[1], t->dim[2], t->origin[0], t->origin[1], t->origin[2],
                       t->voxel_size, im_h, im_w, t->trunc_margin, obs_weight, fu, fd, color_im, depth_im, rem_im);
  LT_HIP(hipGetLastError());
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
  if (tsdf) *tsdf = t->tsdf;
  if (weight) *weight = t->weight;
  if (color) *color = t->color;
  if (rem) *rem = t->rem;
  return LT_OK;
}
