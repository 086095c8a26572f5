// ORACLE / TEST INFRASTRUCTURE ONLY -- not part of liblidarhip.so; nothing in lidar_transfer_amd/ links or loads it.
//
// One thread per voxel: the reference's pycuda kernel `integrate` (/root/reference auxiliary/fusion_lidar.py:66-229,
// CUDA source inside a Python string) restated statement by statement for the GPU, as the A/B baseline of the shipped
// column-aware kernel (lidar_transfer_amd/csrc/lt_tsdf.hip: k_tsdf_integrate_cols).  The arithmetic order is the
// reference's -- that is what the bit-parity tests of tests/test_tsdf_gpu.py compare the product against.
//
// PINNED to the reference's source: tests/test_tsdf_ref_kernel_gpu.py runs the reference's kernel text, compiled unmodified
// by hipcc for gfx950 (oracle/build_ref_tsdf.py -> oracle/_ref/libref_tsdf_integrate.so), next to this file's kernel on the
// same observations -- all four volumes bit-identical, up to 72 M voxels.  The explicit __fmaf_rn below are therefore the
// contractions hipcc applies to the reference's text under nvcc's default (-fmad=true == -ffp-contract=fast); this file is
// built with -ffp-contract=off so that nothing else fuses.  NOT pinned: a CUDA run (CUDA's own norm3df / atan2 / asinf).
//
// Built by oracle/Makefile (target `dense`) into oracle/liblt_tsdf_dense.so; loaded by oracle.binding.dense_lib();
// operates on raw device pointers.
#include <hip/hip_runtime.h>
#include <math.h>

#define LT_PI_D 3.14159265358979323846

template <bool MERGE>
__device__ __forceinline__ void dense_update(float* __restrict__ tsdf_vol, float* __restrict__ weight_vol,
                                             float* __restrict__ color_vol, float* __restrict__ rem_vol, int voxel_idx,
                                             float dist, float obs_weight, float new_color, float new_rem) {
  if (!MERGE) {  // fusion_lidar.py:215-228 (plain running average, colour channels averaged separately)
    const float w_old = weight_vol[voxel_idx];
    const float w_new = w_old + obs_weight;
    weight_vol[voxel_idx] = w_new;
    tsdf_vol[voxel_idx] = __fmaf_rn(tsdf_vol[voxel_idx], w_old, dist) / w_new;
    const float old_color = color_vol[voxel_idx];
    const float old_b = floorf(old_color / (256 * 256));
    const float old_g = floorf((old_color - old_b * 256 * 256) / 256);
    const float old_r = old_color - old_b * 256 * 256 - old_g * 256;
    float new_b = floorf(new_color / (256 * 256));
    float new_g = floorf((new_color - new_b * 256 * 256) / 256);
    float new_r = new_color - new_b * 256 * 256 - new_g * 256;
    new_b = fminf(roundf(__fmaf_rn(old_b, w_old, new_b) / w_new), 255.0f);
    new_g = fminf(roundf(__fmaf_rn(old_g, w_old, new_g) / w_new), 255.0f);
    new_r = fminf(roundf(__fmaf_rn(old_r, w_old, new_r) / w_new), 255.0f);
    color_vol[voxel_idx] = new_b * 256 * 256 + new_g * 256 + new_r;
    rem_vol[voxel_idx] = __fmaf_rn(rem_vol[voxel_idx], w_old, new_rem) / w_new;
  } else {  // fusion_lidar.py:177-213 (class-aware: same class integrates, another class wins when closer)
    const float dist_old = weight_vol[voxel_idx];  // sic: the reference compares against the weight volume
    const float old_color = color_vol[voxel_idx];
    if (old_color == new_color) {
      const float w_old = weight_vol[voxel_idx];
      const float w_new = w_old + obs_weight;
      weight_vol[voxel_idx] = w_new;
      tsdf_vol[voxel_idx] = __fmaf_rn(tsdf_vol[voxel_idx], w_old, dist) / w_new;
      rem_vol[voxel_idx] = __fmaf_rn(rem_vol[voxel_idx], w_old, new_rem) / w_new;
    } else if (dist < dist_old) {
      tsdf_vol[voxel_idx] = dist;
      const float new_b = floorf(new_color / (256 * 256));
      const float new_g = floorf((new_color - new_b * 256 * 256) / 256);
      const float new_r = new_color - new_b * 256 * 256 - new_g * 256;
      color_vol[voxel_idx] = new_b * 256 * 256 + new_g * 256 + new_r;
      rem_vol[voxel_idx] = new_rem;
    }
  }
}

template <bool MERGE>
__global__ __launch_bounds__(256) void k_dense_integrate(float* __restrict__ tsdf_vol, float* __restrict__ weight_vol,
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
  // norm3df: the device library's, as the reference's own source gets it when hipcc compiles it (oracle/_ref/
  // libref_tsdf_integrate.so -- bit-identical volumes, tests/test_tsdf_ref_kernel_gpu.py).  (ocml: the magnitudes sorted
  // a >= b >= c, scaled by a's exponent, sqrt(fma(a, a, fma(b, b, c * c))) with the hardware's v_sqrt_f32.)
  const float depth = norm3df(pt_x, pt_y, pt_z);
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
  dense_update<MERGE>(tsdf_vol, weight_vol, color_vol, rem_vol, voxel_idx, dist, obs_weight, color_im[py * im_w + px],
                      rem_im[py * im_w + px]);
}

// fov_*_deg: the degrees the TSDFVolume was constructed with; converted like the reference's launch does
// (other_params[6] * PI / 180.0 in double, stored to float -- fusion_lidar.py:124-125, :278-280).  Returns the hipError.
extern "C" int lt_test_tsdf_integrate_dense(float* tsdf, float* weight, float* color, float* rem, const int* dims,
                                            const float* origin, float voxel_size, float trunc_margin, float fov_up_deg,
                                            float fov_down_deg, const float* color_im, const float* depth_im,
                                            const float* rem_im, int im_h, int im_w, float obs_weight, int merge,
                                            void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const float fu = (float)((double)fov_up_deg * LT_PI_D / 180.0), fd = (float)((double)fov_down_deg * LT_PI_D / 180.0);
  const long long n = (long long)dims[0] * dims[1] * dims[2];
  const unsigned nb = (unsigned)((n + 255) / 256);
  if (merge)
    hipLaunchKernelGGL(k_dense_integrate<true>, dim3(nb), dim3(256), 0, stream, tsdf, weight, color, rem, dims[0], dims[1],
                       dims[2], origin[0], origin[1], origin[2], voxel_size, im_h, im_w, trunc_margin, obs_weight, fu, fd,
                       color_im, depth_im, rem_im);
  else
    hipLaunchKernelGGL(k_dense_integrate<false>, dim3(nb), dim3(256), 0, stream, tsdf, weight, color, rem, dims[0], dims[1],
                       dims[2], origin[0], origin[1], origin[2], voxel_size, im_h, im_w, trunc_margin, obs_weight, fu, fd,
                       color_im, depth_im, rem_im);
  return (int)hipGetLastError();
}
