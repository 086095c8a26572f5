/*
 * lt_tsdf_oracle.c -- CPU restatement of the reference's TSDF `integrate` kernel (TEST INFRASTRUCTURE).
 *
 * PARITY: the reference implements `integrate` only as a CUDA kernel embedded in Python (auxiliary/
 * fusion_lidar.py:66-229, run through pycuda); this file restates that source line by line in C.  What pins it:
 *   - its GPU twin oracle/lt_tsdf_dense.hip is bit-identical, on all four volumes, to the reference's kernel source
 *     compiled unmodified by hipcc for gfx950 (oracle/_ref/libref_tsdf_integrate.so, tests/test_tsdf_ref_kernel_gpu.py);
 *     this C file agrees with both up to what a CPU cannot restate: the device library's atan2f / asinf and its
 *     norm3df (sorted magnitudes, sqrt(fma(a, a, fma(b, b, c * c))) with the hardware's 1-ulp v_sqrt_f32 -- the same
 *     formula below with libm's correctly rounded sqrtf): class and weight volumes equal but for voxels on a pixel /
 *     fov / truncation boundary, tsdf within a few ulp (tests/test_tsdf_gpu.py);
 *   - the `merge == false` branch (:178-205) is additionally cross-checked against the reference's own numpy CPU mode
 *     (fusion_lidar.py:289-392) by the goldens F8 (float64 pixel maths there, float32 here -> tolerance).
 * NOT pinned: a CUDA run (no NVIDIA device here) -- CUDA's last-ulp behaviour of norm3df / atan2 / asinf.
 * `a + b * c` is written fmaf(b, c, a): nvcc contracts these by default.
 */
#include <math.h>
#include <stddef.h>

#define PI 3.14159265358979323846

void lto_tsdf_integrate(float* tsdf_vol, float* weight_vol, float* color_vol, float* rem_vol, int dx, int dy, int dz,
                        const float* origin, float voxel_size, int im_h, int im_w, float trunc_margin,
                        float obs_weight, float fov_up_deg, float fov_down_deg, const float* color_im,
                        const float* depth_im, const float* rem_im, int merge) {
  const float fov_up = (float)((double)fov_up_deg * PI / 180.0), fov_down = (float)((double)fov_down_deg * PI / 180.0);
  const float fov = fabsf(fov_up) + fabsf(fov_down);
  const long long n = (long long)dx * dy * dz;
#pragma omp parallel for schedule(static)
  for (long long g = 0; g < n; ++g) {
    const int voxel_idx = (int)g;
    const float voxel_x = floorf(((float)voxel_idx) / ((float)(dy * dz)));
    const float voxel_y = floorf(((float)(voxel_idx - ((int)voxel_x) * dy * dz)) / ((float)dz));
    const float voxel_z = (float)(voxel_idx - ((int)voxel_x) * dy * dz - ((int)voxel_y) * dz);
    const float pt_x = fmaf(voxel_x, voxel_size, origin[0]);
    const float pt_y = fmaf(voxel_y, voxel_size, origin[1]);
    const float pt_z = fmaf(voxel_z, voxel_size, origin[2]);
    float depth;
    { /* norm3df as the device library computes it (ocml len3): magnitudes sorted a >= b >= c */
      float a = fabsf(pt_x), b = fabsf(pt_y), c = fabsf(pt_z), t;
      if (a < b) { t = a; a = b; b = t; }
      if (a < c) { t = a; a = c; c = t; }
      if (b < c) { t = b; b = c; c = t; }
      depth = sqrtf(fmaf(a, a, fmaf(b, b, c * c)));
    }
    const float yaw = -atan2f(pt_y, pt_x);
    const float pitch = asinf(pt_z / depth);
    if (pitch > fov_up || pitch < fov_down) continue;
    float proj_x = (float)(0.5 * ((double)yaw / PI + 1.0));
    float proj_y = (float)(1.0 - (double)((pitch + fabsf(fov_down)) / fov));
    proj_x *= (float)im_w;
    proj_y *= (float)im_h;
    int px = (int)floorf(proj_x);
    px = px < im_w - 1 ? px : im_w - 1;
    px = px > 0 ? px : 0;
    int py = (int)floorf(proj_y);
    py = py < im_h - 1 ? py : im_h - 1;
    py = py > 0 ? py : 0;
    const float depth_value = depth_im[py * im_w + px];
    if (depth_value == 0.f) continue;
    const float depth_diff = depth_value - depth;
    if (depth_diff < -trunc_margin) continue;
    const float dist = fminf(1.0f, depth_diff / trunc_margin);
    if (!merge) {
      const float w_old = weight_vol[g], w_new = w_old + obs_weight;
      weight_vol[g] = w_new;
      tsdf_vol[g] = fmaf(tsdf_vol[g], w_old, dist) / w_new;
      const float old_color = color_vol[g];
      const float old_b = floorf(old_color / (256 * 256));
      const float old_g = floorf((old_color - old_b * 256 * 256) / 256);
      const float old_r = old_color - old_b * 256 * 256 - old_g * 256;
      const float new_color = color_im[py * im_w + px];
      float new_b = floorf(new_color / (256 * 256));
      float new_g = floorf((new_color - new_b * 256 * 256) / 256);
      float new_r = new_color - new_b * 256 * 256 - new_g * 256;
      new_b = fminf(roundf(fmaf(old_b, w_old, new_b) / w_new), 255.0f);
      new_g = fminf(roundf(fmaf(old_g, w_old, new_g) / w_new), 255.0f);
      new_r = fminf(roundf(fmaf(old_r, w_old, new_r) / w_new), 255.0f);
      color_vol[g] = new_b * 256 * 256 + new_g * 256 + new_r;
      rem_vol[g] = fmaf(rem_vol[g], w_old, rem_im[py * im_w + px]) / w_new;
    } else {
      const float dist_old = weight_vol[g];
      const float old_color = color_vol[g], new_color = color_im[py * im_w + px];
      if (old_color == new_color) {
        const float w_old = weight_vol[g], w_new = w_old + obs_weight;
        weight_vol[g] = w_new;
        tsdf_vol[g] = fmaf(tsdf_vol[g], w_old, dist) / w_new;
        rem_vol[g] = fmaf(rem_vol[g], w_old, rem_im[py * im_w + px]) / w_new;
      } else if (dist < dist_old) {
        tsdf_vol[g] = dist;
        const float new_b = floorf(new_color / (256 * 256));
        const float new_g = floorf((new_color - new_b * 256 * 256) / 256);
        const float new_r = new_color - new_b * 256 * 256 - new_g * 256;
        color_vol[g] = new_b * 256 * 256 + new_g * 256 + new_r;
        rem_vol[g] = rem_im[py * im_w + px];
      }
    }
  }
}
