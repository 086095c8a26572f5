// lt_post.hip -- what the reference does with the rendered images right after the hot path:
//   * do_reverse_projection_new  (auxiliary/laserscan.py:475-501): range image + pixel coordinates -> xyz
//   * MultiSemLaserScan.write    (auxiliary/laserscan.py:1121-1178): drop invalid cells, pack the scan as
//     SemanticKITTI `.bin` ([N,4] f32: x, y, z, remission) and `.label` ([N] u32) -- the per-point
//     struct.pack loop of the reference becomes one stable stream compaction on the GPU
//   * compare / iouEval          (auxiliary/laserscan.py:1181-1301, np_ioueval.py:31-70): masked confusion
//     matrix (atomic histogram) and squared range / remission differences
#include "lt_internal.h"
#include <math.h>
#include <mutex>

// ---- reverse projection (float64, as numpy computes it from int32 / float64 pixel coordinates) ------------
template <typename P>
__global__ __launch_bounds__(256) void k_reverse(const float* __restrict__ range, const P* __restrict__ px,
                                                 const P* __restrict__ py, int n, double W, double H, double fov,
                                                 double abs_fov_down, double* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double depth = (double)range[i];
  const double x = (double)px[i] / W, y = (double)py[i] / H;
  const double yaw = (x * 2 - 1.0) * M_PI;                                  // theta
  const double pitch = M_PI / 2 - (1.0 * fov - y * fov - abs_fov_down);     // 90 - phi
  const double sp = sin(pitch);
  out[3 * (size_t)i] = depth * sp * cos(-yaw);
  out[3 * (size_t)i + 1] = depth * sp * sin(-yaw);
  out[3 * (size_t)i + 2] = depth * cos(pitch);
}

extern "C" int lt_reverse_projection_dev(const float* range_img, const void* proj_x, const void* proj_y,
                                         int coords_are_f64, double fov_up, double fov_down, int H, int W,
                                         double* back_points, void* stream) {
  if (H <= 0 || W <= 0 || !range_img || !proj_x || !proj_y || !back_points) {
    lt_set_error("lt_reverse_projection_dev: invalid argument");
    return LT_ERR_INVALID_ARG;
  }
  const double fu = fov_up / 180.0 * M_PI, fd = fov_down / 180.0 * M_PI;
  const double fov = fabs(fd) + fabs(fu);
  const int n = H * W;
  hipStream_t st = (hipStream_t)stream;
  if (coords_are_f64)
    hipLaunchKernelGGL(k_reverse<double>, dim3((n + 255) / 256), dim3(256), 0, st, range_img, (const double*)proj_x,
                       (const double*)proj_y, n, (double)W, (double)H, fov, fabs(fd), back_points);
  else
    hipLaunchKernelGGL(k_reverse<int>, dim3((n + 255) / 256), dim3(256), 0, st, range_img, (const int*)proj_x,
                       (const int*)proj_y, n, (double)W, (double)H, fov, fabs(fd), back_points);
  LT_HIP(hipGetLastError());
  return LT_OK;
}

// ---- scan packer ----------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ bool pack_keep(const T* __restrict__ pts, const int* __restrict__ label,
                                          const int* __restrict__ index, int i) {
  if (index && !(index[i] > 0)) return false;            // 'cp': index > 0 (laserscan.py:1138)
  if (label[i] < 0) return false;                        // laserscan.py:1147
  const T s = (pts[3 * (size_t)i] + pts[3 * (size_t)i + 1]) + pts[3 * (size_t)i + 2];
  return s != (T)0;                                      // remove points with (0, 0, 0) (laserscan.py:1151)
}

template <typename T>
__global__ __launch_bounds__(256) void k_pack_count(const T* __restrict__ pts, const int* __restrict__ label,
                                                    const int* __restrict__ index, int n,
                                                    int* __restrict__ blockcount) {
  __shared__ int wcnt[4];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool keep = i < n && pack_keep<T>(pts, label, index, i);
  const unsigned long long m = __ballot(keep);
  if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) blockcount[blockIdx.x] = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
}

__global__ __launch_bounds__(1024) void k_pack_scan(int* __restrict__ blockcount, int nblocks) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int carry = 0;
  for (int base = 0; base < nblocks; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < nblocks ? blockcount[i] : 0;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    if (i < nblocks) blockcount[i] = carry + woff + inc - v;
    if (threadIdx.x == 1023) carry_s = carry + woff + inc;
    __syncthreads();
    carry = carry_s;
  }
  if (threadIdx.x == 0) blockcount[nblocks] = carry;
}

template <typename T>
__global__ __launch_bounds__(256) void k_pack_write(const T* __restrict__ pts, const float* __restrict__ rem,
                                                    const int* __restrict__ label, const int* __restrict__ index,
                                                    int n, const int* __restrict__ blockoff,
                                                    float4* __restrict__ out_bin, unsigned* __restrict__ out_label) {
  __shared__ int wcnt[4];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool keep = i < n && pack_keep<T>(pts, label, index, i);
  const unsigned long long m = __ballot(keep);
  if (lane == 0) wcnt[wave] = __popcll(m);
  __syncthreads();
  if (!keep) return;
  int k = blockoff[blockIdx.x] + __popcll(m & ((1ull << lane) - 1ull));
  for (int w = 0; w < wave; ++w) k += wcnt[w];
  // struct.pack("ffff", x, y, z, remission): double -> float conversion rounds to nearest, as C does
  out_bin[k] = make_float4((float)pts[3 * (size_t)i], (float)pts[3 * (size_t)i + 1], (float)pts[3 * (size_t)i + 2],
                           rem[i]);
  out_label[k] = (unsigned)label[i];
}

namespace {
std::mutex g_post_mu;
int* g_blockcount = nullptr;
size_t g_blockcount_cap = 0;
int g_post_dev = -1;
}  // namespace

extern "C" int lt_pack_scan_dev(const void* points, int is_f64, const float* rem, const int* label,
                                const int* index, int n, float* out_bin, unsigned* out_label, int* n_out,
                                void* stream) {
  if (n < 0 || (n > 0 && (!points || !rem || !label || !out_bin || !out_label))) {
    lt_set_error("lt_pack_scan_dev: invalid argument (n=%d)", n);
    return LT_ERR_INVALID_ARG;
  }
  if (n_out) *n_out = 0;
  if (n == 0) return LT_OK;
  std::lock_guard<std::mutex> lock(g_post_mu);
  int dev = 0;
  LT_HIP(hipGetDevice(&dev));
  const int nb = (n + 255) / 256;
  if (dev != g_post_dev || (size_t)nb + 2 > g_blockcount_cap) {
    if (g_blockcount) {
      LT_HIP(hipDeviceSynchronize());
      (void)hipFree(g_blockcount);
      g_blockcount = nullptr;
    }
    g_blockcount_cap = (size_t)nb * 2 + 1024;
    LT_HIP(hipMalloc((void**)&g_blockcount, g_blockcount_cap * sizeof(int)));
    g_post_dev = dev;
  }
  hipStream_t st = (hipStream_t)stream;
  if (is_f64) {
    hipLaunchKernelGGL(k_pack_count<double>, dim3(nb), dim3(256), 0, st, (const double*)points, label, index, n,
                       g_blockcount);
    hipLaunchKernelGGL(k_pack_scan, dim3(1), dim3(1024), 0, st, g_blockcount, nb);
    hipLaunchKernelGGL(k_pack_write<double>, dim3(nb), dim3(256), 0, st, (const double*)points, rem, label, index, n,
                       (const int*)g_blockcount, (float4*)out_bin, out_label);
  } else {
    hipLaunchKernelGGL(k_pack_count<float>, dim3(nb), dim3(256), 0, st, (const float*)points, label, index, n,
                       g_blockcount);
    hipLaunchKernelGGL(k_pack_scan, dim3(1), dim3(1024), 0, st, g_blockcount, nb);
    hipLaunchKernelGGL(k_pack_write<float>, dim3(nb), dim3(256), 0, st, (const float*)points, rem, label, index, n,
                       (const int*)g_blockcount, (float4*)out_bin, out_label);
  }
  LT_HIP(hipGetLastError());
  int kept = 0;
  LT_HIP(hipMemcpyAsync(&kept, g_blockcount + nb, sizeof(int), hipMemcpyDeviceToHost, st));
  LT_HIP(hipStreamSynchronize(st));
  if (n_out) *n_out = kept;
  return LT_OK;
}

// ---- compare: masked confusion matrix + squared differences ----------------------------------------------------
// Masks as laserscan.py:1200-1210: cells whose SOURCE colour is black, or whose source label is 0, are
// background in both images.  conf[pred = target][gt = source] += 1 over raw label values < n_labels
// (np_ioueval.py:40-47: rows = predictions, columns = ground truth).
__global__ __launch_bounds__(256) void k_compare(const int* __restrict__ src_label, const float* __restrict__ src_color,
                                                 const int* __restrict__ tgt_label, const float* __restrict__ src_range,
                                                 const float* __restrict__ tgt_range, const float* __restrict__ src_rem,
                                                 const float* __restrict__ tgt_rem, int n, int n_labels,
                                                 unsigned long long* __restrict__ conf, float* __restrict__ range_diff,
                                                 float* __restrict__ rem_diff, int* __restrict__ src_masked,
                                                 int* __restrict__ tgt_masked, double* __restrict__ sq_sum) {
  __shared__ double red[4];
  const int i = blockIdx.x * 256 + threadIdx.x;
  double sq = 0.0;
  bool counted = false;
  unsigned pair = 0u;
  if (i < n) {
    const float csum = (src_color[3 * (size_t)i] + src_color[3 * (size_t)i + 1]) + src_color[3 * (size_t)i + 2];
    int sl = src_label[i], tl = tgt_label[i];
    if (csum == 0.f) { sl = 0; tl = 0; }
    const bool bg = sl == 0;
    if (bg) tl = 0;
    if (src_masked) src_masked[i] = sl;
    if (tgt_masked) tgt_masked[i] = tl;
    counted = (unsigned)sl < (unsigned)n_labels && (unsigned)tl < (unsigned)n_labels;
    pair = counted ? (unsigned)tl * (unsigned)n_labels + (unsigned)sl : 0u;
    const float sr = bg ? 0.f : src_range[i], tr = bg ? 0.f : tgt_range[i];
    const float d = sr - tr;
    const float d2 = d * d;
    if (range_diff) range_diff[i] = d2;
    sq = (double)d2;
    if (rem_diff) {
      const float a = bg ? 0.f : src_rem[i], b = bg ? 0.f : tgt_rem[i];
      rem_diff[i] = (a - b) * (a - b);
    }
  }
  // the confusion matrix: an image holds a handful of label pairs, so an atomic per cell is tens of thousands of memory-side
  // atomics in a row on the same few words (~12 ns each).  The lanes of a wave that hold the same pair count themselves with
  // ballots; one atomic per distinct pair and wave.
  for (unsigned long long rest = __ballot(counted); rest;) {
    const int leader = __ffsll((long long)rest) - 1;
    const unsigned k = (unsigned)__shfl((int)pair, leader, 64);
    const unsigned long long same = __ballot(counted && pair == k);
    if ((int)(threadIdx.x & 63) == leader) atomicAdd(&conf[k], (unsigned long long)__popcll(same));
    rest &= ~same;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(sq_sum, (red[0] + red[1]) + (red[2] + red[3]));
}

extern "C" int lt_compare_dev(const int* src_label, const float* src_color, const int* tgt_label,
                              const float* src_range, const float* tgt_range, const float* src_rem,
                              const float* tgt_rem, int n, int n_labels, unsigned long long* conf,
                              float* range_diff, float* rem_diff, int* src_masked, int* tgt_masked,
                              double* sq_sum, void* stream) {
  if (n < 0 || n_labels <= 0 || !src_label || !src_color || !tgt_label || !src_range || !tgt_range || !conf ||
      !sq_sum || (rem_diff && (!src_rem || !tgt_rem))) {
    lt_set_error("lt_compare_dev: invalid argument");
    return LT_ERR_INVALID_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  LT_HIP(hipMemsetAsync(conf, 0, (size_t)n_labels * n_labels * sizeof(unsigned long long), st));
  LT_HIP(hipMemsetAsync(sq_sum, 0, sizeof(double), st));
  if (n > 0)
    hipLaunchKernelGGL(k_compare, dim3((n + 255) / 256), dim3(256), 0, st, src_label, src_color, tgt_label, src_range,
                       tgt_range, src_rem, tgt_rem, n, n_labels, conf, range_diff, rem_diff, src_masked, tgt_masked,
                       sq_sum);
  LT_HIP(hipGetLastError());
  return LT_OK;
}
