// lt_project.hip -- point cloud -> spherical range image with atomic z-min on gfx950.
//
// Replaces the numpy / pure-Python projections of the reference's scan model:
//   LaserScan.do_range_projection        auxiliary/laserscan.py:202-292  (argsort by depth, scatter)
//   LaserScan.do_range_projection_new    auxiliary/laserscan.py:294-391  (per-point Python z-min loop)
//   SemLaserScan.do_label_projection[_new]  auxiliary/laserscan.py:645-649, :672-676
// and create_rays (auxiliary/laserscan.py:1092-1119) as a device kernel.
//
// Semantics: the closest point of a cell wins; among equal depths the LOWEST point index wins --
// exactly the `_new` loop (`depth[i] < range_image[...]`, laserscan.py:376) and one valid outcome of
// the unstable argsort of the old variant.  Implemented as two order-independent atomic passes:
// atomicMin of the depth bits per cell (positive IEEE doubles order like unsigned integers), then
// atomicMin of the point index among the points that equal the cell minimum.
//
// Arithmetic follows numpy's dtype rules for the array the reference holds: float32 when the scan
// was read from file (laserscan.py:131-136), float64 after a pose was applied (laserscan.py:98-104).
// float32 transcendental results are the correctly rounded value (computed in double, rounded once).
#include "lt_internal.h"
#include <math.h>
#include <string.h>
#include <algorithm>
#include <mutex>
#include <new>

#define LT_EMPTY_IDX 0x7F7F7F7F  // hipMemset(0x7F) pattern: larger than any point index

template <typename T>
struct proj_out {
  int cell;   // py * W + px, or -1 when the point is dropped
  T depth, xf, yf;
  int px, py;
};

__device__ __forceinline__ float lt_atan2(float y, float x) { return (float)atan2((double)y, (double)x); }
__device__ __forceinline__ double lt_atan2(double y, double x) { return atan2(y, x); }
__device__ __forceinline__ float lt_asin(float q) { return (float)asin((double)q); }
__device__ __forceinline__ double lt_asin(double q) { return asin(q); }
__device__ __forceinline__ float lt_sqrt(float v) { return sqrtf(v); }
__device__ __forceinline__ double lt_sqrt(double v) { return sqrt(v); }
__device__ __forceinline__ float lt_floor(float v) { return floorf(v); }
__device__ __forceinline__ double lt_floor(double v) { return floor(v); }

// one point through laserscan.py:214-262 (resp. :304-351); all constants pre-rounded to T by the host
template <typename T>
__device__ __forceinline__ proj_out<T> project_point(T x, T y, T z, T pi_t, T abs_fov_down, T fov, int H, int W,
                                                     const double* __restrict__ beams, int n_beams,
                                                     bool drop_zero, bool drop_outside) {
  proj_out<T> o;
  const T depth = lt_sqrt((x * x + y * y) + z * z);  // np.linalg.norm(points, 2, axis=1)
  T yaw = -lt_atan2(y, x);
  T pitch = lt_asin(z / depth);
  if (n_beams > 0) {  // nearest hard-coded beam angle, first minimum (laserscan.py:233-238)
    double best = fabs((double)pitch - beams[0]);
    int bi = 0;
    for (int k = 1; k < n_beams; ++k) {
      const double dlt = fabs((double)pitch - beams[k]);
      if (dlt < best) { best = dlt; bi = k; }
    }
    pitch = (T)beams[bi];
  }
  T px = (T)0.5 * (yaw / pi_t + (T)1.0);
  T py = (T)1.0 - (pitch + abs_fov_down) / fov;
  bool keep = true;
  if (drop_zero && depth == (T)0) keep = false;
  if (drop_outside && !(py >= (T)0 && py <= (T)1)) keep = false;
  if (!(depth == depth) || !(px == px) || !(py == py)) keep = false;  // NaN never reaches an image
  px *= (T)W;
  py *= (T)H;
  o.xf = px;
  o.yf = py;
  T fx = lt_floor(px), fy = lt_floor(py);
  fx = fx < (T)(W - 1) ? fx : (T)(W - 1);
  fx = fx > (T)0 ? fx : (T)0;
  fy = fy < (T)(H - 1) ? fy : (T)(H - 1);
  fy = fy > (T)0 ? fy : (T)0;
  o.px = (int)fx;
  o.py = (int)fy;
  o.depth = depth;
  o.cell = keep ? o.py * W + o.px : -1;
  return o;
}

// pass 1: project, z-min of the depth bits, per-workgroup count of kept points
template <typename T>
__global__ __launch_bounds__(256) void k_project(const T* __restrict__ pts, int n, T pi_t, T abs_fov_down, T fov,
                                                 int H, int W, const double* __restrict__ beams, int n_beams,
                                                 int drop_zero, int drop_outside, int round_key,
                                                 int* __restrict__ cell,
                                                 double* __restrict__ depth_d, T* __restrict__ xf,
                                                 T* __restrict__ yf, unsigned long long* __restrict__ cellmin,
                                                 int* __restrict__ blockcount) {
  __shared__ int wcnt[4];
  const int i = blockIdx.x * 256 + threadIdx.x;
  bool keep = false;
  if (i < n) {
    const proj_out<T> o = project_point<T>(pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2], pi_t,
                                           abs_fov_down, fov, H, W, beams, n_beams, drop_zero, drop_outside);
    cell[i] = o.cell;
    depth_d[i] = (double)o.depth;
    xf[i] = o.xf;
    yf[i] = o.yf;
    keep = o.cell >= 0;
    // `_new` keeps its running minimum in a float32 image (laserscan.py:366, :376): the cell minimum is
    // taken over the float32-ROUNDED depths there
    const double key = round_key ? (double)(float)o.depth : (double)o.depth;
    if (keep) atomicMin(&cellmin[o.cell], (unsigned long long)__double_as_longlong(key));
  }
  const unsigned long long m = __ballot(keep);
  if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) blockcount[blockIdx.x] = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
}

// pass 2: compacted index of every kept point (stable), compacted per-point outputs, index z-min.
// Every workgroup sums the kept-counts of the workgroups before it itself (n / 256 integers from L2: cheaper
// than a scan kernel between the two passes); the last one also leaves the total in blockcount[gridDim.x].
template <typename T>
__global__ __launch_bounds__(256) void k_assign(const T* __restrict__ pts, const float* __restrict__ rem,
                                                const unsigned* __restrict__ label, int n, int W,
                                                const int* __restrict__ cell, const double* __restrict__ depth_d,
                                                const T* __restrict__ xf, const T* __restrict__ yf,
                                                const unsigned long long* __restrict__ cellmin,
                                                int* __restrict__ blockcount, int round_key,
                                                int* __restrict__ idxmin, int* __restrict__ idxlast,
                                                int* __restrict__ orig_of, T* __restrict__ pts_k,
                                                float* __restrict__ rem_k, unsigned* __restrict__ label_k,
                                                T* __restrict__ depth_k, int* __restrict__ px_k,
                                                int* __restrict__ py_k, T* __restrict__ xf_k, T* __restrict__ yf_k) {
  __shared__ int wcnt[4], wbefore[4];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = i < n ? cell[i] : -1;
  const bool keep = c >= 0;
  const unsigned long long m = __ballot(keep);
  int before = 0;
  for (int b = threadIdx.x; b < (int)blockIdx.x; b += 256) before += blockcount[b];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o, 64);
  if (lane == 0) { wcnt[wave] = __popcll(m); wbefore[wave] = before; }
  __syncthreads();
  int off = (wbefore[0] + wbefore[1]) + (wbefore[2] + wbefore[3]);
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0)
    blockcount[gridDim.x] = off + (wcnt[0] + wcnt[1]) + (wcnt[2] + wcnt[3]);  // number of kept points
  for (int w = 0; w < wave; ++w) off += wcnt[w];
  if (!keep) return;
  const int k = off + __popcll(m & ((1ull << lane) - 1ull));
  orig_of[k] = i;
  const double d = depth_d[i];
  if (pts_k) {
    pts_k[3 * (size_t)k] = pts[3 * (size_t)i];
    pts_k[3 * (size_t)k + 1] = pts[3 * (size_t)i + 1];
    pts_k[3 * (size_t)k + 2] = pts[3 * (size_t)i + 2];
  }
  if (rem_k && rem) rem_k[k] = rem[i];
  if (label_k && label) label_k[k] = label[i];
  if (depth_k) depth_k[k] = (T)d;
  if (px_k) px_k[k] = c % W;
  if (py_k) py_k[k] = c / W;
  if (xf_k) xf_k[k] = xf[i];
  if (yf_k) yf_k[k] = yf[i];
  const double key = round_key ? (double)(float)d : d;
  const unsigned long long mbits = cellmin[c];
  if ((unsigned long long)__double_as_longlong(key) == mbits) {
    atomicMin(&idxmin[c], k);
    // `depth[i] < range_image[cell]` compares a float64 depth with the float32-rounded minimum: every
    // later point of the minimum's float32 bucket that rounded UP replaces the incumbent (see k_resolve)
    if (round_key && d < __longlong_as_double((long long)mbits)) atomicMax(&idxlast[c], k);
  }
}

// pass 3: one thread per cell gathers the winner (laserscan.py:283-292, :376-382, :645-649) and re-arms the
// cell's three z-min words for the next projection (no memsets between calls)
template <typename T>
__global__ __launch_bounds__(256) void k_resolve(const T* __restrict__ pts, const float* __restrict__ rem,
                                                 const unsigned* __restrict__ label, const double* __restrict__ depth_d,
                                                 unsigned long long* __restrict__ cellmin,
                                                 int* __restrict__ idxmin, int* __restrict__ idxlast,
                                                 const int* __restrict__ orig_of, int ncells, const float* __restrict__ lut, int lut_len,
                                                 float range_init, float rem_init, float xyz_init,
                                                 int* __restrict__ idx_img, float* __restrict__ range_img,
                                                 float* __restrict__ xyz_img, float* __restrict__ rem_img,
                                                 int* __restrict__ label_img, float* __restrict__ color_img,
                                                 float* __restrict__ mask_img) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= ncells) return;
  // winner of the reference's sequential loop: the first point of the minimum bucket, unless later points
  // of that bucket lie below the float32 minimum -- then the last of those (idxlast is -1 otherwise and
  // always for float32 input, where rounding is the identity)
  const int first = idxmin[c], last = idxlast[c];
  cellmin[c] = ~0ull;
  idxmin[c] = LT_EMPTY_IDX;
  idxlast[c] = -1;
  const int k = (last >= 0 && last != first) ? last : first;
  const bool has = first != LT_EMPTY_IDX;
  const int i = has ? orig_of[k] : 0;
  if (idx_img) idx_img[c] = has ? k : -1;
  if (range_img) range_img[c] = has ? (float)depth_d[i] : range_init;
  if (xyz_img) {
    xyz_img[3 * (size_t)c] = has ? (float)pts[3 * (size_t)i] : xyz_init;
    xyz_img[3 * (size_t)c + 1] = has ? (float)pts[3 * (size_t)i + 1] : xyz_init;
    xyz_img[3 * (size_t)c + 2] = has ? (float)pts[3 * (size_t)i + 2] : xyz_init;
  }
  if (rem_img) rem_img[c] = (has && rem) ? rem[i] : rem_init;
  const unsigned lab = (has && label) ? label[i] : 0u;
  if (label_img) label_img[c] = (int)lab;
  if (color_img) {
    const bool ok = has && lut && lab < (unsigned)lut_len;
    color_img[3 * (size_t)c] = ok ? lut[3 * (size_t)lab] : 0.f;
    color_img[3 * (size_t)c + 1] = ok ? lut[3 * (size_t)lab + 1] : 0.f;
    color_img[3 * (size_t)c + 2] = ok ? lut[3 * (size_t)lab + 2] : 0.f;
  }
  if (mask_img) mask_img[c] = (has && k > 0) ? 1.f : 0.f;  // proj_idx > 0 (sic, laserscan.py:292)
}

// ---- create_rays (laserscan.py:1092-1119): float64 trigonometry, cast to float32 last ---------------------
__global__ __launch_bounds__(256) void k_create_rays(double fov_up, double fov_down, int H, int W,
                                                     float* __restrict__ rays) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= H * W) return;
  const int h = idx / W, w = idx - h * W;
  // np.linspace(a, b, n)[i] = a + i * ((b - a) / (n - 1)), last element forced to b
  double yaw_deg = W > 1 ? (w == W - 1 ? 360.0 : 0.0 + w * (360.0 / (W - 1))) : 0.0;
  yaw_deg += 180.0;
  if (yaw_deg > 360.0) yaw_deg -= 360.0;
  const double yaw = yaw_deg / 180. * M_PI;
  double pd = H > 1 ? (h == H - 1 ? fov_down : fov_up + h * ((fov_down - fov_up) / (H - 1))) : fov_up;
  const double p = M_PI / 2 - pd / 180. * M_PI;
  const double sp = sin(p);
  rays[3 * (size_t)idx] = (float)(sp * cos(-yaw));
  rays[3 * (size_t)idx + 1] = (float)(sp * sin(-yaw));
  rays[3 * (size_t)idx + 2] = (float)(cos(p) * 1.0);
}

extern "C" int lt_create_rays_dev(double fov_up, double fov_down, int H, int W, float* rays, void* stream) {
  if (H <= 0 || W <= 0 || !rays) {
    lt_set_error("lt_create_rays_dev: invalid argument (H=%d W=%d)", H, W);
    return LT_ERR_INVALID_ARG;
  }
  hipLaunchKernelGGL(k_create_rays, dim3((H * W + 255) / 256), dim3(256), 0, (hipStream_t)stream, fov_up, fov_down,
                     H, W, rays);
  LT_HIP(hipGetLastError());
  return LT_OK;
}

// ---- host orchestration -------------------------------------------------------------------------------------
namespace {
struct proj_ws {
  int device = -1;
  size_t cap_n = 0, cap_cells = 0;
  int* cell = nullptr;
  double* depth_d = nullptr;
  void* xf = nullptr;
  void* yf = nullptr;
  int* blockcount = nullptr;
  int* orig_of = nullptr;
  unsigned long long* cellmin = nullptr;
  int* idxmin = nullptr;
  int* idxlast = nullptr;
  double* beams = nullptr;
  bool armed = false;  // cellmin / idxmin / idxlast hold their empty patterns (k_resolve leaves them so)
};
std::mutex g_pmu;
proj_ws g_pws;

void pws_free(proj_ws& w) {
  void* ps[] = {w.cell, w.depth_d, w.xf, w.yf, w.blockcount, w.orig_of, w.cellmin, w.idxmin, w.idxlast, w.beams};
  for (void* p : ps)
    if (p) (void)hipFree(p);
  w = proj_ws();
}

int pws_reserve(proj_ws& w, int device, size_t n, size_t cells) {
  if (w.device == device && n <= w.cap_n && cells <= w.cap_cells) return LT_OK;
  if (w.device >= 0) {
    LT_HIP(hipDeviceSynchronize());
    pws_free(w);
  }
  const size_t cn = n + n / 4 + 1024, cc = cells + 1024;
  LT_HIP(hipMalloc((void**)&w.cell, cn * sizeof(int)));
  LT_HIP(hipMalloc((void**)&w.depth_d, cn * sizeof(double)));
  LT_HIP(hipMalloc(&w.xf, cn * sizeof(double)));
  LT_HIP(hipMalloc(&w.yf, cn * sizeof(double)));
  LT_HIP(hipMalloc((void**)&w.blockcount, (cn / 256 + 2) * sizeof(int)));
  LT_HIP(hipMalloc((void**)&w.orig_of, cn * sizeof(int)));
  LT_HIP(hipMalloc((void**)&w.cellmin, cc * sizeof(unsigned long long)));
  LT_HIP(hipMalloc((void**)&w.idxmin, cc * sizeof(int)));
  LT_HIP(hipMalloc((void**)&w.idxlast, cc * sizeof(int)));
  LT_HIP(hipMalloc((void**)&w.beams, 1024 * sizeof(double)));
  w.device = device;
  w.cap_n = cn;
  w.cap_cells = cc;
  return LT_OK;
}

template <typename T>
int run_projection(proj_ws& w, const T* pts, const float* rem, const unsigned* label, int n, double fov_up_deg,
                   double fov_down_deg, int H, int W, int n_beams, unsigned flags, const float* lut, int lut_len,
                   T* pts_k, float* rem_k, unsigned* label_k, T* depth_k, int* px_k, int* py_k, T* xf_k, T* yf_k,
                   int* idx_img, float* range_img, float* xyz_img, float* rem_img, int* label_img, float* color_img,
                   float* mask_img, float range_init, float rem_init, float xyz_init, int* n_kept, hipStream_t st) {
  // laser parameters exactly as laserscan.py:207-209 (python floats), then rounded once to the array dtype
  const double fu = fov_up_deg / 180.0 * M_PI, fd = fov_down_deg / 180.0 * M_PI;
  const double fov = fabs(fd) + fabs(fu);
  const int cells = H * W;
  const int nb = (n + 255) / 256;
  if (!w.armed) {  // first use of the workspace, or the previous call failed half way
    LT_HIP(hipMemsetAsync(w.cellmin, 0xFF, w.cap_cells * sizeof(unsigned long long), st));
    LT_HIP(hipMemsetAsync(w.idxmin, 0x7F, w.cap_cells * sizeof(int), st));
    LT_HIP(hipMemsetAsync(w.idxlast, 0xFF, w.cap_cells * sizeof(int), st));
  }
  w.armed = false;
  const int round_key = (flags & LT_PROJ_NEW) ? 1 : 0;
  if (n > 0) {
    hipLaunchKernelGGL(k_project<T>, dim3(nb), dim3(256), 0, st, pts, n, (T)M_PI, (T)fabs(fd), (T)fov, H, W,
                       (const double*)w.beams, n_beams, (flags & (LT_PROJ_REMOVE | LT_PROJ_NEW)) ? 1 : 0,
                       (flags & LT_PROJ_REMOVE) ? 1 : 0, round_key, w.cell, w.depth_d, (T*)w.xf, (T*)w.yf, w.cellmin,
                       w.blockcount);
    hipLaunchKernelGGL(k_assign<T>, dim3(nb), dim3(256), 0, st, pts, rem, label, n, W, (const int*)w.cell,
                       (const double*)w.depth_d, (const T*)w.xf, (const T*)w.yf,
                       (const unsigned long long*)w.cellmin, w.blockcount, round_key, w.idxmin,
                       w.idxlast, w.orig_of, pts_k,
                       rem_k, label_k, depth_k, px_k, py_k, xf_k, yf_k);
  }
  hipLaunchKernelGGL(k_resolve<T>, dim3((cells + 255) / 256), dim3(256), 0, st, pts, rem, label,
                     (const double*)w.depth_d, w.cellmin, w.idxmin, w.idxlast, (const int*)w.orig_of,
                     cells, lut, lut_len,
                     range_init, rem_init, xyz_init, idx_img, range_img, xyz_img, rem_img, label_img, color_img,
                     mask_img);
  LT_HIP(hipGetLastError());
  int kept = 0;
  if (n > 0) LT_HIP(hipMemcpyAsync(&kept, w.blockcount + nb, sizeof(int), hipMemcpyDeviceToHost, st));
  LT_HIP(hipStreamSynchronize(st));
  w.armed = true;
  if (n_kept) *n_kept = kept;
  return LT_OK;
}
}  // namespace

extern "C" int lt_range_projection_dev(const void* points, int is_f64, const float* rem, const unsigned* label,
                                       int n, double fov_up, double fov_down, int H, int W,
                                       const double* beam_angles, int n_beams, unsigned flags,
                                       const float* color_lut, int lut_len, void* points_kept, float* rem_kept,
                                       unsigned* label_kept, void* depth_kept, int* proj_x_kept, int* proj_y_kept,
                                       void* proj_xf_kept, void* proj_yf_kept, int* idx_img, float* range_img,
                                       float* xyz_img, float* rem_img, int* label_img, float* color_img,
                                       float* mask_img, float range_init, float rem_init, float xyz_init,
                                       int* n_kept, void* stream) {
  if (n < 0 || H <= 0 || W <= 0 || (n > 0 && !points) || n_beams < 0 || n_beams > 1024 ||
      (n_beams > 0 && !beam_angles)) {
    lt_set_error("lt_range_projection: invalid argument (n=%d H=%d W=%d n_beams=%d)", n, H, W, n_beams);
    return LT_ERR_INVALID_ARG;
  }
  std::lock_guard<std::mutex> lock(g_pmu);
  int dev = 0;
  LT_HIP(hipGetDevice(&dev));
  LT_CHECK(pws_reserve(g_pws, dev, (size_t)n, (size_t)H * W));
  hipStream_t st = (hipStream_t)stream;
  if (n_beams > 0)
    LT_HIP(hipMemcpyAsync(g_pws.beams, beam_angles, n_beams * sizeof(double), hipMemcpyHostToDevice, st));
  if (is_f64)
    return run_projection<double>(g_pws, (const double*)points, rem, label, n, fov_up, fov_down, H, W, n_beams, flags,
                                  color_lut, lut_len, (double*)points_kept, rem_kept, label_kept, (double*)depth_kept,
                                  proj_x_kept, proj_y_kept, (double*)proj_xf_kept, (double*)proj_yf_kept, idx_img,
                                  range_img, xyz_img, rem_img, label_img, color_img, mask_img, range_init, rem_init,
                                  xyz_init, n_kept, st);
  return run_projection<float>(g_pws, (const float*)points, rem, label, n, fov_up, fov_down, H, W, n_beams, flags,
                               color_lut, lut_len, (float*)points_kept, rem_kept, label_kept, (float*)depth_kept,
                               proj_x_kept, proj_y_kept, (float*)proj_xf_kept, (float*)proj_yf_kept, idx_img,
                               range_img, xyz_img, rem_img, label_img, color_img, mask_img, range_init, rem_init,
                               xyz_init, n_kept, st);
}

// ---- batched, synchronisation-free projection (lt_range_projection_batch_dev) ------------------------------------------
// The `number_of_scans` clouds of one output scan (laserscan.py:874-881: do_range_projection_new + do_label_projection_new
// per scan) in ONE launch sequence on the caller's stream, nothing read back by the host:
//   k_pb_project   a thread per point of every cloud: the reference's per-point expressions (project_point, above) and ONE
//                  64-bit atomicMin per kept point -- no per-point temporaries are written (the single-cloud call stores
//                  cell / depth / xf / yf per point: 28 B written and read again per point);
//   [k_pb_assign]  only for the OLD variant on float64 clouds, whose order key (a float64 depth) leaves no room for the index
//   [k_pb_prefix]  only when an output needs the numbering of the KEPT points (idx / mask images, proj_x .. images, n_kept)
//   k_pb_resolve   a thread per cell of every image: decodes the winner, RE-COMPUTES its projection from the point itself
//                  (the same deterministic function), gathers remission / label / colour, writes the images and re-arms the
//                  cell's key for the next call (no memsets between calls).
// The key.  `_new` keeps its running minimum in a float32 image and replaces when `depth[i] < image` (laserscan.py:366, :376):
// the winner is the first point of the minimum's float32 bucket unless points of that bucket lie BELOW the float32 value
// (rounded up) -- then the last of those.  hi word = float32 bits of the depth (positive floats order like unsigned
// integers); lo word = 0x7fffffff - index for a rounded-up point (they beat the others, the highest index first),
// 0x80000000 | index otherwise (lowest index first): one atomicMin yields exactly that point.  float32 clouds never round.
// The OLD variant on float32 clouds: closest point, lowest index among equal depths (k_project's rule) -- the same key.
#define LT_PB_MAX 8
#define LT_PB_EMPTY (~0ull)

struct pb_cloud {
  const void* pts; const float* rem; const unsigned* label;
  int n, block0;                 // points; first workgroup of this cloud in the per-point grids
  unsigned long long* key;       // [cells] z-min key, LT_PB_EMPTY when no point fell into the cell
  unsigned long long* dmin;      // [cells] float64 depth bits (OLD variant on float64 clouds), else unused
  unsigned long long* keep;      // [ceil(n / 64)] kept-point mask per wave of k_pb_project
  int* wprefix;                  // [ceil(n / 64)] kept points before the wave (k_pb_prefix)
  int* meta;                     // [2] number of kept points, original index of the last kept point (-1: none)
  int* idx_img; float* range_img; float* xyz_img; float* rem_img; int* label_img; float* color_img; float* mask_img;
  float* fold_img; int* px_img; int* py_img; void* xf_img; void* yf_img; int* n_kept;
  unsigned long long* bacc;      // [blocks of the cloud][6] per-WORKGROUP order-preserving keys of the kept points' bounds (min x,
                                 // max x, min y, ...), written by k_pb_project, folded by k_pb_prefix / k_pb_bnds; NULL: not wanted
  double* bnds_out;              // [6] get_bnds() of the kept points (laserscan.py:678-681), written by k_pb_prefix
};
struct pb_args { pb_cloud c[LT_PB_MAX]; int n_clouds; };

__device__ __forceinline__ int pb_find_cloud(const pb_args& A, int block) {
  int c = 0;
#pragma unroll
  for (int k = 1; k < LT_PB_MAX; ++k)
    if (k < A.n_clouds && block >= A.c[k].block0) c = k;
  return c;
}

template <typename T>
__device__ __forceinline__ unsigned long long pb_key(T depth, int i) {
  const float df = (float)depth;
  const bool up = (double)depth < (double)df;  // lies below its float32 value: replaces an incumbent of the same bucket
  const unsigned lo = up ? (0x7fffffffu - (unsigned)i) : (0x80000000u | (unsigned)i);
  return ((unsigned long long)__float_as_uint(df) << 32) | lo;
}
__device__ __forceinline__ int pb_key_index(unsigned long long k) {
  const unsigned lo = (unsigned)k;
  return (lo & 0x80000000u) ? (int)(lo & 0x7fffffffu) : (int)(0x7fffffffu - lo);
}

// order-preserving map double -> uint64 (and back): unsigned comparison of the keys == comparison of the values
__device__ __forceinline__ unsigned long long pb_ord(double d) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(d);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double pb_unord(unsigned long long k) {
  return __longlong_as_double((long long)((k >> 63) ? (k & 0x7fffffffffffffffull) : ~k));
}

// MODE 0: the single-key variants (NEW on any dtype, OLD on float32).  MODE 1: OLD on float64 -- depth minimum only.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void k_pb_project(pb_args A, T pi_t, T abs_fov_down, T fov, int H, int W,
                                                    const double* __restrict__ beams, int n_beams, int drop_zero,
                                                    int drop_outside) {
  const int ci = pb_find_cloud(A, blockIdx.x);
  const pb_cloud& c = A.c[ci];
  const int i = (blockIdx.x - c.block0) * 256 + threadIdx.x;
  bool keep = false;
  if (i < c.n) {
    const T* pts = (const T*)c.pts;
    const proj_out<T> o = project_point<T>(pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2], pi_t,
                                           abs_fov_down, fov, H, W, beams, n_beams, drop_zero, drop_outside);
    keep = o.cell >= 0;
    if (keep) {
      if (MODE == 0) atomicMin(&c.key[o.cell], pb_key<T>(o.depth, i));
      else atomicMin(&c.dmin[o.cell], (unsigned long long)__double_as_longlong((double)o.depth));
    }
  }
  const unsigned long long m = __ballot(keep);
  if ((threadIdx.x & 63) == 0 && i < c.n) c.keep[i >> 6] = m;
  if (c.bacc) {  // bounds of the kept points: SemLaserScan.get_bnds after remove_points (laserscan.py:678-681)
    // Per WORKGROUP, no atomics: six shared words for the whole cloud were 2 000 memory-side atomics in a row per word (150 us
    // for a 130 k-point cloud), and even LOOKING at them first (a load per wave on six hot addresses) cost 25 us.
    __shared__ unsigned long long s_b[4][6];
    const T* pts = (const T*)c.pts;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double q = keep ? (double)pts[3 * (size_t)i + a] : 0.0;
      unsigned long long lo = keep ? pb_ord(q) : ~0ull, hi = keep ? pb_ord(q) : 0ull;
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) {
        const unsigned long long l2 = __shfl_xor(lo, o), h2 = __shfl_xor(hi, o);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
      }
      if ((threadIdx.x & 63) == 0) { s_b[threadIdx.x >> 6][2 * a] = lo; s_b[threadIdx.x >> 6][2 * a + 1] = hi; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
      unsigned long long v = s_b[0][threadIdx.x];
      for (int w = 1; w < 4; ++w) {
        const unsigned long long o = s_b[w][threadIdx.x];
        v = (threadIdx.x & 1) ? (o > v ? o : v) : (o < v ? o : v);
      }
      c.bacc[6 * (size_t)(blockIdx.x - c.block0) + threadIdx.x] = v;
    }
  }
}

// fold the workgroups' partial bounds of a cloud (all 64 lanes of one wave help); lane k < 6 writes word k
__device__ __forceinline__ void pb_fold_bounds(const pb_cloud& c, int lane) {
  const int nb = (c.n + 255) >> 8;
  unsigned long long v[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) v[k] = (k & 1) ? 0ull : ~0ull;
  for (int b = lane; b < nb; b += 64)
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const unsigned long long o = c.bacc[6 * (size_t)b + k];
      v[k] = (k & 1) ? (o > v[k] ? o : v[k]) : (o < v[k] ? o : v[k]);
    }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const unsigned long long x = __shfl_xor(v[k], o);
      v[k] = (k & 1) ? (x > v[k] ? x : v[k]) : (x < v[k] ? x : v[k]);
    }
    if (lane == k) {  // no kept point: (+inf, -inf) -- numpy's amin of an empty array raises
      const bool is_min = (k & 1) == 0;
      const bool none = is_min ? v[k] == ~0ull : v[k] == 0ull;
      c.bnds_out[k] = none ? (is_min ? (double)INFINITY : -(double)INFINITY) : pb_unord(v[k]);
    }
  }
}

// OLD variant, float64: among the points that equal the cell's depth minimum the lowest index wins
__global__ __launch_bounds__(256) void k_pb_assign(pb_args A, double pi_t, double abs_fov_down, double fov, int H, int W,
                                                   const double* __restrict__ beams, int n_beams, int drop_zero,
                                                   int drop_outside) {
  const int ci = pb_find_cloud(A, blockIdx.x);
  const pb_cloud& c = A.c[ci];
  const int i = (blockIdx.x - c.block0) * 256 + threadIdx.x;
  if (i >= c.n) return;
  if (!((c.keep[i >> 6] >> (i & 63)) & 1ull)) return;
  const double* pts = (const double*)c.pts;
  const proj_out<double> o = project_point<double>(pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2], pi_t,
                                                   abs_fov_down, fov, H, W, beams, n_beams, drop_zero, drop_outside);
  if ((unsigned long long)__double_as_longlong(o.depth) == c.dmin[o.cell])
    atomicMin(&c.key[o.cell], (unsigned long long)(0x80000000u | (unsigned)i));
}

// one workgroup per cloud: exclusive prefix of the waves' kept counts, the total, the last kept point
__global__ __launch_bounds__(256) void k_pb_prefix(pb_args A) {
  __shared__ int part[256];
  __shared__ int last_s;
  const pb_cloud& c = A.c[blockIdx.x];
  const int nw = (c.n + 63) >> 6;
  const int per = (nw + 255) / 256;
  const int w0 = threadIdx.x * per, w1 = min(nw, w0 + per);
  if (threadIdx.x == 0) last_s = -1;
  int sum = 0, last = -1;
  for (int w = w0; w < w1; ++w) {
    const unsigned long long m = c.keep[w];
    sum += __popcll(m);
    if (m) last = w * 64 + 63 - __clzll((long long)m);
  }
  part[threadIdx.x] = sum;
  __syncthreads();
  if (last >= 0) atomicMax(&last_s, last);
  // Hillis-Steele over 256 partials
  for (int o = 1; o < 256; o <<= 1) {
    const int v = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  int run = part[threadIdx.x] - sum;
  for (int w = w0; w < w1; ++w) {
    c.wprefix[w] = run;
    run += __popcll(c.keep[w]);
  }
  if (threadIdx.x == 255) {
    c.meta[0] = part[255];
    c.meta[1] = last_s;
    if (c.n_kept) *c.n_kept = part[255];
  }
  if (c.bacc && threadIdx.x < 64) pb_fold_bounds(c, threadIdx.x);
}

// the bounds alone (no image wants the kept points' prefix): one thread per word, one workgroup per cloud
__global__ __launch_bounds__(64) void k_pb_bnds(pb_args A) {
  const pb_cloud& c = A.c[blockIdx.x];
  if (c.bacc) pb_fold_bounds(c, threadIdx.x);
}

template <typename T, int MODE>
__global__ __launch_bounds__(256) void k_pb_resolve(pb_args A, int blocks_per_cloud, T pi_t, T abs_fov_down, T fov, int H,
                                                    int W, const double* __restrict__ beams, int n_beams, int drop_zero,
                                                    int drop_outside, const float* __restrict__ lut, int lut_len,
                                                    float range_init, float rem_init, float xyz_init, int have_prefix) {
  const int ci = blockIdx.x / blocks_per_cloud;
  const pb_cloud& c = A.c[ci];
  const int cell = (blockIdx.x - ci * blocks_per_cloud) * 256 + threadIdx.x;
  if (cell >= H * W) return;
  const unsigned long long key = c.key[cell];
  c.key[cell] = LT_PB_EMPTY;
  if (MODE == 1) c.dmin[cell] = LT_PB_EMPTY;
  const bool has = key != LT_PB_EMPTY;
  const T* pts = (const T*)c.pts;
  const bool want_xy = c.px_img || c.py_img || c.xf_img || c.yf_img;
  // an empty cell's pixel coordinates are those of the LAST kept point: numpy's index -1 (laserscan.py:384-388)
  int i = has ? pb_key_index(key) : ((want_xy && have_prefix) ? c.meta[1] : -1);
  T x = (T)0, y = (T)0, z = (T)0;
  proj_out<T> o;
  o.depth = (T)0; o.xf = (T)0; o.yf = (T)0; o.px = 0; o.py = 0; o.cell = -1;
  if (i >= 0) {
    x = pts[3 * (size_t)i]; y = pts[3 * (size_t)i + 1]; z = pts[3 * (size_t)i + 2];
    if (has ? (want_xy || c.range_img) : true)
      o = project_point<T>(x, y, z, pi_t, abs_fov_down, fov, H, W, beams, n_beams, drop_zero, drop_outside);
  }
  int k = -1;
  if (has && have_prefix) k = c.wprefix[i >> 6] + __popcll(c.keep[i >> 6] & ((1ull << (i & 63)) - 1ull));
  if (c.idx_img) c.idx_img[cell] = has ? k : -1;
  if (c.range_img) c.range_img[cell] = has ? (float)o.depth : range_init;
  if (c.xyz_img) {
    c.xyz_img[3 * (size_t)cell] = has ? (float)x : xyz_init;
    c.xyz_img[3 * (size_t)cell + 1] = has ? (float)y : xyz_init;
    c.xyz_img[3 * (size_t)cell + 2] = has ? (float)z : xyz_init;
  }
  if (c.rem_img) c.rem_img[cell] = (has && c.rem) ? c.rem[i] : rem_init;
  const unsigned lab = (has && c.label) ? c.label[i] : 0u;
  if (c.label_img) c.label_img[cell] = (int)lab;
  // the colour image `integrate` is handed, already folded (laserscan.py:893-895: the label in channel 0;
  // fusion_lidar.py:260-264: float32, floor(c0 * 256 * 256 + c1 * 256 + c2))
  if (c.fold_img) c.fold_img[cell] = floorf((float)lab * 256.0f * 256.0f);
  if (c.color_img) {
    const bool ok = has && lut && lab < (unsigned)lut_len;
    c.color_img[3 * (size_t)cell] = ok ? lut[3 * (size_t)lab] : 0.f;
    c.color_img[3 * (size_t)cell + 1] = ok ? lut[3 * (size_t)lab + 1] : 0.f;
    c.color_img[3 * (size_t)cell + 2] = ok ? lut[3 * (size_t)lab + 2] : 0.f;
  }
  if (c.mask_img) c.mask_img[cell] = (has && k > 0) ? 1.f : 0.f;  // proj_idx > 0 (sic, laserscan.py:292)
  if (c.px_img) c.px_img[cell] = o.px;
  if (c.py_img) c.py_img[cell] = o.py;
  if (c.xf_img) ((T*)c.xf_img)[cell] = o.xf;
  if (c.yf_img) ((T*)c.yf_img)[cell] = o.yf;
}

struct lt_projector {
  int device = 0;
  size_t cap_n = 0, cap_cells = 0;          // per cloud slot
  bool armed = false, dmin_armed = false;
  unsigned long long* key[LT_PB_MAX] = {};
  unsigned long long* dmin[LT_PB_MAX] = {};
  unsigned long long* keep[LT_PB_MAX] = {};
  int* wprefix[LT_PB_MAX] = {};
  int* meta = nullptr;                      // [LT_PB_MAX][2]
  double* beams = nullptr;                  // [1024]
  double beams_host[1024];
  int n_beams_cached = -1;
  unsigned long long* bacc = nullptr;       // [LT_PB_MAX][bacc_blocks][6] per-workgroup partial bounds of the clouds (no arming: every
  size_t bacc_blocks = 0;                   // workgroup of k_pb_project writes its six words)
  float* img = nullptr;                     // lt_deform_scan_dev: [n][3][H * W] source images (range, remission, folded label)
  size_t img_cap = 0;                       // floats
  double* mm_bnds = nullptr;                // lt_mergemesh_scan_dev: [6] the kept points' bounds of the scan
  std::mutex mu;
};

namespace {
void pj_free(lt_projector* p) {
  for (int k = 0; k < LT_PB_MAX; ++k) {
    void* ps[] = {p->key[k], p->dmin[k], p->keep[k], p->wprefix[k]};
    for (void* q : ps)
      if (q) (void)hipFree(q);
    p->key[k] = p->dmin[k] = p->keep[k] = nullptr;
    p->wprefix[k] = nullptr;
  }
  p->cap_n = p->cap_cells = 0;
  p->armed = p->dmin_armed = false;
}

int pj_reserve(lt_projector* p, size_t n_max, size_t cells, bool need_dmin, hipStream_t st) {
  if (n_max > p->cap_n || cells > p->cap_cells) {
    if (p->cap_n || p->cap_cells) LT_HIP(hipStreamSynchronize(st));
    const size_t cn = std::max(p->cap_n, n_max + n_max / 4 + 1024), cc = std::max(p->cap_cells, cells + 1024);
    pj_free(p);
    for (int k = 0; k < LT_PB_MAX; ++k) {
      LT_HIP(hipMalloc((void**)&p->key[k], cc * sizeof(unsigned long long)));
      LT_HIP(hipMalloc((void**)&p->keep[k], (cn / 64 + 2) * sizeof(unsigned long long)));
      LT_HIP(hipMalloc((void**)&p->wprefix[k], (cn / 64 + 2) * sizeof(int)));
    }
    p->cap_n = cn;
    p->cap_cells = cc;
  }
  if (need_dmin && !p->dmin[0]) {
    for (int k = 0; k < LT_PB_MAX; ++k) LT_HIP(hipMalloc((void**)&p->dmin[k], p->cap_cells * sizeof(unsigned long long)));
    p->dmin_armed = false;
  }
  if (!p->armed)
    for (int k = 0; k < LT_PB_MAX; ++k)
      LT_HIP(hipMemsetAsync(p->key[k], 0xFF, p->cap_cells * sizeof(unsigned long long), st));
  if (need_dmin && !p->dmin_armed)
    for (int k = 0; k < LT_PB_MAX; ++k)
      LT_HIP(hipMemsetAsync(p->dmin[k], 0xFF, p->cap_cells * sizeof(unsigned long long), st));
  return LT_OK;
}

template <typename T>
int pj_run(lt_projector* p, pb_args& A, int total_blocks, bool old_f64, bool need_prefix, bool bnds_only, double fov_up_deg,
           double fov_down_deg, int H, int W, int n_beams, unsigned flags, const float* lut, int lut_len, float range_init,
           float rem_init, float xyz_init, hipStream_t st) {
  const double fu = fov_up_deg / 180.0 * M_PI, fd = fov_down_deg / 180.0 * M_PI;
  const double fov = fabs(fd) + fabs(fu);
  const int drop_zero = (flags & (LT_PROJ_REMOVE | LT_PROJ_NEW)) ? 1 : 0, drop_outside = (flags & LT_PROJ_REMOVE) ? 1 : 0;
  const int cells = H * W, bpc = (cells + 255) / 256;
  if (total_blocks > 0) {
    if (old_f64) {
      hipLaunchKernelGGL((k_pb_project<T, 1>), dim3(total_blocks), dim3(256), 0, st, A, (T)M_PI, (T)fabs(fd), (T)fov, H, W,
                         (const double*)p->beams, n_beams, drop_zero, drop_outside);
      hipLaunchKernelGGL(k_pb_assign, dim3(total_blocks), dim3(256), 0, st, A, M_PI, fabs(fd), fov, H, W,
                         (const double*)p->beams, n_beams, drop_zero, drop_outside);
    } else {
      hipLaunchKernelGGL((k_pb_project<T, 0>), dim3(total_blocks), dim3(256), 0, st, A, (T)M_PI, (T)fabs(fd), (T)fov, H, W,
                         (const double*)p->beams, n_beams, drop_zero, drop_outside);
    }
  }
  if (need_prefix) hipLaunchKernelGGL(k_pb_prefix, dim3(A.n_clouds), dim3(256), 0, st, A);
  else if (bnds_only) hipLaunchKernelGGL(k_pb_bnds, dim3(A.n_clouds), dim3(64), 0, st, A);
  if (old_f64)
    hipLaunchKernelGGL((k_pb_resolve<T, 1>), dim3(bpc * A.n_clouds), dim3(256), 0, st, A, bpc, (T)M_PI, (T)fabs(fd), (T)fov,
                       H, W, (const double*)p->beams, n_beams, drop_zero, drop_outside, lut, lut_len, range_init, rem_init,
                       xyz_init, need_prefix ? 1 : 0);
  else
    hipLaunchKernelGGL((k_pb_resolve<T, 0>), dim3(bpc * A.n_clouds), dim3(256), 0, st, A, bpc, (T)M_PI, (T)fabs(fd), (T)fov,
                       H, W, (const double*)p->beams, n_beams, drop_zero, drop_outside, lut, lut_len, range_init, rem_init,
                       xyz_init, need_prefix ? 1 : 0);
  LT_HIP(hipGetLastError());
  return LT_OK;
}
}  // namespace

extern "C" int lt_projector_create(lt_projector** pj, int device) {
  if (!pj) {
    lt_set_error("lt_projector_create: NULL handle pointer");
    return LT_ERR_INVALID_ARG;
  }
  *pj = nullptr;
  int dev = device;
  if (dev < 0) LT_HIP(hipGetDevice(&dev));
  LT_HIP(hipSetDevice(dev));
  lt_projector* p = new (std::nothrow) lt_projector();
  if (!p) {
    lt_set_error("lt_projector_create: out of host memory");
    return LT_ERR_NO_MEMORY;
  }
  p->device = dev;
  if (hipMalloc((void**)&p->meta, LT_PB_MAX * 2 * sizeof(int)) != hipSuccess ||
      hipMalloc((void**)&p->beams, 1024 * sizeof(double)) != hipSuccess) {
    if (p->meta) (void)hipFree(p->meta);
    if (p->beams) (void)hipFree(p->beams);
    delete p;
    (void)hipGetLastError();
    lt_set_error("lt_projector_create: out of device memory");
    return LT_ERR_NO_MEMORY;
  }
  *pj = p;
  return LT_OK;
}

extern "C" int lt_projector_destroy(lt_projector* p) {
  if (!p) return LT_OK;
  (void)hipSetDevice(p->device);
  (void)hipDeviceSynchronize();
  pj_free(p);
  if (p->meta) (void)hipFree(p->meta);
  if (p->beams) (void)hipFree(p->beams);
  if (p->bacc) (void)hipFree(p->bacc);
  if (p->mm_bnds) (void)hipFree(p->mm_bnds);
  if (p->img) (void)hipFree(p->img);
  delete p;
  return LT_OK;
}

extern "C" int lt_range_projection_batch_dev(lt_projector* p, int n_clouds, const lt_cloud* clouds, int is_f64,
                                             double fov_up, double fov_down, int H, int W, const double* beam_angles,
                                             int n_beams, unsigned flags, const float* color_lut, int lut_len,
                                             const lt_proj_images* out, float range_init, float rem_init, float xyz_init,
                                             void* stream) {
  if (!p || n_clouds < 0 || (n_clouds > 0 && (!clouds || !out)) || H <= 0 || W <= 0 || n_beams < 0 || n_beams > 1024 ||
      (n_beams > 0 && !beam_angles)) {
    lt_set_error("lt_range_projection_batch_dev: invalid argument (n_clouds=%d H=%d W=%d n_beams=%d)", n_clouds, H, W, n_beams);
    return LT_ERR_INVALID_ARG;
  }
  size_t n_max = 0;
  for (int k = 0; k < n_clouds; ++k) {
    if (clouds[k].n < 0 || (clouds[k].n > 0 && !clouds[k].points)) {
      lt_set_error("lt_range_projection_batch_dev: cloud %d: n=%d points=%p", k, clouds[k].n, clouds[k].points);
      return LT_ERR_INVALID_ARG;
    }
    n_max = std::max(n_max, (size_t)clouds[k].n);
  }
  std::lock_guard<std::mutex> lock(p->mu);
  LT_HIP(hipSetDevice(p->device));
  hipStream_t st = (hipStream_t)stream;
  const bool old_f64 = is_f64 && !(flags & LT_PROJ_NEW);
  LT_CHECK(pj_reserve(p, n_max, (size_t)H * W, old_f64, st));
  {  // the per-workgroup partial bounds of the clouds that want get_bnds()
    bool want = false;
    for (int k = 0; k < n_clouds; ++k) want = want || out[k].bnds;
    const size_t nb = n_max / 256 + 2;
    if (want && nb > p->bacc_blocks) {
      if (p->bacc) { LT_HIP(hipStreamSynchronize(st)); (void)hipFree(p->bacc); p->bacc = nullptr; p->bacc_blocks = 0; }
      const size_t cap = nb + nb / 4 + 64;
      LT_HIP(hipMalloc((void**)&p->bacc, LT_PB_MAX * cap * 6 * sizeof(unsigned long long)));
      p->bacc_blocks = cap;
    }
  }
  p->armed = false;  // (until the resolve pass of this call has been queued)
  if (old_f64) p->dmin_armed = false;
  if (n_beams > 0 && (n_beams != p->n_beams_cached || memcmp(p->beams_host, beam_angles, n_beams * sizeof(double)) != 0)) {
    // the table is read by kernels of EARLIER calls on this stream: the copy is stream-ordered behind them; a pageable
    // source is staged by the runtime before the call returns, so beams_host may be overwritten by the next call
    memcpy(p->beams_host, beam_angles, n_beams * sizeof(double));
    LT_HIP(hipMemcpyAsync(p->beams, p->beams_host, n_beams * sizeof(double), hipMemcpyHostToDevice, st));
    p->n_beams_cached = n_beams;
  }
  for (int g0 = 0; g0 < n_clouds; g0 += LT_PB_MAX) {
    pb_args A;
    A.n_clouds = std::min(LT_PB_MAX, n_clouds - g0);
    int blocks = 0;
    bool need_prefix = false;
    for (int k = 0; k < A.n_clouds; ++k) {
      const lt_cloud& ci = clouds[g0 + k];
      const lt_proj_images& o = out[g0 + k];
      pb_cloud& c = A.c[k];
      c.pts = ci.points; c.rem = ci.rem; c.label = ci.label; c.n = ci.n; c.block0 = blocks;
      blocks += (ci.n + 255) / 256;
      c.key = p->key[k]; c.dmin = p->dmin[k]; c.keep = p->keep[k]; c.wprefix = p->wprefix[k]; c.meta = p->meta + 2 * k;
      c.idx_img = o.idx; c.range_img = o.range; c.xyz_img = o.xyz; c.rem_img = o.rem; c.label_img = o.label;
      c.color_img = o.color; c.mask_img = o.mask; c.fold_img = o.label_folded; c.px_img = o.proj_x; c.py_img = o.proj_y;
      c.xf_img = o.proj_xf; c.yf_img = o.proj_yf; c.n_kept = o.n_kept;
      c.bacc = o.bnds ? p->bacc + 6 * p->bacc_blocks * k : nullptr; c.bnds_out = o.bnds;
      need_prefix = need_prefix || o.idx || o.mask || o.proj_x || o.proj_y || o.proj_xf || o.proj_yf || o.n_kept;
    }
    for (int k = A.n_clouds; k < LT_PB_MAX; ++k) { A.c[k] = A.c[0]; A.c[k].n = 0; A.c[k].block0 = 0x7fffffff; }
    bool wants_bnds = false;
    for (int k = 0; k < A.n_clouds; ++k) wants_bnds = wants_bnds || A.c[k].bacc;
    const int rc = is_f64 ? pj_run<double>(p, A, blocks, old_f64, need_prefix, wants_bnds, fov_up, fov_down, H, W, n_beams, flags,
                                           color_lut, lut_len, range_init, rem_init, xyz_init, st)
                          : pj_run<float>(p, A, blocks, false, need_prefix, wants_bnds, fov_up, fov_down, H, W, n_beams, flags,
                                          color_lut, lut_len, range_init, rem_init, xyz_init, st);
    if (rc != LT_OK) return rc;
  }
  p->armed = true;  // k_pb_resolve re-armed every cell it looked at
  if (old_f64) p->dmin_armed = true;
  return LT_OK;
}

// One output scan of the `mesh` adaption FROM POINT CLOUDS in one call (see include/lidarhip.h): projection of the source
// scans into images the projector owns, then lt_fusion_scan_dev on them.  One native call per output scan keeps a host
// thread's share at a few microseconds -- from Python the interpreter lock is released for all of it.
extern "C" int lt_deform_scan_dev(lt_projector* p, lt_tsdf* vol, lt_mesh* mesh, lt_scene* scene, lt_rayset* rayset,
                                  int n_clouds, const lt_cloud* clouds, int is_f64, double fov_up, double fov_down, int H,
                                  int W, const double* beam_angles, int n_beams, float obs_weight, unsigned tsdf_flags,
                                  const float* origin, float* endpoints, int* endcolors, float* range, float* endrem,
                                  int* tri, unsigned trace_flags, void* stream, int sync) {
  if (!p || n_clouds < 0 || n_clouds > 64 || H <= 0 || W <= 0) {
    lt_set_error("lt_deform_scan_dev: invalid argument (n_clouds=%d H=%d W=%d)", n_clouds, H, W);
    return LT_ERR_INVALID_ARG;
  }
  const size_t cells = (size_t)H * W, need = (size_t)(n_clouds > 0 ? n_clouds : 1) * 3 * cells;
  {
    std::lock_guard<std::mutex> lock(p->mu);
    LT_HIP(hipSetDevice(p->device));
    if (need > p->img_cap) {
      if (p->img) { LT_HIP(hipStreamSynchronize((hipStream_t)stream)); (void)hipFree(p->img); p->img = nullptr; p->img_cap = 0; }
      LT_HIP(hipMalloc((void**)&p->img, need * sizeof(float)));
      p->img_cap = need;
    }
  }
  lt_proj_images out[64];
  const float* color_ims[64];
  const float* depth_ims[64];
  const float* rem_ims[64];
  memset(out, 0, sizeof(out));
  for (int k = 0; k < n_clouds; ++k) {
    float* base = p->img + (size_t)k * 3 * cells;
    out[k].range = base; out[k].rem = base + cells; out[k].label_folded = base + 2 * cells;
    depth_ims[k] = base; rem_ims[k] = base + cells; color_ims[k] = base + 2 * cells;
  }
  // do_range_projection_new(fov, remove=True) + do_label_projection_new per source scan (laserscan.py:874-881)
  LT_CHECK(lt_range_projection_batch_dev(p, n_clouds, clouds, is_f64, fov_up, fov_down, H, W, beam_angles, n_beams,
                                         LT_PROJ_NEW | LT_PROJ_REMOVE, nullptr, 0, out, 0.0f, -1.0f, 0.0f, stream));
  return lt_fusion_scan_dev(vol, mesh, scene, rayset, n_clouds, color_ims, depth_ims, rem_ims, H, W, obs_weight, tsdf_flags,
                            origin, endpoints, endcolors, range, endrem, tri, trace_flags, stream, sync);
}

// deform('mergemesh') of one output scan in one call (see include/lidarhip.h)
extern "C" int lt_mergemesh_scan_dev(lt_projector* p, lt_mm_state* mm, int seq, lt_tsdf* vol, lt_mesh* mesh, lt_scene* scene,
                                     lt_rayset* rayset, const lt_cloud* cloud, int is_f64, double fov_up, double fov_down,
                                     int H, int W, const double* beam_angles, int n_beams, float obs_weight,
                                     unsigned tsdf_flags, const float* origin, float* endpoints, int* endcolors, float* range,
                                     float* endrem, int* tri, unsigned trace_flags, void* stream, lt_mm_geometry* geo,
                                     int* done) {
  int ticket = -1;
  auto fail = [&](int rc) {  // the scans behind this one must not wait for a turn that will not come
    if (mm && seq >= 0 && ticket < 0) (void)lt_mm_geometry_dev(mm, nullptr, seq, &ticket, stream);
    return rc;
  };
  if (!p || !mm || !mesh || !scene || !rayset || !cloud || !geo || !done || H <= 0 || W <= 0) {
    lt_set_error("lt_mergemesh_scan_dev: invalid argument");
    return fail(LT_ERR_INVALID_ARG);
  }
  *done = 0;
  const size_t cells = (size_t)H * W;
  {
    std::lock_guard<std::mutex> lock(p->mu);
    if (hipSetDevice(p->device) != hipSuccess) return fail(LT_ERR_HIP);
    if (3 * cells > p->img_cap) {
      if (p->img) { (void)hipStreamSynchronize((hipStream_t)stream); (void)hipFree(p->img); p->img = nullptr; p->img_cap = 0; }
      if (hipMalloc((void**)&p->img, 3 * cells * sizeof(float)) != hipSuccess) {
        lt_set_error("lt_mergemesh_scan_dev: hipMalloc of the source images failed");
        return fail(LT_ERR_NO_MEMORY);
      }
      p->img_cap = 3 * cells;
    }
    if (!p->mm_bnds && hipMalloc((void**)&p->mm_bnds, 6 * sizeof(double)) != hipSuccess) {
      lt_set_error("lt_mergemesh_scan_dev: hipMalloc failed");
      return fail(LT_ERR_NO_MEMORY);
    }
  }
  lt_proj_images out;
  memset(&out, 0, sizeof(out));
  out.range = p->img; out.rem = p->img + cells; out.label_folded = p->img + 2 * cells; out.bnds = p->mm_bnds;
  int rc = lt_range_projection_batch_dev(p, 1, cloud, is_f64, fov_up, fov_down, H, W, beam_angles, n_beams,
                                         LT_PROJ_NEW | LT_PROJ_REMOVE, nullptr, 0, &out, 0.0f, -1.0f, 0.0f, stream);
  if (rc != LT_OK) return fail(rc);
  rc = lt_mm_geometry_dev(mm, p->mm_bnds, seq, &ticket, stream);
  if (rc != LT_OK) return rc;
  if (vol) {  // the chain on the expected geometry; it waits for the stream once (the mesh sizes): the record is in by then
    LT_CHECK(lt_mergemesh_rerun_dev(p, vol, mesh, scene, rayset, H, W, obs_weight, tsdf_flags, origin, endpoints, endcolors, range,
                                    endrem, tri, trace_flags, stream));
  }
  LT_CHECK(lt_mm_geometry_get(mm, ticket, geo));
  *done = (vol && geo->status == 0 && memcmp(vol->bnds_given, geo->bnds_given, sizeof(geo->bnds_given)) == 0) ? 1 : 0;
  return LT_OK;
}

extern "C" int lt_mergemesh_rerun_dev(lt_projector* p, lt_tsdf* vol, lt_mesh* mesh, lt_scene* scene, lt_rayset* rayset, int H, int W,
                                      float obs_weight, unsigned tsdf_flags, const float* origin, float* endpoints, int* endcolors,
                                      float* range, float* endrem, int* tri, unsigned trace_flags, void* stream) {
  if (!p || !vol || H <= 0 || W <= 0 || !p->img || p->img_cap < 3 * (size_t)H * W) {
    lt_set_error("lt_mergemesh_rerun_dev: invalid argument (no projected scan of this shape)");
    return LT_ERR_INVALID_ARG;
  }
  const size_t cells = (size_t)H * W;
  const float* depth_im = p->img;
  const float* rem_im = p->img + cells;
  const float* color_im = p->img + 2 * cells;
  return lt_fusion_scan_dev(vol, mesh, scene, rayset, 1, &color_im, &depth_im, &rem_im, H, W, obs_weight, tsdf_flags, origin,
                            endpoints, endcolors, range, endrem, tri, trace_flags, stream, 0);
}

// Host-pointer convenience: stages everything through device buffers, same semantics.
extern "C" int lt_range_projection(const void* points, int is_f64, const float* rem, const unsigned* label, int n,
                                   double fov_up, double fov_down, int H, int W, const double* beam_angles,
                                   int n_beams, unsigned flags, const float* color_lut, int lut_len,
                                   void* points_kept, float* rem_kept, unsigned* label_kept, void* depth_kept,
                                   int* proj_x_kept, int* proj_y_kept, void* proj_xf_kept, void* proj_yf_kept,
                                   int* idx_img, float* range_img, float* xyz_img, float* rem_img, int* label_img,
                                   float* color_img, float* mask_img, float range_init, float rem_init,
                                   float xyz_init, int* n_kept) {
  if (n < 0 || H <= 0 || W <= 0 || (n > 0 && !points)) {
    lt_set_error("lt_range_projection: invalid argument (n=%d H=%d W=%d)", n, H, W);
    return LT_ERR_INVALID_ARG;
  }
  const size_t es = is_f64 ? 8 : 4, cells = (size_t)H * W, N = (size_t)n;
  // one staging allocation: inputs | per-point outputs | images
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t sz[] = {al(N * 3 * es), al(N * 4), al(N * 4), al((size_t)(lut_len > 0 ? lut_len : 0) * 12),
                       al(N * 3 * es), al(N * 4), al(N * 4), al(N * es), al(N * 4), al(N * 4), al(N * es), al(N * es),
                       al(cells * 4), al(cells * 4), al(cells * 12), al(cells * 4), al(cells * 4), al(cells * 12),
                       al(cells * 4)};
  size_t total = 256;
  for (size_t s : sz) total += s;
  char* base = nullptr;
  LT_HIP(hipMalloc((void**)&base, total));
  char* p[19];
  {
    size_t off = 0;
    for (int k = 0; k < 19; ++k) { p[k] = base + off; off += sz[k]; }
  }
  int rc = LT_OK;
  auto H2D = [&](void* d, const void* h, size_t b) {
    if (rc == LT_OK && h && b && hipMemcpy(d, h, b, hipMemcpyHostToDevice) != hipSuccess) {
      lt_set_error("lt_range_projection: H2D copy failed");
      rc = LT_ERR_HIP;
    }
  };
  H2D(p[0], points, N * 3 * es);
  H2D(p[1], rem, N * 4);
  H2D(p[2], label, N * 4);
  H2D(p[3], color_lut, (size_t)(lut_len > 0 ? lut_len : 0) * 12);
  int kept = 0;
  if (rc == LT_OK)
    rc = lt_range_projection_dev(p[0], is_f64, rem ? (float*)p[1] : nullptr, label ? (unsigned*)p[2] : nullptr, n,
                                 fov_up, fov_down, H, W, beam_angles, n_beams, flags,
                                 (color_lut && lut_len > 0) ? (float*)p[3] : nullptr, lut_len,
                                 points_kept ? p[4] : nullptr, rem_kept ? (float*)p[5] : nullptr,
                                 label_kept ? (unsigned*)p[6] : nullptr, depth_kept ? p[7] : nullptr,
                                 proj_x_kept ? (int*)p[8] : nullptr, proj_y_kept ? (int*)p[9] : nullptr,
                                 proj_xf_kept ? p[10] : nullptr, proj_yf_kept ? p[11] : nullptr,
                                 idx_img ? (int*)p[12] : nullptr, range_img ? (float*)p[13] : nullptr,
                                 xyz_img ? (float*)p[14] : nullptr, rem_img ? (float*)p[15] : nullptr,
                                 label_img ? (int*)p[16] : nullptr, color_img ? (float*)p[17] : nullptr,
                                 mask_img ? (float*)p[18] : nullptr, range_init, rem_init, xyz_init, &kept, nullptr);
  auto D2H = [&](void* h, const void* d, size_t b) {
    if (rc == LT_OK && h && b && hipMemcpy(h, d, b, hipMemcpyDeviceToHost) != hipSuccess) {
      lt_set_error("lt_range_projection: D2H copy failed");
      rc = LT_ERR_HIP;
    }
  };
  const size_t K = (size_t)kept;
  D2H(points_kept, p[4], K * 3 * es);
  D2H(rem_kept, p[5], K * 4);
  D2H(label_kept, p[6], K * 4);
  D2H(depth_kept, p[7], K * es);
  D2H(proj_x_kept, p[8], K * 4);
  D2H(proj_y_kept, p[9], K * 4);
  D2H(proj_xf_kept, p[10], K * es);
  D2H(proj_yf_kept, p[11], K * es);
  D2H(idx_img, p[12], cells * 4);
  D2H(range_img, p[13], cells * 4);
  D2H(xyz_img, p[14], cells * 12);
  D2H(rem_img, p[15], cells * 4);
  D2H(label_img, p[16], cells * 4);
  D2H(color_img, p[17], cells * 12);
  D2H(mask_img, p[18], cells * 4);
  (void)hipFree(base);
  if (n_kept) *n_kept = kept;
  return rc;
}
