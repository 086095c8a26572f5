// lt_host.hip -- the HOST-buffer side of the boundary: the reference's one-call `ctrace` and its pipelined form.
//
//   lt_ctrace / lt_ctrace_ex   the 14-parameter drop-in (RayTracer.cpp:116-124): upload, render, download, one scan per
//                              call.  All state is per calling THREAD (scene, ray set, staging, HIP stream): no
//                              process-global mutable state, threads do not serialise each other.
//   lt_hostpipe_*              the same work for a SEQUENCE of scans kept in flight: an uploader thread of the pipe
//                              moves scan i+1 to the device while scan i renders and scan i-1 downloads (three HIP
//                              streams, the link is full duplex); a launcher thread issues render + download of
//                              each uploaded scan so that the uploader is never outside a transfer.  On this platform
//                              a pageable hipMemcpyAsync runs at the wire rate (56 GB/s, tools/pcie_probe.hip) but
//                              blocks its caller; the two threads are what turns that into an asynchronous submit.  Payload per scan: verts + faces + rem as the
//                              reference holds them, colours as the uint8 [V,3] `get_mesh` returns (fusion_lidar.py:423)
//                              instead of their int32 copy, no rays (the sensor model's ray set is built once), no
//                              pre-zeroed images in (the pipe writes every cell: misses are 0 / tri -1, exactly what
//                              the reference's pre-zeroed arrays hold afterwards).
#include "lt_internal.h"
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>
#include <stdlib.h>
#include <string.h>

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// ---- small device helpers -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_words_differ(const unsigned* __restrict__ a, const unsigned* __restrict__ b,
                                                      size_t n, unsigned* __restrict__ flag) {
  bool diff = false;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) diff |= a[i] != b[i];
  if (__ballot(diff) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

// colours as get_mesh returns them (uint8 [V,3], fusion_lidar.py:423) -> the int32 [V,3] ctrace reads (:435)
__global__ __launch_bounds__(256) void k_widen_u8(const unsigned char* __restrict__ in, int* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = (int)in[i];
}

// ---- lt_ctrace: per-thread context -----------------------------------------------------------------------------------
namespace {
struct ctrace_ctx {
  int device = -1;
  hipStream_t stream = nullptr;
  lt_scene* scene = nullptr;
  lt_rayset* rs = nullptr;
  void* io = nullptr;      // rays (new) | endpoints | endcolors | range | endrem | tri
  size_t io_bytes = 0;
  float* rays_cached = nullptr;  // device copy of the rays the ray set was built from
  size_t rays_cached_n = 0;      // floats
  int rs_height = 0;
  unsigned rs_norm = 0;
  unsigned* flag_dev = nullptr;
  unsigned* flag_host = nullptr;  // pinned

  void release() {
    if (device < 0) return;
    if (hipSetDevice(device) != hipSuccess) return;  // runtime already gone (process exit): nothing to free
    (void)hipDeviceSynchronize();
    if (rs) (void)lt_rayset_destroy(rs);
    if (scene) (void)lt_scene_destroy(scene);
    if (io) (void)hipFree(io);
    if (rays_cached) (void)hipFree(rays_cached);
    if (flag_dev) (void)hipFree(flag_dev);
    if (flag_host) (void)hipHostFree(flag_host);
    if (stream) (void)hipStreamDestroy(stream);
    rs = nullptr; scene = nullptr; io = nullptr; rays_cached = nullptr; flag_dev = nullptr; flag_host = nullptr;
    stream = nullptr; io_bytes = 0; rays_cached_n = 0; device = -1;
  }
  ~ctrace_ctx() { release(); }
};
thread_local ctrace_ctx t_ctx;
}  // namespace

static int ctrace_thread(const float* rays, const float* origin, const float* verts, const int* faces,
                         const int* colors, const float* rem, int n_rays, int n_verts, int n_faces, int height,
                         float* endpoints, int* endcolors, float* range, float* endrem, int* tri, lt_stats* stats) {
  ctrace_ctx& c = t_ctx;
  int dev = 0;
  LT_HIP(hipGetDevice(&dev));
  if (c.device != dev) {
    c.release();
    LT_HIP(hipSetDevice(dev));
    // the context counts as initialised (c.device == dev) only once ALL of its resources exist: a failure half way
    // releases what was created, and the next call on this thread starts over instead of finding a NULL scene
    int rc = LT_OK;
    if (hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking) != hipSuccess) rc = LT_ERR_HIP;
    if (rc == LT_OK) rc = lt_scene_create(&c.scene, dev);
    if (rc == LT_OK && hipMalloc((void**)&c.flag_dev, sizeof(unsigned)) != hipSuccess) rc = LT_ERR_NO_MEMORY;
    if (rc == LT_OK && hipHostMalloc((void**)&c.flag_host, sizeof(unsigned), hipHostMallocDefault) != hipSuccess) rc = LT_ERR_NO_MEMORY;
    c.device = dev;  // (release() frees on this device)
    if (rc != LT_OK) {
      lt_set_error("lt_ctrace: creating the per-thread context on device %d failed: %s", dev,
                   hipGetErrorString(hipGetLastError()));
      c.release();  // resets c.device to -1: the next call starts over
      return rc;
    }
  }
  lt_scene* s = c.scene;
  hipStream_t stream = c.stream;
  const int W = n_rays / height;
  const size_t R = (size_t)W * height;
  const size_t b3 = align256(R * 12), b1 = align256(R * 4);
  const size_t total = 3 * b3 + 3 * b1 + 256;
  if (total > c.io_bytes) {
    if (c.io) {
      LT_HIP(hipStreamSynchronize(stream));
      (void)hipFree(c.io);
      c.io = nullptr;
      c.io_bytes = 0;
    }
    LT_HIP(hipMalloc(&c.io, total));
    c.io_bytes = total;
  }
  char* io = (char*)c.io;
  float* d_rays = (float*)io;
  float* d_end = (float*)(io + b3);
  int* d_col = (int*)(io + 2 * b3);
  float* d_range = (float*)(io + 3 * b3);
  float* d_rem = (float*)(io + 3 * b3 + b1);
  int* d_tri = (int*)(io + 3 * b3 + 2 * b1);
  const unsigned norm_flag = lt_env_norm_flag();
  const char* sg = getenv("LIDARHIP_STRATEGY");
  const bool lbvh = sg && strcmp(sg, "lbvh") == 0;
  // The rays first: a sensor model's rays are the same for every scan of a sequence, so the binned ray set is rebuilt
  // only when they change -- decided by comparing the upload with the device copy the ray set was built from ON THE
  // DEVICE, under the mesh upload (no host pass over the rays)
  bool maybe_same = !lbvh && c.rs && c.rs_height == height && c.rs_norm == norm_flag && c.rays_cached_n == R * 3;
  if (R > 0) {
    LT_HIP(hipMemcpyAsync(d_rays, rays, R * 12, hipMemcpyHostToDevice, stream));
    if (maybe_same) {
      LT_HIP(hipMemsetAsync(c.flag_dev, 0, sizeof(unsigned), stream));
      hipLaunchKernelGGL(k_words_differ, dim3(256), dim3(256), 0, stream, (const unsigned*)d_rays,
                         (const unsigned*)c.rays_cached, R * 3, c.flag_dev);
      LT_HIP(hipMemcpyAsync(c.flag_host, c.flag_dev, sizeof(unsigned), hipMemcpyDeviceToHost, stream));
    }
  }
  LT_CHECK(lt_scene_set_mesh_host(s, verts, faces, colors, rem, n_verts, n_faces, stream));
  if (R > 0) {
    // outputs are written only for hits (RayTracer.cpp:73): start from the caller's contents
    if (endpoints) LT_HIP(hipMemcpyAsync(d_end, endpoints, R * 12, hipMemcpyHostToDevice, stream));
    if (endcolors) LT_HIP(hipMemcpyAsync(d_col, endcolors, R * 12, hipMemcpyHostToDevice, stream));
    if (range) LT_HIP(hipMemcpyAsync(d_range, range, R * 4, hipMemcpyHostToDevice, stream));
    if (endrem) LT_HIP(hipMemcpyAsync(d_rem, endrem, R * 4, hipMemcpyHostToDevice, stream));
    if (tri) LT_HIP(hipMemcpyAsync(d_tri, tri, R * 4, hipMemcpyHostToDevice, stream));
  }
  lt_stats st;
  memset(&st, 0, sizeof(st));
  if (lbvh) {
    // LIDARHIP_STRATEGY=lbvh: build the linear BVH and traverse it; default: single-origin triangle scatter
    // (lt_scatter.hip) -- both produce identical images
    LT_CHECK(lt_build_launch(s, stream, stats ? &st : nullptr));
    LT_CHECK(lt_trace_launch(s, d_rays, origin, (int)R, height, endpoints ? d_end : nullptr,
                             endcolors ? d_col : nullptr, range ? d_range : nullptr, endrem ? d_rem : nullptr,
                             tri ? d_tri : nullptr, (stats ? LT_TRACE_COUNT : 0u) | norm_flag, stream,
                             stats ? &st : nullptr));
  } else {
    if (maybe_same && R > 0) {
      LT_HIP(hipStreamSynchronize(stream));  // (the uploads above are done by now anyway when the memory is pageable)
      maybe_same = *c.flag_host == 0u;
    }
    if (!maybe_same) {
      if (c.rs) (void)lt_rayset_destroy(c.rs);
      c.rs = nullptr;
      if (c.rays_cached_n < R * 3 || !c.rays_cached) {
        if (c.rays_cached) {
          LT_HIP(hipStreamSynchronize(stream));
          (void)hipFree(c.rays_cached);
          c.rays_cached = nullptr;
        }
        LT_HIP(hipMalloc((void**)&c.rays_cached, (R * 3 + 1) * sizeof(float)));
      }
      c.rays_cached_n = R * 3;
      if (R > 0) LT_HIP(hipMemcpyAsync(c.rays_cached, d_rays, R * 12, hipMemcpyDeviceToDevice, stream));
      LT_CHECK(lt_rayset_create_dev(&c.rs, d_rays, (int)R, height, norm_flag, stream));
      c.rs_height = height;
      c.rs_norm = norm_flag;
    }
    LT_CHECK(lt_scene_render_dev(s, c.rs, origin, endpoints ? d_end : nullptr, endcolors ? d_col : nullptr,
                                 range ? d_range : nullptr, endrem ? d_rem : nullptr, tri ? d_tri : nullptr,
                                 (stats ? LT_TRACE_COUNT : 0u), stream, stats ? &st : nullptr));
  }
  if (R > 0) {
    if (endpoints) LT_HIP(hipMemcpyAsync(endpoints, d_end, R * 12, hipMemcpyDeviceToHost, stream));
    if (endcolors) LT_HIP(hipMemcpyAsync(endcolors, d_col, R * 12, hipMemcpyDeviceToHost, stream));
    if (range) LT_HIP(hipMemcpyAsync(range, d_range, R * 4, hipMemcpyDeviceToHost, stream));
    if (endrem) LT_HIP(hipMemcpyAsync(endrem, d_rem, R * 4, hipMemcpyDeviceToHost, stream));
    if (tri) LT_HIP(hipMemcpyAsync(tri, d_tri, R * 4, hipMemcpyDeviceToHost, stream));
  }
  LT_HIP(hipStreamSynchronize(stream));
  if (stats) *stats = st;
  return lt_scene_status(s);
}

extern "C" int lt_ctrace_ex(const float* rays, const float* origin, const float* verts, const int* faces,
                            const int* colors, const float* rem, int n_rays, int n_verts, int n_faces,
                            int height, float* endpoints, int* endcolors, float* range, float* endrem, int* tri,
                            lt_stats* stats) {
  if (height <= 0 || n_rays < 0 || !origin || (n_rays > 0 && !rays)) {
    lt_set_error("lt_ctrace: invalid argument (n_rays=%d height=%d)", n_rays, height);
    return LT_ERR_INVALID_ARG;
  }
  LT_CHECK(lt_check_mesh_args("lt_ctrace", verts, faces, colors, rem, n_verts, n_faces));
  const int rc = ctrace_thread(rays, origin, verts, faces, colors, rem, n_rays, n_verts, n_faces, height, endpoints,
                               endcolors, range, endrem, tri, stats);
  // a failure half way may leave copies from / to the caller's arrays queued: they must not outlive the call
  if (rc != LT_OK && t_ctx.stream) (void)hipStreamSynchronize(t_ctx.stream);
  return rc;
}

extern "C" int lt_ctrace(const float* rays, const float* origin, const float* verts, const int* faces,
                         const int* colors, const float* rem, int n_rays, int n_verts, int n_faces, int height,
                         float* endpoints, int* endcolors, float* range, float* endrem) {
  return lt_ctrace_ex(rays, origin, verts, faces, colors, rem, n_rays, n_verts, n_faces, height, endpoints,
                      endcolors, range, endrem, nullptr, nullptr);
}

// ---- pinned host memory for callers ------------------------------------------------------------------------------------
extern "C" int lt_host_alloc(void** p, size_t bytes) {
  if (!p) {
    lt_set_error("lt_host_alloc: NULL out pointer");
    return LT_ERR_INVALID_ARG;
  }
  *p = nullptr;
  LT_HIP(hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocDefault));
  return LT_OK;
}

extern "C" int lt_host_free(void* p) {
  if (p) LT_HIP(hipHostFree(p));
  return LT_OK;
}

// ---- lt_hostpipe --------------------------------------------------------------------------------------------------------
struct hp_reg { const void* ptr; size_t bytes; int count; };

struct hp_job {
  int ticket;
  float origin[3];
  const float* verts; const int* faces; const void* colors; const float* rem;
  int colors_u8, n_verts, n_faces;
  float* endpoints; int* endcolors; float* range; float* endrem; int* tri;
  const void* pinned[9];  // ranges registered for this job (unregistered when it is collected)
  int n_pinned;
  int direct_out;         // the downloads went straight into the caller's arrays
};

struct hp_slot {
  lt_scene* scene = nullptr;
  char* mesh = nullptr; size_t mesh_bytes = 0;   // device: verts | faces | colors i32 | rem | colors u8
  char* out_dev = nullptr;                       // device: endpoints | endcolors | range | endrem | tri
  char* out_pin = nullptr;                       // pinned staging of the same layout
  hipEvent_t ev_up = nullptr, ev_run = nullptr, ev_done = nullptr;
  hp_job job;
  int state = 0;  // 0 free, 1 queued, 2 issued (ev_done recorded), 3 failed
  int rc = LT_OK;
  char err[256];
};

struct lt_hostpipe {
  int device, depth, n_rays, height;
  unsigned flags;
  // LIDARHIP_HOSTPIPE_REGISTER=1: hipHostRegister the caller's arrays and copy asynchronously from / into them instead
  // of blocking pageable uploads + a staged download.  Off by default: measured slower on this platform (0.53 vs
  // 0.47 ms per C2 scan, tools/hostpipe_probe.cpp), where a pageable copy already runs at 88 % of the wire rate.
  bool use_register = false;
  size_t b3, b1, bc;  // bytes of a [R,3] f32 image, a [R] image, the colour image
  lt_rayset* rs = nullptr;
  // LIDARHIP_HOSTPIPE_UPLOADERS (default: 2 when depth >= 4, else 1): uploader threads, each with its own stream, taking
  // scans alternately -- a pageable copy occupies its caller, and the fixed part of one thread's copy (staging ramp-up
  // and drain, ~17 us per array) runs under the other thread's transfer: C2 scans 0.452 -> 0.422 ms (52 GB/s of uploads,
  // 0.92 of the measured wire rate); a third uploader adds nothing
  int n_up = 1;
  hipStream_t s_up[4] = {nullptr, nullptr, nullptr, nullptr}, s_run = nullptr, s_down = nullptr;
  std::vector<hp_slot> slots;
  std::mutex mu;
  std::condition_variable cv_work, cv_launch, cv_done;
  std::deque<int> queue;         // slots to upload, in ticket order
  std::deque<int> launch_queue;  // uploaded slots whose render + download are to be issued
  int uploaders_done = 0;
  int next_ticket = 0;
  double t_issue = 0, t_upload = 0, t_collect = 0;  // seconds spent by the worker issuing / uploading, by callers collecting
  long n_issued = 0;
  // Host ranges this pipe has registered with the HIP runtime (hipHostRegister: ~1 us per array on this platform,
  // tools/pcie_probe.hip) so that copies from / to the CALLER'S arrays are asynchronous DMA instead of blocking staged
  // copies; reference counted because the scans in flight may share arrays.
  std::mutex reg_mu;
  std::vector<hp_reg> regs;
  double trace[256][6];  // debug: per ticket % 256: submit, issue start, upload end, issue end, collect start, collect end
  bool stop = false;
  std::thread worker[4], launcher;
};

static double hp_now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Register [ptr, ptr + bytes) for asynchronous DMA; false = leave it to the blocking pageable path.
static bool hp_pin(lt_hostpipe* p, hp_job& j, const void* ptr, size_t bytes) {
  if (!ptr || bytes == 0) return false;
  std::lock_guard<std::mutex> lk(p->reg_mu);
  for (hp_reg& r : p->regs)
    if (r.ptr == ptr && r.bytes >= bytes) {
      ++r.count;
      j.pinned[j.n_pinned++] = ptr;
      return true;
    }
  if (hipHostRegister(const_cast<void*>(ptr), bytes, hipHostRegisterDefault) != hipSuccess) {
    (void)hipGetLastError();  // e.g. already registered by the caller (lt_host_alloc memory is DMA-able anyway)
    hipPointerAttribute_t at;
    const bool pinned = hipPointerGetAttributes(&at, ptr) == hipSuccess && at.type == hipMemoryTypeHost;
    (void)hipGetLastError();
    return pinned;
  }
  p->regs.push_back({ptr, bytes, 1});
  j.pinned[j.n_pinned++] = ptr;
  return true;
}

static void hp_unpin_all(lt_hostpipe* p, hp_job& j) {
  std::lock_guard<std::mutex> lk(p->reg_mu);
  for (int k = 0; k < j.n_pinned; ++k)
    for (size_t i = 0; i < p->regs.size(); ++i)
      if (p->regs[i].ptr == j.pinned[k]) {
        if (--p->regs[i].count == 0) {
          (void)hipHostUnregister(const_cast<void*>(j.pinned[k]));
          p->regs[i] = p->regs.back();
          p->regs.pop_back();
        }
        break;
      }
  j.n_pinned = 0;
}

// first half of a scan, on the UPLOADER thread: the mesh from the caller's arrays into the slot's device buffer
static int hp_upload(lt_hostpipe* p, hp_slot& sl, hipStream_t s_up) {
  hp_job& j = sl.job;
  const double t0 = hp_now();
  p->trace[j.ticket & 255][1] = t0;
  LT_HIP(hipSetDevice(p->device));
  const size_t bv = align256((size_t)j.n_verts * 12), bf = align256((size_t)j.n_faces * 12),
               br = align256((size_t)j.n_verts * 4), b8 = align256((size_t)j.n_verts * 3);
  const size_t total = 2 * bv + bf + br + b8 + 256;
  if (total > sl.mesh_bytes) {
    if (sl.mesh) {
      LT_HIP(hipDeviceSynchronize());
      (void)hipFree(sl.mesh);
      sl.mesh = nullptr;
      sl.mesh_bytes = 0;
    }
    LT_HIP(hipMalloc((void**)&sl.mesh, total + total / 4));
    sl.mesh_bytes = total + total / 4;
  }
  float* dv = (float*)sl.mesh;
  int* df = (int*)(sl.mesh + bv);
  int* dc = (int*)(sl.mesh + bv + bf);
  float* dr = (float*)(sl.mesh + 2 * bv + bf);
  unsigned char* d8 = (unsigned char*)(sl.mesh + 2 * bv + bf + br);
  // upload: straight from the caller's arrays.  Registered ranges make the copies asynchronous DMA (they queue up
  // behind each other on the upload stream); anything that cannot be registered goes the pageable way, where each
  // call returns when its transfer is done -- which is this thread's job
  if (p->use_register) {
    hp_pin(p, j, j.verts, (size_t)j.n_verts * 12);
    hp_pin(p, j, j.colors, (size_t)j.n_verts * (j.colors_u8 ? 3 : 12));
    hp_pin(p, j, j.rem, (size_t)j.n_verts * 4);
    hp_pin(p, j, j.faces, (size_t)j.n_faces * 12);
  }
  if (j.n_verts > 0) {
    LT_HIP(hipMemcpyAsync(dv, j.verts, (size_t)j.n_verts * 12, hipMemcpyHostToDevice, s_up));
    if (j.colors_u8) LT_HIP(hipMemcpyAsync(d8, j.colors, (size_t)j.n_verts * 3, hipMemcpyHostToDevice, s_up));
    else LT_HIP(hipMemcpyAsync(dc, j.colors, (size_t)j.n_verts * 12, hipMemcpyHostToDevice, s_up));
    LT_HIP(hipMemcpyAsync(dr, j.rem, (size_t)j.n_verts * 4, hipMemcpyHostToDevice, s_up));
  }
  if (j.n_faces > 0) LT_HIP(hipMemcpyAsync(df, j.faces, (size_t)j.n_faces * 12, hipMemcpyHostToDevice, s_up));
  LT_HIP(hipEventRecord(sl.ev_up, s_up));
  {
    std::lock_guard<std::mutex> lk(p->mu);
    p->t_upload += hp_now() - t0;
  }
  p->trace[j.ticket & 255][2] = hp_now();
  return LT_OK;
}

// second half of a scan, issued by the LAUNCHER thread so that the uploader is already inside the next scan's
// transfer: render on s_run behind the upload event, download on s_down behind the render event
static int hp_launch(lt_hostpipe* p, hp_slot& sl) {
  hp_job& j = sl.job;
  const double t0 = hp_now();
  LT_HIP(hipSetDevice(p->device));
  const size_t bv = align256((size_t)j.n_verts * 12), bf = align256((size_t)j.n_faces * 12),
               br = align256((size_t)j.n_verts * 4);
  float* dv = (float*)sl.mesh;
  int* df = (int*)(sl.mesh + bv);
  int* dc = (int*)(sl.mesh + bv + bf);
  float* dr = (float*)(sl.mesh + 2 * bv + bf);
  unsigned char* d8 = (unsigned char*)(sl.mesh + 2 * bv + bf + br);
  LT_HIP(hipStreamWaitEvent(p->s_run, sl.ev_up, 0));
  if (j.colors_u8 && j.n_verts > 0)
    hipLaunchKernelGGL(k_widen_u8, dim3(512), dim3(256), 0, p->s_run, d8, dc, (size_t)j.n_verts * 3);
  LT_CHECK(lt_scene_set_mesh_dev(sl.scene, dv, df, dc, dr, j.n_verts, j.n_faces));
  char* o = sl.out_dev;
  float* d_end = (float*)o;
  int* d_col = (int*)(o + p->b3);
  float* d_range = (float*)(o + p->b3 + p->bc);
  float* d_rem = (float*)(o + p->b3 + p->bc + p->b1);
  int* d_tri = (int*)(o + p->b3 + p->bc + 2 * p->b1);
  LT_CHECK(lt_scene_render_dev(sl.scene, p->rs, j.origin, j.endpoints ? d_end : nullptr, j.endcolors ? d_col : nullptr,
                               j.range ? d_range : nullptr, j.endrem ? d_rem : nullptr, j.tri ? d_tri : nullptr,
                               p->flags | LT_TRACE_WRITE_MISSES, p->s_run, nullptr));
  LT_HIP(hipEventRecord(sl.ev_run, p->s_run));
  // download into pinned staging (asynchronous); lt_hostpipe_wait hands it to the caller's arrays
  LT_HIP(hipStreamWaitEvent(p->s_down, sl.ev_run, 0));
  const size_t R = (size_t)p->n_rays;
  const size_t ncol = (p->flags & LT_TRACE_LABEL_IMAGE) ? R * 4 : R * 12;
  char* h = sl.out_pin;
  // straight into the caller's arrays when all of them can be registered, else into pinned staging
  // (lt_hostpipe_wait copies from there)
  j.direct_out = 0;
  if (p->use_register && R > 0) {
    bool all = true;
    if (j.endpoints) all = all && hp_pin(p, j, j.endpoints, R * 12);
    if (j.endcolors) all = all && hp_pin(p, j, j.endcolors, ncol);
    if (j.range) all = all && hp_pin(p, j, j.range, R * 4);
    if (j.endrem) all = all && hp_pin(p, j, j.endrem, R * 4);
    if (j.tri) all = all && hp_pin(p, j, j.tri, R * 4);
    j.direct_out = all ? 1 : 0;
  }
  // staged: the five images are one block on the device and in the staging buffer -> ONE transfer
  if (R > 0 && !j.direct_out) {
    LT_HIP(hipMemcpyAsync(h, o, p->b3 + p->bc + 3 * p->b1, hipMemcpyDeviceToHost, p->s_down));
  } else if (R > 0) {
    const bool d = j.direct_out != 0;
    if (j.endpoints) LT_HIP(hipMemcpyAsync(d ? (void*)j.endpoints : (void*)h, d_end, R * 12, hipMemcpyDeviceToHost, p->s_down));
    if (j.endcolors) LT_HIP(hipMemcpyAsync(d ? (void*)j.endcolors : (void*)(h + p->b3), d_col, ncol, hipMemcpyDeviceToHost, p->s_down));
    if (j.range) LT_HIP(hipMemcpyAsync(d ? (void*)j.range : (void*)(h + p->b3 + p->bc), d_range, R * 4, hipMemcpyDeviceToHost, p->s_down));
    if (j.endrem) LT_HIP(hipMemcpyAsync(d ? (void*)j.endrem : (void*)(h + p->b3 + p->bc + p->b1), d_rem, R * 4, hipMemcpyDeviceToHost, p->s_down));
    if (j.tri) LT_HIP(hipMemcpyAsync(d ? (void*)j.tri : (void*)(h + p->b3 + p->bc + 2 * p->b1), d_tri, R * 4, hipMemcpyDeviceToHost, p->s_down));
  }
  LT_HIP(hipEventRecord(sl.ev_done, p->s_down));
  p->t_issue += hp_now() - t0;
  p->trace[j.ticket & 255][3] = hp_now();
  p->n_issued += 1;
  return LT_OK;
}

// debug helper (not part of the documented ABI): {scans issued, worker seconds issuing, of which uploading, caller
// seconds collecting}
extern "C" int lt_debug_hostpipe_times(lt_hostpipe* p, double* out) {
  if (!p || !out) return LT_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lk(p->mu);
  out[0] = (double)p->n_issued; out[1] = p->t_issue; out[2] = p->t_upload; out[3] = p->t_collect;
  return LT_OK;
}
extern "C" int lt_debug_hostpipe_trace(lt_hostpipe* p, double* out) {  // [256][6]
  if (!p || !out) return LT_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lk(p->mu);
  memcpy(out, p->trace, sizeof(p->trace));
  return LT_OK;
}

static void hp_fail(lt_hostpipe* p, hp_slot& sl, int rc) {
  {
    std::lock_guard<std::mutex> lk(p->mu);
    sl.rc = rc;
    strncpy(sl.err, lt_last_error(), sizeof(sl.err) - 1);
    sl.err[sizeof(sl.err) - 1] = 0;
    sl.state = 3;
  }
  p->cv_done.notify_all();
}

static void hp_uploader(lt_hostpipe* p, int which) {
  for (;;) {
    int k;
    {
      std::unique_lock<std::mutex> lk(p->mu);
      p->cv_work.wait(lk, [&] { return p->stop || !p->queue.empty(); });
      if (p->queue.empty()) break;  // stop requested and nothing left
      k = p->queue.front();
      p->queue.pop_front();
    }
    hp_slot& sl = p->slots[k];
    const int rc = hp_upload(p, sl, p->s_up[which]);
    if (rc != LT_OK) {
      hp_fail(p, sl, rc);
      continue;
    }
    {
      std::lock_guard<std::mutex> lk(p->mu);
      p->launch_queue.push_back(k);
    }
    p->cv_launch.notify_one();
  }
  {
    std::lock_guard<std::mutex> lk(p->mu);
    p->uploaders_done += 1;
  }
  p->cv_launch.notify_all();
}

static void hp_launcher(lt_hostpipe* p) {
  for (;;) {
    int k;
    {
      std::unique_lock<std::mutex> lk(p->mu);
      p->cv_launch.wait(lk, [&] { return p->uploaders_done == p->n_up || !p->launch_queue.empty(); });
      if (p->launch_queue.empty()) return;
      k = p->launch_queue.front();
      p->launch_queue.pop_front();
    }
    hp_slot& sl = p->slots[k];
    const int rc = hp_launch(p, sl);
    if (rc != LT_OK) {
      hp_fail(p, sl, rc);
      continue;
    }
    {
      std::lock_guard<std::mutex> lk(p->mu);
      sl.rc = LT_OK;
      sl.state = 2;
    }
    p->cv_done.notify_all();
  }
}

extern "C" int lt_hostpipe_destroy(lt_hostpipe* p) {
  if (!p) return LT_OK;
  {
    std::lock_guard<std::mutex> lk(p->mu);
    p->stop = true;
  }
  p->cv_work.notify_all();
  for (std::thread& w : p->worker)
    if (w.joinable()) w.join();
  if (p->launcher.joinable()) p->launcher.join();
  (void)hipSetDevice(p->device);
  (void)hipDeviceSynchronize();
  for (hp_slot& sl : p->slots) {
    if (sl.scene) (void)lt_scene_destroy(sl.scene);
    if (sl.mesh) (void)hipFree(sl.mesh);
    if (sl.out_dev) (void)hipFree(sl.out_dev);
    if (sl.out_pin) (void)hipHostFree(sl.out_pin);
    if (sl.ev_up) (void)hipEventDestroy(sl.ev_up);
    if (sl.ev_run) (void)hipEventDestroy(sl.ev_run);
    if (sl.ev_done) (void)hipEventDestroy(sl.ev_done);
  }
  if (p->rs) (void)lt_rayset_destroy(p->rs);
  for (hipStream_t su : p->s_up)
    if (su) (void)hipStreamDestroy(su);
  if (p->s_run) (void)hipStreamDestroy(p->s_run);
  if (p->s_down) (void)hipStreamDestroy(p->s_down);
  delete p;
  return LT_OK;
}

extern "C" int lt_hostpipe_create(lt_hostpipe** out, const float* rays, int n_rays, int height, int depth,
                                  unsigned flags, int device) {
  if (!out || n_rays < 0 || height <= 0 || depth < 1 || depth > 64 || (n_rays > 0 && !rays) ||
      (flags & ~(LT_TRACE_NORM_EXACT | LT_TRACE_NORM_AMD | LT_TRACE_LABEL_IMAGE | LT_TRACE_WRITE_MISSES))) {
    lt_set_error("lt_hostpipe_create: invalid argument (n_rays=%d height=%d depth=%d flags=%u)", n_rays, height, depth,
                 flags);
    return LT_ERR_INVALID_ARG;
  }
  *out = nullptr;
  if (device < 0) LT_HIP(hipGetDevice(&device));
  LT_HIP(hipSetDevice(device));
  lt_hostpipe* p = new lt_hostpipe();
  p->device = device;
  p->depth = depth;
  p->height = height;
  p->n_rays = (n_rays / height) * height;
  p->flags = flags & (LT_TRACE_LABEL_IMAGE | LT_TRACE_WRITE_MISSES);
  {
    const char* e = getenv("LIDARHIP_HOSTPIPE_REGISTER");
    p->use_register = e && strcmp(e, "1") == 0;
  }
  const size_t R = (size_t)p->n_rays;
  p->b3 = align256(R * 12);
  p->b1 = align256(R * 4);
  p->bc = (flags & LT_TRACE_LABEL_IMAGE) ? p->b1 : p->b3;
  int rc = LT_OK;
  auto ok = [&](hipError_t e, const char* what) {
    if (e != hipSuccess && rc == LT_OK) {
      lt_set_error("lt_hostpipe_create: %s failed: %s", what, hipGetErrorString(e));
      rc = e == hipErrorOutOfMemory ? LT_ERR_NO_MEMORY : LT_ERR_HIP;
    }
  };
  {
    const char* e = getenv("LIDARHIP_HOSTPIPE_UPLOADERS");
    p->n_up = e ? atoi(e) : (depth >= 4 ? 2 : 1);  // (two uploaders with only three slots starve the render: measured slower)
    p->n_up = p->n_up < 1 ? 1 : (p->n_up > 4 ? 4 : p->n_up);
    if (p->n_up > depth) p->n_up = depth;
  }
  for (int u = 0; u < p->n_up; ++u) ok(hipStreamCreateWithFlags(&p->s_up[u], hipStreamNonBlocking), "hipStreamCreate");
  ok(hipStreamCreateWithFlags(&p->s_run, hipStreamNonBlocking), "hipStreamCreate");
  ok(hipStreamCreateWithFlags(&p->s_down, hipStreamNonBlocking), "hipStreamCreate");
  // the ray set of the sensor model, once (create_rays depends only on the YAML, laserscan.py:1092-1119)
  float* d_rays = nullptr;
  if (rc == LT_OK) {
    ok(hipMalloc((void**)&d_rays, (R * 3 + 1) * sizeof(float)), "hipMalloc");
    if (rc == LT_OK && R > 0) ok(hipMemcpy(d_rays, rays, R * 12, hipMemcpyHostToDevice), "hipMemcpy");
    if (rc == LT_OK)
      rc = lt_rayset_create_dev(&p->rs, d_rays, (int)R, height, flags & (LT_TRACE_NORM_EXACT | LT_TRACE_NORM_AMD),
                                p->s_run);
    if (d_rays) (void)hipFree(d_rays);
  }
  p->slots.resize(depth);
  const size_t out_bytes = p->b3 + p->bc + 3 * p->b1 + 256;
  for (int k = 0; rc == LT_OK && k < depth; ++k) {
    hp_slot& sl = p->slots[k];
    rc = lt_scene_create(&sl.scene, device);
    if (rc != LT_OK) break;
    ok(hipMalloc((void**)&sl.out_dev, out_bytes), "hipMalloc");
    ok(hipHostMalloc((void**)&sl.out_pin, out_bytes, hipHostMallocDefault), "hipHostMalloc");
    ok(hipEventCreateWithFlags(&sl.ev_up, hipEventDisableTiming), "hipEventCreate");
    ok(hipEventCreateWithFlags(&sl.ev_run, hipEventDisableTiming), "hipEventCreate");
    ok(hipEventCreateWithFlags(&sl.ev_done, hipEventDisableTiming), "hipEventCreate");
  }
  if (rc != LT_OK) {
    lt_hostpipe_destroy(p);
    return rc;
  }
  for (int u = 0; u < p->n_up; ++u) p->worker[u] = std::thread(hp_uploader, p, u);
  p->launcher = std::thread(hp_launcher, p);
  *out = p;
  return LT_OK;
}

// copy a finished slot's images into the caller's arrays and free the slot; mu NOT held
static int hp_collect(lt_hostpipe* p, hp_slot& sl) {
  int rc;
  const double t0 = hp_now();
  p->trace[sl.job.ticket & 255][4] = t0;
  {
    std::unique_lock<std::mutex> lk(p->mu);
    p->cv_done.wait(lk, [&] { return sl.state >= 2; });
    rc = sl.rc;
    if (rc != LT_OK) lt_set_error("%s", sl.err);
  }
  if (rc == LT_OK) {
    const hipError_t e = hipEventSynchronize(sl.ev_done);
    if (e != hipSuccess) {
      lt_set_error("lt_hostpipe: hipEventSynchronize failed: %s", hipGetErrorString(e));
      rc = LT_ERR_HIP;
    }
  }
  static const bool no_copy = getenv("LIDARHIP_HOSTPIPE_DEBUG_NOCOPY") != nullptr;
  if (rc == LT_OK && !sl.job.direct_out && !no_copy) {
    const hp_job& j = sl.job;
    const size_t R = (size_t)p->n_rays;
    const size_t ncol = (p->flags & LT_TRACE_LABEL_IMAGE) ? R * 4 : R * 12;
    const char* h = sl.out_pin;
    if (j.endpoints) memcpy(j.endpoints, h, R * 12);
    if (j.endcolors) memcpy(j.endcolors, h + p->b3, ncol);
    if (j.range) memcpy(j.range, h + p->b3 + p->bc, R * 4);
    if (j.endrem) memcpy(j.endrem, h + p->b3 + p->bc + p->b1, R * 4);
    if (j.tri) memcpy(j.tri, h + p->b3 + p->bc + 2 * p->b1, R * 4);
  }
  if (rc != LT_OK) {
    for (int u = 0; u < p->n_up; ++u) (void)hipStreamSynchronize(p->s_up[u]);
    (void)hipStreamSynchronize(p->s_down);
  }  // nothing may still touch the caller's arrays
  hp_unpin_all(p, sl.job);
  {
    std::lock_guard<std::mutex> lk(p->mu);
    sl.state = 0;
    p->t_collect += hp_now() - t0;
    p->trace[sl.job.ticket & 255][5] = hp_now();
  }
  return rc;
}

extern "C" int lt_hostpipe_wait(lt_hostpipe* p, int ticket) {
  if (!p || ticket < 0) {
    lt_set_error("lt_hostpipe_wait: invalid argument");
    return LT_ERR_INVALID_ARG;
  }
  hp_slot& sl = p->slots[ticket % p->depth];
  {
    std::lock_guard<std::mutex> lk(p->mu);
    if (sl.state == 0 || sl.job.ticket != ticket) return LT_OK;  // already collected
  }
  return hp_collect(p, sl);
}

extern "C" int lt_hostpipe_submit(lt_hostpipe* p, const float* origin, const float* verts, const int* faces,
                                  const void* colors, int colors_are_u8, const float* rem, int n_verts, int n_faces,
                                  float* endpoints, int* endcolors, float* range, float* endrem, int* tri, int* ticket) {
  if (!p || !origin) {
    lt_set_error("lt_hostpipe_submit: NULL pipe / origin");
    return LT_ERR_INVALID_ARG;
  }
  LT_CHECK(lt_check_mesh_args("lt_hostpipe_submit", verts, faces, colors, rem, n_verts, n_faces));
  int t;
  {
    std::lock_guard<std::mutex> lk(p->mu);
    t = p->next_ticket;
  }
  hp_slot& sl = p->slots[t % p->depth];
  bool busy;
  {
    std::lock_guard<std::mutex> lk(p->mu);
    busy = sl.state != 0;
  }
  int rc_old = LT_OK;
  if (busy) rc_old = hp_collect(p, sl);  // the scan submitted `depth` tickets ago: its images go to its arrays now
  {
    std::lock_guard<std::mutex> lk(p->mu);
    hp_job& j = sl.job;
    j.ticket = t;
    memcpy(j.origin, origin, 3 * sizeof(float));
    j.verts = verts; j.faces = faces; j.colors = colors; j.rem = rem;
    j.colors_u8 = colors_are_u8 ? 1 : 0;
    j.n_verts = n_verts; j.n_faces = n_faces;
    j.endpoints = endpoints; j.endcolors = endcolors; j.range = range; j.endrem = endrem; j.tri = tri;
    j.n_pinned = 0;
    j.direct_out = 0;
    sl.state = 1;
    sl.rc = LT_OK;
    p->trace[t & 255][0] = hp_now();
    p->queue.push_back(t % p->depth);
    p->next_ticket = t + 1;
  }
  p->cv_work.notify_one();
  if (ticket) *ticket = t;
  return rc_old;
}

extern "C" int lt_hostpipe_flush(lt_hostpipe* p) {
  if (!p) {
    lt_set_error("lt_hostpipe_flush: NULL pipe");
    return LT_ERR_INVALID_ARG;
  }
  int rc = LT_OK, first, last;
  {
    std::lock_guard<std::mutex> lk(p->mu);
    last = p->next_ticket;
    first = last - p->depth < 0 ? 0 : last - p->depth;
  }
  for (int t = first; t < last; ++t) {
    const int r = lt_hostpipe_wait(p, t);
    if (r != LT_OK && rc == LT_OK) rc = r;
  }
  for (hp_slot& sl : p->slots) {  // deferred device-side errors (faces referencing vertices out of range)
    const int r = lt_scene_status(sl.scene);
    if (r != LT_OK && rc == LT_OK) rc = r;
  }
  return rc;
}
