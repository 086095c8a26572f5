// lt_api.hip -- C-ABI entry points of liblidarhip.so (declared in include/lidarhip.h).
#include "lt_internal.h"
#include <cpuid.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

static thread_local char g_err[512] = "";

void lt_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int lt_cu_count(int device) {
  static int cache[64];  // 0 = not asked yet (benign race: every writer stores the same value)
  const int slot = (device >= 0 && device < 64) ? device : 0;
  int v = __atomic_load_n(&cache[slot], __ATOMIC_RELAXED);
  if (v > 0) return v;
  hipDeviceProp_t prop;
  v = (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  __atomic_store_n(&cache[slot], v, __ATOMIC_RELAXED);
  return v;
}

extern "C" const char* lt_last_error(void) { return g_err; }
extern "C" const char* lt_version(void) { return "lidarhip 0.1 (gfx950)"; }
extern "C" int lt_abi_version(void) { return LT_ABI_VERSION; }

template <typename T>
static int dev_alloc(T** p, size_t count) {
  *p = nullptr;
  if (count == 0) count = 1;
  LT_HIP(hipMalloc((void**)p, count * sizeof(T)));
  return LT_OK;
}

static void ws_free(lt_scene* s) {
  for (int k = 0; k < 2; ++k) {
    if (s->keys[k]) (void)hipFree(s->keys[k]);
    if (s->vals[k]) (void)hipFree(s->vals[k]);
    s->keys[k] = s->vals[k] = nullptr;
  }
  if (s->hist) (void)hipFree(s->hist);
  if (s->tris) (void)hipFree(s->tris);
  if (s->seg) (void)hipFree(s->seg);
  if (s->nodes) (void)hipFree(s->nodes);
  if (s->nodes4) (void)hipFree(s->nodes4);
  s->nodes4 = nullptr;
  s->hist = nullptr; s->tris = nullptr; s->seg = nullptr; s->nodes = nullptr;
  s->cap_faces = 0;
}

int lt_scene_reserve(lt_scene* s, int n_faces) {
  if (n_faces <= s->cap_faces) return LT_OK;
  LT_HIP(hipSetDevice(s->device));
  if (s->last_stream || s->cap_faces) LT_HIP(hipDeviceSynchronize());
  ws_free(s);
  size_t cap = (size_t)n_faces + (size_t)n_faces / 4 + 1024;  // headroom: meshes of a sequence vary in size
  if (cap > (size_t)LT_MAX_FACES) cap = LT_MAX_FACES;
  size_t np = 1;
  while (np < cap) np <<= 1;
  const size_t nb = (cap + LT_SORT_TILE - 1) / LT_SORT_TILE;
  for (int k = 0; k < 2; ++k) {
    LT_CHECK(dev_alloc(&s->keys[k], cap));
    LT_CHECK(dev_alloc(&s->vals[k], cap));
  }
  LT_CHECK(dev_alloc(&s->hist, 1024 * nb + 1024));  // LT_RD digits x tiles + digit totals
  LT_CHECK(dev_alloc(&s->tris, 3 * cap));
  LT_CHECK(dev_alloc(&s->seg, 4 * np));
  LT_CHECK(dev_alloc(&s->nodes, 4 * cap));
  LT_CHECK(dev_alloc(&s->nodes4, 8 * cap));
  s->cap_faces = (int)cap;
  return LT_OK;
}

int lt_scene_reserve_rays(lt_scene* s, int n_rays) {
  if (n_rays <= s->cap_rays) return LT_OK;
  LT_HIP(hipSetDevice(s->device));
  if (s->overflow) {
    LT_HIP(hipDeviceSynchronize());
    (void)hipFree(s->overflow);
    s->overflow = nullptr;
  }
  LT_CHECK(dev_alloc(&s->overflow, (size_t)n_rays * (LT_STACK4_MAX - LT_STACK4_LDS)));  // >= the binary kernel's 32
  if (s->tail_stack) (void)hipFree(s->tail_stack);
  if (s->tail_meta) (void)hipFree(s->tail_meta);
  if (s->tail_queue) (void)hipFree(s->tail_queue);
  s->tail_stack = nullptr; s->tail_meta = nullptr; s->tail_queue = nullptr;
  LT_CHECK(dev_alloc(&s->tail_stack, (size_t)n_rays * LT_TAIL_SAVE));
  LT_CHECK(dev_alloc(&s->tail_meta, (size_t)n_rays));
  LT_CHECK(dev_alloc(&s->tail_queue, (size_t)n_rays));
  if (!s->tail_count) {
    LT_CHECK(dev_alloc(&s->tail_count, 4));
    LT_HIP(hipMemset(s->tail_count, 0, 4 * sizeof(int)));
  }
  s->cap_rays = n_rays;
  return LT_OK;
}

extern "C" int lt_scene_create(lt_scene** out, int device) {
  if (!out) {
    lt_set_error("lt_scene_create: NULL out pointer");
    return LT_ERR_INVALID_ARG;
  }
  *out = nullptr;
  if (device < 0) LT_HIP(hipGetDevice(&device));
  LT_HIP(hipSetDevice(device));
  lt_scene* s = (lt_scene*)calloc(1, sizeof(lt_scene));
  if (!s) {
    lt_set_error("lt_scene_create: out of host memory");
    return LT_ERR_NO_MEMORY;
  }
  s->device = device;
  int rc = dev_alloc(&s->partial, 6 * LT_BOUNDS_BLOCKS);
  if (rc == LT_OK) rc = dev_alloc(&s->params, 8);
  if (rc == LT_OK) rc = dev_alloc(&s->flags, 4);
  if (rc == LT_OK) rc = dev_alloc(&s->counters, 8 + 4 * LT_DBG_WAVES);
  if (rc == LT_OK && hipMemset(s->flags, 0, 4 * sizeof(unsigned)) != hipSuccess) rc = LT_ERR_HIP;
  for (int k = 0; rc == LT_OK && k < 10; ++k) {
    if (hipEventCreate(&s->ev[k]) != hipSuccess) {
      lt_set_error("hipEventCreate failed");
      rc = LT_ERR_HIP;
    } else {
      s->have_events = k + 1;
    }
  }
  if (rc != LT_OK) {
    lt_scene_destroy(s);
    return rc;
  }
  *out = s;
  return LT_OK;
}

extern "C" int lt_scene_destroy(lt_scene* s) {
  if (!s) return LT_OK;
  (void)hipSetDevice(s->device);
  (void)hipDeviceSynchronize();
  ws_free(s);
  if (s->owned_mesh) (void)hipFree(s->owned_mesh);
  if (s->partial) (void)hipFree(s->partial);
  if (s->params) (void)hipFree(s->params);
  if (s->flags) (void)hipFree(s->flags);
  if (s->counters) (void)hipFree(s->counters);
  if (s->overflow) (void)hipFree(s->overflow);
  if (s->tail_stack) (void)hipFree(s->tail_stack);
  if (s->tail_meta) (void)hipFree(s->tail_meta);
  if (s->tail_queue) (void)hipFree(s->tail_queue);
  if (s->tail_count) (void)hipFree(s->tail_count);
  if (s->sc_cell) (void)hipFree(s->sc_cell);
  if (s->sc_large) (void)hipFree(s->sc_large);
  if (s->sc_slices) (void)hipFree(s->sc_slices);
  if (s->sc_large_count) (void)hipFree(s->sc_large_count);
  for (int k = 0; k < s->have_events; ++k) (void)hipEventDestroy(s->ev[k]);
  free(s);
  return LT_OK;
}

int lt_check_mesh_args(const char* who, const void* verts, const void* faces, const void* colors,
                           const void* rem, int n_verts, int n_faces) {
  if (n_verts < 0 || n_faces < 0 || (n_faces > 0 && (!verts || !faces || !colors || !rem))) {
    lt_set_error("%s: invalid mesh (n_verts=%d n_faces=%d, NULL array?)", who, n_verts, n_faces);
    return LT_ERR_INVALID_ARG;
  }
  if (n_faces >= LT_MAX_FACES) {
    lt_set_error("%s: %d faces exceed LT_MAX_FACES", who, n_faces);
    return LT_ERR_TOO_LARGE;
  }
  return LT_OK;
}

extern "C" int lt_scene_set_mesh_dev(lt_scene* s, const float* verts, const int* faces, const int* colors,
                                     const float* rem, int n_verts, int n_faces) {
  if (!s) {
    lt_set_error("lt_scene_set_mesh_dev: NULL scene");
    return LT_ERR_INVALID_ARG;
  }
  LT_CHECK(lt_check_mesh_args("lt_scene_set_mesh_dev", verts, faces, colors, rem, n_verts, n_faces));
  s->verts = verts; s->faces = faces; s->colors = colors; s->rem = rem;
  s->n_verts = n_verts; s->n_faces = n_faces;
  s->built = 0;
  return LT_OK;
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }  // (also in lt_host.hip)

extern "C" int lt_scene_set_mesh_host(lt_scene* s, const float* verts, const int* faces, const int* colors,
                                      const float* rem, int n_verts, int n_faces, void* stream_) {
  if (!s) {
    lt_set_error("lt_scene_set_mesh_host: NULL scene");
    return LT_ERR_INVALID_ARG;
  }
  LT_CHECK(lt_check_mesh_args("lt_scene_set_mesh_host", verts, faces, colors, rem, n_verts, n_faces));
  hipStream_t stream = (hipStream_t)stream_;
  LT_HIP(hipSetDevice(s->device));
  const size_t bv = align256((size_t)n_verts * 12), bf = align256((size_t)n_faces * 12), bc = bv,
               br = align256((size_t)n_verts * 4);
  const size_t total = bv + bf + bc + br + 256;
  if (total > s->owned_mesh_bytes) {
    if (s->owned_mesh) {
      LT_HIP(hipDeviceSynchronize());
      (void)hipFree(s->owned_mesh);
      s->owned_mesh = nullptr;
      s->owned_mesh_bytes = 0;
    }
    LT_HIP(hipMalloc(&s->owned_mesh, total + total / 4));
    s->owned_mesh_bytes = total + total / 4;
  }
  char* base = (char*)s->owned_mesh;
  float* dv = (float*)base;
  int* df = (int*)(base + bv);
  int* dc = (int*)(base + bv + bf);
  float* dr = (float*)(base + bv + bf + bc);
  if (n_verts > 0) {
    LT_HIP(hipMemcpyAsync(dv, verts, (size_t)n_verts * 12, hipMemcpyHostToDevice, stream));
    if (colors) LT_HIP(hipMemcpyAsync(dc, colors, (size_t)n_verts * 12, hipMemcpyHostToDevice, stream));
    if (rem) LT_HIP(hipMemcpyAsync(dr, rem, (size_t)n_verts * 4, hipMemcpyHostToDevice, stream));
  }
  if (n_faces > 0) LT_HIP(hipMemcpyAsync(df, faces, (size_t)n_faces * 12, hipMemcpyHostToDevice, stream));
  s->verts = dv; s->faces = df; s->colors = dc; s->rem = dr;
  s->n_verts = n_verts; s->n_faces = n_faces;
  s->built = 0;
  s->last_stream = stream;
  return LT_OK;
}

extern "C" int lt_scene_build(lt_scene* s, void* stream, lt_stats* stats) {
  if (!s) {
    lt_set_error("lt_scene_build: NULL scene");
    return LT_ERR_INVALID_ARG;
  }
  LT_HIP(hipSetDevice(s->device));
  return lt_build_launch(s, (hipStream_t)stream, stats);
}

extern "C" int lt_scene_trace_dev(lt_scene* s, const float* rays, const float* origin, int n_rays, int height,
                                  float* endpoints, int* endcolors, float* range, float* endrem, int* tri,
                                  unsigned flags, void* stream, lt_stats* stats) {
  if (!s) {
    lt_set_error("lt_scene_trace_dev: NULL scene");
    return LT_ERR_INVALID_ARG;
  }
  LT_HIP(hipSetDevice(s->device));
  return lt_trace_launch(s, rays, origin, n_rays, height, endpoints, endcolors, range, endrem, tri, flags,
                         (hipStream_t)stream, stats);
}

// debug helper (not part of the documented ABI): wave start/end clocks of the last LT_TRACE_COUNT launch
extern "C" int lt_debug_wave_times(lt_scene* s, unsigned long long* out, int n_waves) {
  if (!s || !out || n_waves < 0 || n_waves > 2 * LT_DBG_WAVES) return LT_ERR_INVALID_ARG;
  LT_HIP(hipSetDevice(s->device));
  LT_HIP(hipDeviceSynchronize());
  LT_HIP(hipMemcpy(out, s->counters + 8, (size_t)n_waves * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return LT_OK;
}

// debug helper (not part of the documented ABI): compare lt_rcp_ieee with the IEEE division for every one of the
// 2^32 float bit patterns; returns the number of patterns where the bits differ in *mismatches, and how many
// patterns took the 5-instruction path in *fast
__global__ __launch_bounds__(256) void k_verify_rcp(unsigned long long* __restrict__ out) {
  unsigned long long bad = 0, fast = 0;
  const unsigned long long n = 1ull << 32;
  for (unsigned long long b = (unsigned long long)blockIdx.x * 256 + threadIdx.x; b < n;
       b += (unsigned long long)gridDim.x * 256) {
    const float a = __uint_as_float((unsigned)b);
    const float x = lt_rcp_ieee(a), y = 1.0f / a;
    if (__float_as_uint(x) != __float_as_uint(y) && !(x != x && y != y)) ++bad;
    if (fabsf(a) >= 5.421010862e-20f && fabsf(a) <= 1.8446744e19f) ++fast;
  }
  if (bad) atomicAdd(&out[0], bad);
  atomicAdd(&out[1], fast);
}

extern "C" int lt_debug_verify_rcp(unsigned long long* mismatches, unsigned long long* fast) {
  unsigned long long* d = nullptr;
  LT_HIP(hipMalloc((void**)&d, 2 * sizeof(unsigned long long)));
  LT_HIP(hipMemset(d, 0, 2 * sizeof(unsigned long long)));
  hipLaunchKernelGGL(k_verify_rcp, dim3(8192), dim3(256), 0, 0, d);
  unsigned long long h[2] = {0, 0};
  const hipError_t e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  LT_HIP(e);
  if (mismatches) *mismatches = h[0];
  if (fast) *fast = h[1];
  return LT_OK;
}

extern "C" int lt_scene_set_probe(lt_scene* s, void* ev_start, void* ev_stop) {
  if (!s) {
    lt_set_error("lt_scene_set_probe: NULL scene");
    return LT_ERR_INVALID_ARG;
  }
  s->probe[0] = (hipEvent_t)ev_start;
  s->probe[1] = (hipEvent_t)ev_stop;
  return LT_OK;
}

extern "C" int lt_scene_status(lt_scene* s) {
  if (!s) {
    lt_set_error("lt_scene_status: NULL scene");
    return LT_ERR_INVALID_ARG;
  }
  LT_HIP(hipSetDevice(s->device));
  LT_HIP(hipStreamSynchronize(s->last_stream));
  unsigned f = 0;
  LT_HIP(hipMemcpy(&f, s->flags, sizeof(f), hipMemcpyDeviceToHost));
  if (f) LT_HIP(hipMemset(s->flags, 0, sizeof(unsigned)));
  if (f & LT_FLAG_BAD_INDEX) {
    lt_set_error("mesh has faces referencing vertices outside [0, n_verts); those faces were ignored");
    return LT_ERR_BAD_INDEX;
  }
  if (f & LT_FLAG_HIER_OVERFLOW) {
    lt_set_error("LBVH build: a hierarchy window overflowed (internal error); rebuild with LIDARHIP_HIER=seg");
    return LT_ERR_HIP;
  }
  return LT_OK;
}

// LIDARHIP_NORMALIZE selects the 1/sqrt seed of the reference's normalize() (Vector3.h:83, _mm_rsqrt_ps -- its bits
// depend on the CPU vendor the reference runs on) for lt_ctrace:
//   intel (default)  replay the seed measured on GenuineIntel (the golden vectors were made on such a host)
//   amd              replay the seed measured on AuthenticAMD (EPYC 9575F, the MI355X box's host)
//   host             whichever of the two this process's CPU is: "the reference as it would run right here"
//   exact            correctly rounded 1/sqrt, vendor independent (LT_TRACE_NORM_EXACT)
unsigned lt_env_norm_flag() {
  const char* nm = getenv("LIDARHIP_NORMALIZE");
  if (!nm || !*nm || strcmp(nm, "intel") == 0) return 0u;
  if (strcmp(nm, "exact") == 0) return LT_TRACE_NORM_EXACT;
  if (strcmp(nm, "amd") == 0) return LT_TRACE_NORM_AMD;
  if (strcmp(nm, "host") == 0) {
    unsigned a = 0, b = 0, c = 0, d = 0;
    char vendor[13] = {0};
    if (__get_cpuid(0, &a, &b, &c, &d)) {
      memcpy(vendor, &b, 4); memcpy(vendor + 4, &d, 4); memcpy(vendor + 8, &c, 4);
    }
    return strcmp(vendor, "AuthenticAMD") == 0 ? LT_TRACE_NORM_AMD : 0u;
  }
  return 0u;
}
