// lt_mergemesh.hip -- the volume geometry of deform('mergemesh'), decided on the device.
//
// The reference clips ONE `voxel_bounds` array scan after scan (auxiliary/laserscan.py:957-962 on the array
// lidar_deform.py:321 hands every MultiSemLaserScan; auxiliary/fusion_lidar.py:33-37 then rewrites its upper bounds):
//
//     merged_bnds = np.rint(merged.get_bnds()).astype(int)                     laserscan.py:957
//     vol_bnds[:, 0] = np.maximum(vol_bnds[:, 0], merged_bnds[:, 0])           :961
//     vol_bnds[:, 1] = np.minimum(vol_bnds[:, 1], merged_bnds[:, 1])           :962
//     vol_dim = np.ceil((vol_bnds[:, 1] - vol_bnds[:, 0]) / voxel).astype(int) fusion_lidar.py:34
//     vol_bnds[:, 1] = vol_bnds[:, 0] + vol_dim * voxel                        :36  (an int array truncates)
//
// so the geometry of scan k's volume is STATE: it depends on every earlier scan of the sequence.  Round 5 read the kept
// points' bounds back (48 bytes, a stream synchronisation between projection and fusion) and ran these statements in numpy.
// Here the state lives in HBM: one single-thread kernel per scan applies the statements in stream order to the bounds the
// projection left on the device (lt_proj_images.bnds), and mirrors the result -- the bounds the volume is built from, the
// bounds left behind, the dimensions, a status -- into a pinned host record with an event behind it.  The host does not
// wait for it: it launches the scan's chain on the geometry of the PREVIOUS scan (the bounds only shrink and are integers
// after the clip: they settle within a few scans of a sequence) and compares when the chain's own read-back (the mesh
// sizes) has passed; a scan whose geometry moved is run again on the right volume.  float64 throughout: subtraction,
// division, ceil, multiplication, addition and rint are correctly rounded IEEE operations on both sides (-ffp-contract=off).
#include "lt_internal.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <condition_variable>
#include <mutex>

#define LT_MM_RING 64

struct lt_mm_state {
  int device;
  int is_int;            // the caller's bounds array has an integer dtype (config/lidar_transfer.yaml: ints)
  double voxel;
  double* vb;            // device [6]: the bounds state {xmin, xmax, ymin, ymax, zmin, zmax}
  lt_mm_geometry* dev_rec;   // device [LT_MM_RING]
  lt_mm_geometry* host_rec;  // pinned [LT_MM_RING]
  hipEvent_t ev[LT_MM_RING];
  int issued[LT_MM_RING];
  unsigned next;
  hipStream_t last_stream;
  // several chains (host threads, streams) of one sequence: the calls are made in the order of the scans' numbers
  std::mutex mu;
  std::condition_variable cv;
  int next_seq;
};

__global__ void k_mm_geometry(double* __restrict__ vb, const double* __restrict__ pb, int is_int, double voxel,
                              lt_mm_geometry* __restrict__ rec) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  lt_mm_geometry g;
  bool none = false;
  for (int a = 0; a < 3; ++a) none = none || !(pb[2 * a] <= pb[2 * a + 1]);  // (+inf, -inf): no point was kept
  for (int k = 0; k < 6; ++k) { g.bnds_given[k] = vb[k]; g.bnds_after[k] = vb[k]; }
  g.dim[0] = g.dim[1] = g.dim[2] = 0;
  g.status = 0;
  g.ticket = 0;
  g.reserved = 0;
  if (none) {  // numpy: "zero-size array to reduction operation minimum" before any statement touched the array
    g.status = 1;
    *rec = g;
    return;
  }
  for (int a = 0; a < 3; ++a) {
    const double m0 = (double)(long long)rint(pb[2 * a]), m1 = (double)(long long)rint(pb[2 * a + 1]);  // np.rint(..).astype(int)
    const double lo = vb[2 * a] > m0 ? vb[2 * a] : m0;          // np.maximum
    const double hi = vb[2 * a + 1] < m1 ? vb[2 * a + 1] : m1;  // np.minimum
    g.bnds_given[2 * a] = lo;
    g.bnds_given[2 * a + 1] = hi;
  }
  bool empty = false;
  for (int a = 0; a < 3; ++a) {
    const double d = ceil((g.bnds_given[2 * a + 1] - g.bnds_given[2 * a]) / voxel);
    g.dim[a] = (int)d;
    empty = empty || !(d > 0.0);
  }
  for (int k = 0; k < 6; ++k) g.bnds_after[k] = g.bnds_given[k];
  if (empty) {  // the reference dies in TSDFVolume.__init__ (a negative dimension) with the clip already in its array
    g.status = 2;
  } else {
    for (int a = 0; a < 3; ++a) {
      double hi = g.bnds_given[2 * a] + (double)g.dim[a] * voxel;
      if (is_int) hi = trunc(hi);  // assignment into an integer array
      g.bnds_after[2 * a + 1] = hi;
    }
  }
  for (int k = 0; k < 6; ++k) vb[k] = g.bnds_after[k];
  *rec = g;
}

extern "C" int lt_mm_state_create(lt_mm_state** out, const double* vol_bnds, int bounds_are_int, double voxel_size, int device) {
  if (!out || !vol_bnds || !(voxel_size > 0.0)) {
    lt_set_error("lt_mm_state_create: invalid argument");
    return LT_ERR_INVALID_ARG;
  }
  *out = nullptr;
  if (device < 0) LT_HIP(hipGetDevice(&device));
  LT_HIP(hipSetDevice(device));
  lt_mm_state* s = new (std::nothrow) lt_mm_state();
  if (!s) return LT_ERR_NO_MEMORY;
  s->vb = nullptr; s->dev_rec = nullptr; s->host_rec = nullptr; s->next = 0; s->last_stream = nullptr; s->next_seq = 0;
  for (int k = 0; k < LT_MM_RING; ++k) { s->ev[k] = nullptr; s->issued[k] = 0; }
  s->device = device;
  s->is_int = bounds_are_int ? 1 : 0;
  s->voxel = voxel_size;
  if (hipMalloc((void**)&s->vb, 6 * sizeof(double)) != hipSuccess ||
      hipMalloc((void**)&s->dev_rec, LT_MM_RING * sizeof(lt_mm_geometry)) != hipSuccess ||
      hipHostMalloc((void**)&s->host_rec, LT_MM_RING * sizeof(lt_mm_geometry), hipHostMallocDefault) != hipSuccess ||
      hipMemcpy(s->vb, vol_bnds, 6 * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipGetLastError();
    lt_set_error("lt_mm_state_create: allocation failed");
    lt_mm_state_destroy(s);
    return LT_ERR_NO_MEMORY;
  }
  for (int k = 0; k < LT_MM_RING; ++k)
    if (hipEventCreateWithFlags(&s->ev[k], hipEventDisableTiming) != hipSuccess) {
      lt_set_error("lt_mm_state_create: hipEventCreate failed");
      lt_mm_state_destroy(s);
      return LT_ERR_HIP;
    }
  *out = s;
  return LT_OK;
}

extern "C" int lt_mm_state_destroy(lt_mm_state* s) {
  if (!s) return LT_OK;
  (void)hipSetDevice(s->device);
  (void)hipDeviceSynchronize();
  if (s->vb) (void)hipFree(s->vb);
  if (s->dev_rec) (void)hipFree(s->dev_rec);
  if (s->host_rec) (void)hipHostFree(s->host_rec);
  for (int k = 0; k < LT_MM_RING; ++k)
    if (s->ev[k]) (void)hipEventDestroy(s->ev[k]);
  delete s;
  return LT_OK;
}

// new bounds (a new sequence: the reference starts a fresh process with the configured bounds per sequence,
// experiments/run_lidar_deform.sh), in stream order behind the geometry kernels already queued
extern "C" int lt_mm_state_reset(lt_mm_state* s, const double* vol_bnds, void* stream) {
  if (!s || !vol_bnds) {
    lt_set_error("lt_mm_state_reset: invalid argument");
    return LT_ERR_INVALID_ARG;
  }
  std::lock_guard<std::mutex> lock(s->mu);
  LT_HIP(hipSetDevice(s->device));
  s->next_seq = 0;  // (the scans of the new sequence are numbered from 0 again)
  if (s->next > 0) LT_HIP(hipStreamWaitEvent((hipStream_t)stream, s->ev[(s->next - 1) % LT_MM_RING], 0));
  LT_HIP(hipMemcpyAsync(s->vb, vol_bnds, 6 * sizeof(double), hipMemcpyHostToDevice, (hipStream_t)stream));
  LT_HIP(hipStreamSynchronize((hipStream_t)stream));  // (the caller's array is read before the call returns; once per sequence)
  s->last_stream = nullptr;  // (synchronised: the next call needs no event)
  return LT_OK;
}

// seq >= 0: this is scan `seq` of the sequence -- the call waits (on the host) until the scans before it have made theirs;
// seq < 0: the caller issues its scans in order itself.
extern "C" int lt_mm_geometry_dev(lt_mm_state* s, const double* point_bnds, int seq, int* ticket, void* stream) {
  if (!s || !ticket) {
    lt_set_error("lt_mm_geometry_dev: invalid argument");
    return LT_ERR_INVALID_ARG;
  }
  std::unique_lock<std::mutex> lock(s->mu);
  if (seq >= 0) s->cv.wait(lock, [&] { return s->next_seq >= seq; });
  struct turn_done {  // whatever happens below, the scans behind this one must get their turn
    lt_mm_state* s; int seq;
    ~turn_done() { if (seq >= 0 && s->next_seq == seq) { s->next_seq = seq + 1; s->cv.notify_all(); } }
  } td{s, seq};
  if (!point_bnds) {  // lt_mm_skip: a scan that failed before its projection only gives up its turn
    *ticket = -1;
    return LT_OK;
  }
  LT_HIP(hipSetDevice(s->device));
  const int slot = (int)(s->next % LT_MM_RING);
  if (s->issued[slot]) LT_HIP(hipEventSynchronize(s->ev[slot]));  // (64 scans behind: long done)
  // the statements are STATE: this call's kernel runs behind the previous call's, whatever stream that was issued on
  if (s->next > 0 && s->last_stream && s->last_stream != (hipStream_t)stream)
    LT_HIP(hipStreamWaitEvent((hipStream_t)stream, s->ev[(s->next - 1) % LT_MM_RING], 0));
  s->last_stream = (hipStream_t)stream;
  hipLaunchKernelGGL(k_mm_geometry, dim3(1), dim3(64), 0, (hipStream_t)stream, s->vb, point_bnds, s->is_int, s->voxel,
                     s->dev_rec + slot);
  LT_HIP(hipGetLastError());
  LT_HIP(hipMemcpyAsync(s->host_rec + slot, s->dev_rec + slot, sizeof(lt_mm_geometry), hipMemcpyDeviceToHost, (hipStream_t)stream));
  LT_HIP(hipEventRecord(s->ev[slot], (hipStream_t)stream));
  s->issued[slot] = 1;
  *ticket = (int)s->next;
  s->next++;
  return LT_OK;
}

extern "C" int lt_mm_geometry_get(lt_mm_state* s, int ticket, lt_mm_geometry* out) {
  if (!s || !out) {
    lt_set_error("lt_mm_geometry_get: invalid argument");
    return LT_ERR_INVALID_ARG;
  }
  {
    std::lock_guard<std::mutex> lock(s->mu);
    if (ticket < 0 || (unsigned)ticket >= s->next || s->next - (unsigned)ticket > LT_MM_RING) {
      lt_set_error("lt_mm_geometry_get: unknown ticket %d", ticket);
      return LT_ERR_INVALID_ARG;
    }
  }
  const int slot = ticket % LT_MM_RING;
  LT_HIP(hipEventSynchronize(s->ev[slot]));  // (passed already when the caller read the chain's mesh sizes)
  *out = s->host_rec[slot];
  out->ticket = ticket;
  out->reserved = 0;
  return LT_OK;
}
