// pcie_probe.hip -- host<->device transfer rates on this box, the numbers the host-mesh pipeline is designed around:
// pageable vs pinned hipMemcpyAsync, H2D / D2H / both at once, and the host memcpy into a pinned staging buffer
// with 1..16 threads.      hipcc --offload-arch=gfx950 -O2 -o /tmp/pcie_probe tools/pcie_probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static void par_copy(char* dst, const char* src, size_t n, int nt) {
  std::vector<std::thread> th;
  const size_t chunk = (n + nt - 1) / nt;
  for (int t = 0; t < nt; ++t) {
    const size_t o = t * chunk;
    if (o >= n) break;
    th.emplace_back([=] { memcpy(dst + o, src + o, o + chunk > n ? n - o : chunk); });
  }
  for (auto& t : th) t.join();
}
int main() {
  const size_t N = 26 << 20, M = 4 << 20;  // a C2 mesh in, its images out
  char *pg = (char*)malloc(N), *pg2 = (char*)malloc(N), *pin, *pin2, *dev, *dev2;
  memset(pg, 1, N); memset(pg2, 2, N);
  CK(hipHostMalloc((void**)&pin, N, hipHostMallocDefault));
  CK(hipHostMalloc((void**)&pin2, N, hipHostMallocDefault));
  CK(hipMalloc((void**)&dev, N)); CK(hipMalloc((void**)&dev2, N));
  memset(pin, 1, N); memset(pin2, 1, N);
  hipStream_t s1, s2;
  CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  auto rate = [&](const char* what, size_t bytes, auto fn) {
    for (int i = 0; i < 3; ++i) fn();
    hipDeviceSynchronize();
    const int reps = 20;
    const double t0 = now();
    for (int i = 0; i < reps; ++i) fn();
    hipDeviceSynchronize();
    const double dt = (now() - t0) / reps;
    printf("%-58s %8.3f ms  %7.2f GB/s\n", what, dt * 1e3, bytes / dt / 1e9);
  };
  rate("H2D 26 MB pageable, hipMemcpyAsync", N, [&] { hipMemcpyAsync(dev, pg, N, hipMemcpyHostToDevice, s1); hipStreamSynchronize(s1); });
  rate("H2D 26 MB pinned, hipMemcpyAsync", N, [&] { hipMemcpyAsync(dev, pin, N, hipMemcpyHostToDevice, s1); hipStreamSynchronize(s1); });
  rate("D2H 4 MB pinned", M, [&] { hipMemcpyAsync(pin2, dev2, M, hipMemcpyDeviceToHost, s2); hipStreamSynchronize(s2); });
  rate("D2H 4 MB pageable", M, [&] { hipMemcpyAsync(pg2, dev2, M, hipMemcpyDeviceToHost, s2); hipStreamSynchronize(s2); });
  rate("H2D 26 MB + D2H 4 MB pinned, two streams", N, [&] {
    hipMemcpyAsync(dev, pin, N, hipMemcpyHostToDevice, s1); hipMemcpyAsync(pin2, dev2, M, hipMemcpyDeviceToHost, s2);
    hipStreamSynchronize(s1); hipStreamSynchronize(s2); });
  rate("H2D 26 MB + D2H 26 MB pinned, two streams", N, [&] {
    hipMemcpyAsync(dev, pin, N, hipMemcpyHostToDevice, s1); hipMemcpyAsync(pin2, dev2, N, hipMemcpyDeviceToHost, s2);
    hipStreamSynchronize(s1); hipStreamSynchronize(s2); });
  rate("H2D 4 x 6.5 MB pinned (four arrays of a mesh)", N, [&] {
    for (int k = 0; k < 4; ++k) hipMemcpyAsync(dev + k * (N / 4), pin + k * (N / 4), N / 4, hipMemcpyHostToDevice, s1);
    hipStreamSynchronize(s1); });
  // does the CALL return before the transfer is done?  (time inside hipMemcpyAsync vs time to completion)
  auto call_time = [&](const char* what, auto fn, hipStream_t s) {
    double tc = 0, tt = 0;
    for (int i = 0; i < 10; ++i) {
      hipDeviceSynchronize();
      const double t0 = now();
      fn();
      const double t1 = now();
      hipStreamSynchronize(s);
      tc += t1 - t0; tt += now() - t0;
    }
    printf("%-58s call returns after %7.3f ms, done after %7.3f ms\n", what, tc * 100, tt * 100);
  };
  call_time("H2D 26 MB pageable", [&] { hipMemcpyAsync(dev, pg, N, hipMemcpyHostToDevice, s1); }, s1);
  call_time("H2D 26 MB pinned", [&] { hipMemcpyAsync(dev, pin, N, hipMemcpyHostToDevice, s1); }, s1);
  call_time("D2H 4 MB pageable", [&] { hipMemcpyAsync(pg2, dev2, M, hipMemcpyDeviceToHost, s2); }, s2);
  call_time("D2H 4 MB pinned", [&] { hipMemcpyAsync(pin2, dev2, M, hipMemcpyDeviceToHost, s2); }, s2);
  {
    char* fresh = (char*)malloc(N);  // never touched: page faults inside the copy?
    const double t0 = now();
    hipMemcpyAsync(dev, fresh, N, hipMemcpyHostToDevice, s1); hipStreamSynchronize(s1);
    printf("%-58s %8.3f ms\n", "H2D 26 MB from a fresh untouched malloc", (now() - t0) * 1e3);
    memset(fresh, 3, N);
    const double t1 = now();
    hipMemcpyAsync(dev, fresh, N, hipMemcpyHostToDevice, s1); hipStreamSynchronize(s1);
    printf("%-58s %8.3f ms\n", "H2D 26 MB from a fresh malloc after memset (first copy)", (now() - t1) * 1e3);
    const double t2 = now();
    hipMemcpyAsync(dev, fresh, N, hipMemcpyHostToDevice, s1); hipStreamSynchronize(s1);
    printf("%-58s %8.3f ms\n", "   ... second copy", (now() - t2) * 1e3);
    free(fresh);
  }
  {
    // the four arrays of a C2 mesh (verts 6 MB, colours u8 1.5 MB, rem 2 MB, faces 12 MB), as separate mallocs
    const size_t sz[4] = {6 << 20, 3 << 19, 2 << 20, 12 << 20};
    char* a[4]; size_t off[4]; size_t o = 0;
    for (int k = 0; k < 4; ++k) { a[k] = (char*)malloc(sz[k] + 64) + 16; memset(a[k], k, sz[k]); off[k] = o; o += sz[k]; }
    rate("H2D 4 pageable arrays (6 + 1.5 + 2 + 12 MB), blocking calls", o, [&] {
      for (int k = 0; k < 4; ++k) hipMemcpyAsync(dev + off[k], a[k], sz[k], hipMemcpyHostToDevice, s1);
      hipStreamSynchronize(s1); });
    rate("   ... registered first, async copies, unregistered after", o, [&] {
      for (int k = 0; k < 4; ++k) hipHostRegister(a[k], sz[k], hipHostRegisterDefault);
      for (int k = 0; k < 4; ++k) hipMemcpyAsync(dev + off[k], a[k], sz[k], hipMemcpyHostToDevice, s1);
      hipStreamSynchronize(s1);
      for (int k = 0; k < 4; ++k) hipHostUnregister(a[k]); });
    double tr = 0, tu = 0;
    for (int i = 0; i < 10; ++i) {
      const double t0 = now();
      for (int k = 0; k < 4; ++k) hipHostRegister(a[k], sz[k], hipHostRegisterDefault);
      const double t1 = now();
      for (int k = 0; k < 4; ++k) hipHostUnregister(a[k]);
      tr += t1 - t0; tu += now() - t1;
    }
    printf("%-58s register %7.3f ms, unregister %7.3f ms\n", "hipHostRegister of the four arrays (21.5 MB)", tr * 100, tu * 100);
  }
  {
    // uploads of one thread while ANOTHER thread downloads (the host pipe's situation)
    const size_t sz[4] = {6 << 20, 3 << 19, 2 << 20, 12 << 20};
    char* a[4]; size_t off[4]; size_t o = 0;
    for (int k = 0; k < 4; ++k) { a[k] = (char*)malloc(sz[k] + 64) + 16; memset(a[k], k, sz[k]); off[k] = o; o += sz[k]; }
    for (int mode = 0; mode < 6; ++mode) {
      // mode 0: uploads alone (pageable, blocking)      1: + concurrent pinned D2H 4.7 MB per upload set
      // mode 2: registered async uploads alone          3: registered + concurrent D2H
      // mode 4: pageable uploads + D2H on the SAME thread after each set (serial)
      // mode 5: registered async uploads, D2H issued on the same thread into a second stream (no second thread)
      const bool reg = mode == 2 || mode == 3 || mode == 5, second = mode == 1 || mode == 3;
      if (reg) for (int k = 0; k < 4; ++k) hipHostRegister(a[k], sz[k], hipHostRegisterDefault);
      volatile bool stop = false;
      std::thread other;
      if (second) other = std::thread([&] {
        while (!stop) { hipMemcpyAsync(pin2, dev2, 4700000, hipMemcpyDeviceToHost, s2); hipStreamSynchronize(s2);
                        std::this_thread::sleep_for(std::chrono::microseconds(300)); } });
      const int reps = 40;
      hipDeviceSynchronize();
      const double t0 = now();
      for (int i = 0; i < reps; ++i) {
        for (int k = 0; k < 4; ++k) hipMemcpyAsync(dev + off[k], a[k], sz[k], hipMemcpyHostToDevice, s1);
        if (mode == 4) { hipMemcpyAsync(pin2, dev2, 4700000, hipMemcpyDeviceToHost, s2); hipStreamSynchronize(s2); }
        if (mode == 5) { hipMemcpyAsync(pin2, dev2, 4700000, hipMemcpyDeviceToHost, s2); }
        if (reg && (i & 1)) hipStreamSynchronize(s1);
      }
      hipStreamSynchronize(s1); hipStreamSynchronize(s2);
      const double dt = (now() - t0) / reps;
      stop = true;
      if (second) other.join();
      if (reg) for (int k = 0; k < 4; ++k) hipHostUnregister(a[k]);
      printf("pipe situation, mode %d: %8.3f ms per 21.5 MB upload set  %7.2f GB/s\n", mode, dt * 1e3, o / dt / 1e9);
    }
  }
  for (int nt : {1, 2, 4, 8, 16})  {
    char buf[96]; snprintf(buf, sizeof buf, "host memcpy 26 MB pageable -> pinned, %d thread(s)", nt);
    rate(buf, N, [&] { par_copy(pin, pg, N, nt); });
  }
  rate("hipHostRegister + H2D + hipHostUnregister, 26 MB", N, [&] {
    hipHostRegister(pg, N, hipHostRegisterDefault); hipMemcpyAsync(dev, pg, N, hipMemcpyHostToDevice, s1);
    hipStreamSynchronize(s1); hipHostUnregister(pg); });
  return 0;
}
