// hostpipe_probe.cpp -- drive lt_hostpipe from a plain C++ host (no Python): ms per scan on a C2-sized grid mesh.
//   g++ -O2 -I include -o /tmp/hostpipe_probe tools/hostpipe_probe.cpp -L lidar_transfer_amd/lib -llidarhip -Wl,-rpath,$PWD/lidar_transfer_amd/lib -lpthread
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "lidarhip.h"
extern "C" int lt_debug_hostpipe_times(lt_hostpipe* p, double* out);
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  const int depth = argc > 1 ? atoi(argv[1]) : 3, n = argc > 2 ? atoi(argv[2]) : 200, n_meshes = 4;
  const int H = 64, W = 2048, R = H * W, G = 708;  // (G-1)^2 * 2 ~ 1.0 M triangles, G^2 ~ 0.5 M vertices
  std::vector<float> rays(3 * R);
  for (int h = 0; h < H; ++h)
    for (int w = 0; w < W; ++w) {
      const double yaw = -M_PI + 2 * M_PI * w / W, pitch = (3.0 - 28.0 * h / (H - 1)) * M_PI / 180;
      rays[3 * (h * W + w)] = (float)(cos(pitch) * cos(yaw)); rays[3 * (h * W + w) + 1] = (float)(cos(pitch) * sin(yaw));
      rays[3 * (h * W + w) + 2] = (float)sin(pitch);
    }
  struct mesh { std::vector<float> v, r; std::vector<int> f; std::vector<unsigned char> c; };
  std::vector<mesh> ms(n_meshes);
  for (int m = 0; m < n_meshes; ++m) {
    mesh& M = ms[m];
    M.v.resize(3 * G * G); M.r.resize(G * G); M.c.resize(3 * G * G); M.f.resize(6 * (G - 1) * (G - 1));
    for (int i = 0; i < G; ++i)
      for (int j = 0; j < G; ++j) {
        const int k = i * G + j;
        M.v[3 * k] = -50.f + 100.f * i / (G - 1); M.v[3 * k + 1] = -50.f + 100.f * j / (G - 1);
        M.v[3 * k + 2] = -1.7f + 0.1f * sinf(0.3f * i + m) * cosf(0.2f * j);
        M.r[k] = 0.5f; M.c[3 * k] = 1; M.c[3 * k + 1] = 2; M.c[3 * k + 2] = 40;
      }
    int t = 0;
    for (int i = 0; i + 1 < G; ++i)
      for (int j = 0; j + 1 < G; ++j) {
        const int a = i * G + j, b = a + G, c = b + 1, d = a + 1;
        M.f[t++] = a; M.f[t++] = b; M.f[t++] = c; M.f[t++] = a; M.f[t++] = c; M.f[t++] = d;
      }
  }
  lt_hostpipe* p = nullptr;
  if (lt_hostpipe_create(&p, rays.data(), R, H, depth, 0, -1) != LT_OK) { printf("create: %s\n", lt_last_error()); return 1; }
  const int NO = depth + 1;
  std::vector<std::vector<float>> ep(NO, std::vector<float>(3 * R)), rg(NO, std::vector<float>(R)), rm(NO, std::vector<float>(R));
  std::vector<std::vector<int>> ec(NO, std::vector<int>(3 * R)), tr(NO, std::vector<int>(R));
  const float org[3] = {0, 0, 0};
  std::vector<int> tick(n + 16);
  auto submit = [&](int k) {
    mesh& M = ms[k % n_meshes];
    const int o = k % NO;
    if (lt_hostpipe_submit(p, org, M.v.data(), M.f.data(), M.c.data(), 1, M.r.data(), G * G, (int)M.f.size() / 3, ep[o].data(),
                           ec[o].data(), rg[o].data(), rm[o].data(), tr[o].data(), &tick[k]) != LT_OK)
      printf("submit: %s\n", lt_last_error());
  };
  for (int k = 0; k < 8; ++k) submit(k);
  lt_hostpipe_flush(p);
  double d0[4], d1[4];
  lt_debug_hostpipe_times(p, d0);
  const double t0 = now();
  for (int k = 0; k < n; ++k) {
    submit(k);
    if (k >= depth - 1) lt_hostpipe_wait(p, tick[k - depth + 1]);
  }
  lt_hostpipe_flush(p);
  const double dt = (now() - t0) / n;
  lt_debug_hostpipe_times(p, d1);
  int hits = 0;
  for (int i = 0; i < R; ++i) hits += rg[(n - 1) % NO][i] > 0;
  const double mb = (ms[0].v.size() * 4 + ms[0].f.size() * 4 + ms[0].c.size() + ms[0].r.size() * 4) / 1e6;
  printf("{\"host\": \"c++\", \"depth\": %d, \"ms_per_scan\": %.4f, \"h2d_MB\": %.2f, \"GBs\": %.2f, \"worker_issue_ms\": %.4f, "
         "\"worker_upload_ms\": %.4f, \"caller_collect_ms\": %.4f, \"hits\": %d}\n", depth, dt * 1e3, mb, mb / dt / 1e3,
         (d1[1] - d0[1]) / n * 1e3, (d1[2] - d0[2]) / n * 1e3, (d1[3] - d0[3]) / n * 1e3, hits);
  lt_hostpipe_destroy(p);
  return 0;
}
