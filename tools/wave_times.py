#!/usr/bin/env python3
"""Debug: distribution of per-wave start / end clocks of k_trace on the C2 workload."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lidar_transfer_amd import _lib
from lidar_transfer_amd.laserscan import create_rays
from lidar_transfer_amd.raytracer import Scene
from lidar_transfer_amd.synth import WORKLOADS, synth_scene
wl = WORKLOADS["C2"]; dev = torch.device("cuda", 0)
v, f, c, r = synth_scene(0, wl["tris"])
mesh = [torch.from_numpy(x).to(dev) for x in (v, f, c, r)]
rays = torch.from_numpy(create_rays(wl["fov_up"], wl["fov_down"], wl["H"], wl["W"])).to(dev)
sc = Scene(0); out = sc.alloc_outputs(wl["H"] * wl["W"])
sc.set_mesh(*mesh); sc.build()
for _ in range(3):
    o = sc.trace(rays, (0, 0, 0), wl["H"], out=out, count=True, stats=True)
nw = 2048
buf = np.zeros(2 * nw, np.uint64)
lib = _lib.load()
lib.lt_debug_wave_times.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
assert lib.lt_debug_wave_times(sc._h, buf.ctypes.data_as(C.c_void_p), nw) == 0
t = buf.reshape(nw, 2).astype(np.int64)
t0 = t[:, 0].min()
start = t[:, 0] - t0; end = t[:, 1] - t0; dur = end - start
print("trace ms", o["stats"]["ms_trace"], "clock span", end.max())
print("start offset: p50 %d p90 %d max %d" % (np.percentile(start, 50), np.percentile(start, 90), start.max()))
print("duration: mean %d p50 %d p90 %d p99 %d max %d" % (dur.mean(), np.percentile(dur, 50), np.percentile(dur, 90), np.percentile(dur, 99), dur.max()))
print("end: p50 %d p90 %d p99 %d max %d" % (np.percentile(end, 50), np.percentile(end, 90), np.percentile(end, 99), end.max()))
order = np.argsort(-dur)[:8]
print("slowest waves (wave id, start, dur):", [(int(i), int(start[i]), int(dur[i])) for i in order])
