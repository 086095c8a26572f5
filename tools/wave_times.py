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

# ---- quad traversal: production kernel with the internal timing flag (wall clock at 100 MHz per wave)
if "--quad" in sys.argv:
    nw = 8192
    org = (C.c_float * 3)(0, 0, 0)
    R = wl["H"] * wl["W"]

    def run(flags):
        assert lib.lt_scene_trace_dev(sc._h, rays.data_ptr(), org, R, wl["H"], out["endpoints"].data_ptr(),
                                      out["endcolors"].data_ptr(), out["range"].data_ptr(), out["endrem"].data_ptr(),
                                      out["tri"].data_ptr(), flags, None, None) == 0
        torch.cuda.synchronize()

    for _ in range(3):
        run(1 | 0x8000)
    buf = np.zeros(2 * nw, np.uint64)
    assert lib.lt_debug_wave_times(sc._h, buf.ctypes.data_as(C.c_void_p), nw) == 0
    t = buf.reshape(nw, 2)
    start = (t[:, 0] - t[:, 0].min()).astype(np.int64) / 100.0            # us
    dur = (t[:, 1] & np.uint64(0xFFFFFFFF)).astype(np.int64) / 100.0        # us
    end = start + dur
    print("k_trace4 (step cap %s): waves %d, span of the main kernel %.1f us" % (os.environ.get("LIDARHIP_STEP_CAP", "default"), nw, end.max()))
    for name, a in (("start", start), ("duration", dur), ("end", end)):
        print("  %-16s mean %8.1f p50 %8.1f p90 %8.1f p99 %8.1f max %8.1f" % (name, a.mean(), np.percentile(a, 50), np.percentile(a, 90), np.percentile(a, 99), a.max()))
    print("  busy fraction of the chip over the span: %.3f" % (dur.sum() / (nw * end.max())))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ms = []
    for _ in range(20):
        e0.record(); run(1); e1.record(); torch.cuda.synchronize(); ms.append(e0.elapsed_time(e1))
    print("  trace (main + tail kernel), events around the call: median %.1f us" % (np.median(ms) * 1e3))
    # steps per ray (internal flag 0x4000, no cap): where do the long walks live?
    run(1 | 0x4000)
    steps = out["tri"].cpu().numpy().reshape(wl["H"], wl["W"])
    print("  steps per ray (node + leaf): mean %.1f p50 %d p90 %d p98 %d p99 %d p99.9 %d max %d" % (steps.mean(), *[np.percentile(steps, p) for p in (50, 90, 98, 99, 99.9)], steps.max()))
    for cap in (32, 40, 48, 64):
        print("    rays beyond %d steps: %.2f %%" % (cap, 100.0 * (steps > cap).mean()))

# ---- k_hierarchy4 (LIDARHIP_DEBUG_HIER=1): wall-clock stamps per wave: whole kernel, and end of staging / phase 1
if "--hier" in sys.argv:
    sc.set_mesh(*mesh); sc.build(); torch.cuda.synchronize()
    nw = 3908
    buf = np.zeros(4 * 16384, np.uint64)
    lib.lt_debug_wave_times.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    assert lib.lt_debug_wave_times(sc._h, buf.ctypes.data_as(C.c_void_p), 2 * 16384) == 0
    t = buf[:2 * nw].reshape(nw, 2)
    ph = buf[2 * 16384:2 * 16384 + 2 * nw].reshape(nw, 2).astype(np.int64) / 100.0
    start = (t[:, 0] - t[:, 0].min()).astype(np.int64) / 100.0
    dur = t[:, 1].astype(np.int64) / 100.0
    end = start + dur
    print("k_hierarchy4: waves %d, span %.1f us" % (nw, end.max()))
    for name, a in (("start", start), ("phase 1 done at", ph[:, 0]), ("splits done at", ph[:, 1]), ("duration", dur), ("end", end)):
        print("  %-16s mean %8.1f p50 %8.1f p90 %8.1f p99 %8.1f max %8.1f" % (name, a.mean(), *[np.percentile(a, p) for p in (50, 90, 99)], a.max()))
    big = buf[8192:8192 + 6 * 4096].reshape(4096, 6).astype(np.int64)
    big = big[big[:, 0] != 0]
    b0 = (big[:, 0] - big[:, 0].min()) / 100.0
    print("k_hierarchy4_big: %d queued nodes stamped (of at most 4096), starts span %.1f us" % (len(big), b0.max()))
    for name, a in (("start", b0), ("range done at", big[:, 1] / 100.0), ("split done at", big[:, 2] / 100.0), ("child splits at", big[:, 3] / 100.0),
                    ("duration", big[:, 4] / 100.0), ("end", b0 + big[:, 4] / 100.0), ("span (leaves)", big[:, 5].astype(float))):
        print("  %-16s mean %8.1f p50 %8.1f p90 %8.1f p99 %8.1f max %8.1f" % (name, a.mean(), *[np.percentile(a, p) for p in (50, 90, 99)], a.max()))
