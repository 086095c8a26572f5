#!/usr/bin/env python3
"""Throughput of the pipelined host-buffer path (lt_hostpipe) on C2 and where its time goes (one JSON line)."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lidar_transfer_amd import _lib
from lidar_transfer_amd.laserscan import create_rays
from lidar_transfer_amd.pipeline import HostScanPipeline
from lidar_transfer_amd.synth import WORKLOADS, synth_scene
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n_meshes = int(sys.argv[3]) if len(sys.argv) > 3 else 4      # distinct host meshes cycled
few_out = len(sys.argv) > 4 and sys.argv[4] == "range"        # download the range image only
wl = WORKLOADS["C2"]; H, W = wl["H"], wl["W"]
host = []
for i in range(n_meshes):
    v, f, c, r = synth_scene(i, wl["tris"])
    host.append((v, f, (c & 255).astype(np.uint8), r))
rays = create_rays(wl["fov_up"], wl["fov_down"], H, W); org = np.zeros(3, np.float32)
lib = _lib.load()
if os.environ.get("LT_RATE_PINNED") == "1":   # experiment: the meshes in pinned host memory (lt_host_alloc) instead of numpy's pageable
    def pinned_copy(a):
        p = C.c_void_p(); _lib.check(lib.lt_host_alloc(C.byref(p), a.nbytes), "lt_host_alloc")
        b = np.frombuffer((C.c_char * a.nbytes).from_address(p.value), dtype=a.dtype).reshape(a.shape)
        b[...] = a
        return b
    host = [tuple(pinned_copy(a) for a in m) for m in host]
lib.lt_debug_hostpipe_times.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
with HostScanPipeline(rays, H, depth=depth) as pipe:
    outs = [pipe.alloc_outputs() for _ in range(depth + 1)]
    if few_out:
        outs = [{"range": o["range"]} for o in outs]
    for o in outs:
        for a in o.values(): a.fill(0)
    # warm-up.  100 scans, not 8: the HIP 7.0 runtime bundled with the torch wheel stalls ONCE, for 36-54 ms, at the 81st scan
    # of a process (its ~1000th queued command; the system ROCm 7.2 runtime does not) -- profiles/r05/hostpipe_torch.txt.  A
    # sequence of 4541 scans pays that once; a 200-scan measurement that contains it reports 0.61 instead of 0.43 ms per scan.
    n_warm = int(os.environ.get("LT_RATE_WARMUP", "100"))
    tw0 = time.perf_counter(); wstamps = []
    wtick = []
    for k in range(n_warm):
        wtick.append(pipe.submit(*host[k % n_meshes], org, out=outs[k % len(outs)]))
        if k >= depth: pipe.wait(wtick[k - depth])
        wstamps.append(time.perf_counter())
    pipe.flush()
    wg = np.diff(np.array([tw0] + wstamps)) * 1e3 if n_warm else np.zeros(1)
    t0d = (C.c_double * 4)(); lib.lt_debug_hostpipe_times(pipe._h, t0d)
    t0 = time.perf_counter(); tick = []
    stamps = []
    for k in range(n):
        tick.append(pipe.submit(*host[k % n_meshes], org, out=outs[k % len(outs)]))
        if k >= depth - 1: pipe.wait(tick[k - depth + 1])
        stamps.append(time.perf_counter())
    pipe.flush()
    dt = (time.perf_counter() - t0) / n
    g = np.diff(np.array([t0] + stamps)) * 1e3
    if os.environ.get("LT_RATE_GAPS") == "1":   # where the time of the loop goes: per-iteration gaps
        big = np.flatnonzero(g > 3 * np.median(g))
        sys.stderr.write("gaps ms: median %.3f mean %.3f p90 %.3f max %.3f; iterations > 3 x median: %s\n" % (
            np.median(g), g.mean(), np.percentile(g, 90), g.max(), [(int(i), round(float(g[i]), 2)) for i in big[:20]]))
    t1d = (C.c_double * 4)(); lib.lt_debug_hostpipe_times(pipe._h, t1d)
    tr = (C.c_double * (256 * 6))(); lib.lt_debug_hostpipe_trace.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    lib.lt_debug_hostpipe_trace(pipe._h, tr)
    tr = np.array(tr).reshape(256, 6)
    last = (n_warm + n - 1) & 255
    rows = [(last - 12 + i) & 255 for i in range(10)]
    base = tr[rows[0], 0]
    for rI in rows:
        sys.stderr.write("ticket%%256=%3d submit %7.3f | issue %7.3f upload_end %7.3f issue_end %7.3f | collect %7.3f .. %7.3f\n" % ((rI,) + tuple((tr[rI] - base) * 1e3)))
# the single drop-in call in the same process (for comparison: same runtime, same arrays, int32 colours as ctrace takes them)
from lidar_transfer_amd.raytracer import C_Trace
v, f, c8, r = host[0]
c32 = np.ascontiguousarray(c8.astype(np.int32)).reshape(-1)
R = H * W
ts = []
for i in range(14):
    ep = np.zeros(3 * R, np.float32); ec = np.zeros(3 * R, np.int32); rg = np.zeros(R, np.float32); rm = np.zeros(R, np.float32)
    t0s = time.perf_counter()
    C_Trace(rays.reshape(-1), org, v.reshape(-1), f.reshape(-1), c32, r, ep, ec, rg, rm, H, W)
    ts.append(time.perf_counter() - t0s)
single_ms = float(np.median(ts[2:])) * 1e3
h2d = sum(a.nbytes for a in host[0])
d2h = sum(a.nbytes for a in outs[0].values())
hits = int((outs[(n - 1) % len(outs)]["range"] > 0).sum())
print(json.dumps({"single_call_ms": round(single_ms, 4), "h2d_bytes": h2d, "d2h_bytes": d2h, "hits": hits, "n_scans": n, "torch_in_process": "torch" in sys.modules,
                  "depth": depth, "meshes": n_meshes, "outputs": "range" if few_out else "all", "ms_per_scan": round(dt * 1e3, 4), "h2d_MB": round(h2d / 1e6, 2), "GBs": round(h2d / dt / 1e9, 2),
                  "worker_issue_ms": round((t1d[1] - t0d[1]) / n * 1e3, 4), "worker_upload_ms": round((t1d[2] - t0d[2]) / n * 1e3, 4),
                  "caller_collect_ms": round((t1d[3] - t0d[3]) / n * 1e3, 4),
                  # the same loop INCLUDING the runtime's one-time stall (it lies inside the warm-up): what a process that renders
                  # only warmup_scans + n_scans scans pays per scan
                  "ms_per_scan_incl_warmup": round(((wstamps[-1] - tw0 if n_warm else 0.0) + dt * n) / (n_warm + n) * 1e3, 4),
                  "warmup_scans": n_warm, "warmup_longest_gap_ms": round(float(wg.max()), 3), "warmup_longest_gap_at_scan": int(wg.argmax()),
                  "timed_longest_gap_ms": round(float(g.max()), 3)}))
