#!/usr/bin/env python3
"""PCIe-inclusive drop-in call: C_Trace(numpy in, numpy out) on workload C2 (one JSON line)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lidar_transfer_amd.laserscan import create_rays
from lidar_transfer_amd.raytracer import C_Trace
from lidar_transfer_amd.synth import WORKLOADS, synth_scene
wl = WORKLOADS["C2"]; H, W = wl["H"], wl["W"]; n = H * W
v, f, c, r = synth_scene(0, wl["tris"])
rays = create_rays(wl["fov_up"], wl["fov_down"], H, W).reshape(-1); org = np.zeros(3, np.float32)
vv, ff, cc = v.reshape(-1), f.reshape(-1), c.reshape(-1)
ts = []
for i in range(12):
    ep = np.zeros(3 * n, np.float32); ec = np.zeros(3 * n, np.int32); rg = np.zeros(n, np.float32); rm = np.zeros(n, np.float32)
    t = time.perf_counter()
    C_Trace(rays, org, vv, ff, cc, r, ep, ec, rg, rm, H, W)
    ts.append(time.perf_counter() - t)
t = float(np.median(ts[2:]))
print(json.dumps({"metric": "C_Trace drop-in call, host buffers in and out (PCIe inclusive)", "workload": "C2",
                  "ms_per_call": round(t * 1e3, 3), "Mrays_per_s": round(n / t / 1e6, 2), "hits": int((rg > 0).sum()),
                  "strategy": os.environ.get("LIDARHIP_STRATEGY", "scatter")}))
