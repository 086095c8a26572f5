#!/usr/bin/env python3
"""Small driver for rocprofv3 runs: build + trace one workload scene N times (no timing logic)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from lidar_transfer_amd.laserscan import create_rays  # noqa: E402
from lidar_transfer_amd.raytracer import Scene  # noqa: E402
from lidar_transfer_amd.synth import WORKLOADS, synth_scene  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="C2")
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--count", action="store_true")
a = ap.parse_args()
wl = WORKLOADS[a.workload]
dev = torch.device("cuda", 0)
v, f, c, r = synth_scene(0, wl["tris"])
mesh = [torch.from_numpy(x).to(dev) for x in (v, f, c, r)]
rays = torch.from_numpy(create_rays(wl["fov_up"], wl["fov_down"], wl["H"], wl["W"])).to(dev)
sc = Scene(0)
out = sc.alloc_outputs(wl["H"] * wl["W"])
for i in range(a.reps):
    sc.set_mesh(*mesh)
    st = sc.build(stats=(i == a.reps - 1))
    o = sc.trace(rays, (0.0, 0.0, 0.0), wl["H"], out=out, count=a.count and i == a.reps - 1, stats=(i == a.reps - 1))
torch.cuda.synchronize()
print({k: (round(x, 4) if isinstance(x, float) else x) for k, x in st.items()})
print({k: (round(x, 4) if isinstance(x, float) else x) for k, x in o["stats"].items()})
