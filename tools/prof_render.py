#!/usr/bin/env python3
"""rocprofv3 driver: scatter-strategy render of one workload scene N times."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidar_transfer_amd.laserscan import create_rays
from lidar_transfer_amd.raytracer import RaySet, Scene
from lidar_transfer_amd.synth import WORKLOADS, synth_scene
ap = argparse.ArgumentParser(); ap.add_argument("--workload", default="C2"); ap.add_argument("--reps", type=int, default=10)
a = ap.parse_args()
wl = WORKLOADS[a.workload]; dev = torch.device("cuda", 0)
v, f, c, r = synth_scene(0, wl["tris"])
mesh = [torch.from_numpy(x).to(dev) for x in (v, f, c, r)]
rays = torch.from_numpy(create_rays(wl["fov_up"], wl["fov_down"], wl["H"], wl["W"])).to(dev)
sc = Scene(0); rs = RaySet(rays, wl["H"]); out = sc.alloc_outputs(wl["H"] * wl["W"])
for i in range(a.reps):
    sc.set_mesh(*mesh)
    o = sc.render(rs, (0.0, 0.0, 0.0), out=out, count=(i == a.reps - 1), stats=(i == a.reps - 1))
torch.cuda.synchronize()
print(o["stats"])
