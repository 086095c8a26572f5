#!/usr/bin/env python3
"""How long does the HOST need to enqueue one scan (build + trace)?  (launch-bound check)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidar_transfer_amd.laserscan import create_rays
from lidar_transfer_amd.raytracer import Scene
from lidar_transfer_amd.synth import WORKLOADS, synth_scene
wl = WORKLOADS["C2"]; dev = torch.device("cuda", 0)
v, f, c, r = synth_scene(0, wl["tris"])
mesh = [torch.from_numpy(x).to(dev) for x in (v, f, c, r)]
rays = torch.from_numpy(create_rays(wl["fov_up"], wl["fov_down"], wl["H"], wl["W"])).to(dev)
sc = Scene(0); out = sc.alloc_outputs(wl["H"] * wl["W"])
for _ in range(5):
    sc.set_mesh(*mesh); sc.build(); sc.trace(rays, (0, 0, 0), wl["H"], out=out)
torch.cuda.synchronize()
N = 100
t0 = time.perf_counter()
for _ in range(N):
    sc.set_mesh(*mesh); sc.build(); sc.trace(rays, (0, 0, 0), wl["H"], out=out)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/N:.3f} ms/scan, complete {1e3*(t2-t0)/N:.3f} ms/scan")
# python-side only (set_mesh is a pure host call)
t0 = time.perf_counter()
for _ in range(N):
    sc.set_mesh(*mesh)
print(f"set_mesh host call {1e6*(time.perf_counter()-t0)/N:.1f} us")
