#!/usr/bin/env python3
"""How long does the HOST need to enqueue one scan?  (launch-bound check for both strategies)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidar_transfer_amd import _lib
from lidar_transfer_amd.laserscan import create_rays
from lidar_transfer_amd.raytracer import RaySet, Scene
from lidar_transfer_amd.synth import WORKLOADS, synth_scene
wl = WORKLOADS["C2"]; dev = torch.device("cuda", 0)
v, f, c, r = synth_scene(0, wl["tris"])
mesh = [torch.from_numpy(x).to(dev) for x in (v, f, c, r)]
rays = torch.from_numpy(create_rays(wl["fov_up"], wl["fov_down"], wl["H"], wl["W"])).to(dev)
sc = Scene(0); rs = RaySet(rays, wl["H"]); out = sc.alloc_outputs(wl["H"] * wl["W"])
lib = _lib.load(); vp = C.c_void_p
org = (C.c_float * 3)(0, 0, 0)
margs = (vp(mesh[0].data_ptr()), vp(mesh[1].data_ptr()), vp(mesh[2].data_ptr()), vp(mesh[3].data_ptr()), mesh[0].numel() // 3, mesh[1].numel() // 3)
o = {k: vp(t.data_ptr()) for k, t in out.items()}
st = vp(torch.cuda.current_stream(dev).cuda_stream)
def step():
    lib.lt_scene_set_mesh_dev(sc._h, *margs)
    lib.lt_scene_render_dev(sc._h, rs._h, org, o["endpoints"], o["endcolors"], o["range"], o["endrem"], o["tri"], 1, st, None)
for _ in range(20): step()
torch.cuda.synchronize()
N = 400
t0 = time.perf_counter()
for _ in range(N): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"scatter: enqueue {1e6*(t1-t0)/N:.1f} us/scan, complete {1e6*(t2-t0)/N:.1f} us/scan")
