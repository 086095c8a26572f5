#!/usr/bin/env python3
"""Host + command-processor floor of one bench step: the bench's step loop (16 streams, probe events) on a mesh
so small that the kernels take no time.  If this is close to the bench's ms_per_step, the bench is launch-bound."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidar_transfer_amd import _lib
from lidar_transfer_amd.laserscan import create_rays
from lidar_transfer_amd.raytracer import RaySet, Scene
from lidar_transfer_amd.synth import WORKLOADS, synth_scene
wl = WORKLOADS["C2"]; dev = torch.device("cuda", 0)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 16
tris = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
v, f, c, r = synth_scene(0, tris)
mesh = [torch.from_numpy(x).to(dev) for x in (v, f, c, r)]
rays = torch.from_numpy(create_rays(wl["fov_up"], wl["fov_down"], wl["H"], wl["W"])).to(dev)
streams = [torch.cuda.Stream(dev) for _ in range(S)]
scs = [Scene(0) for _ in range(S)]; rss = [RaySet(rays, wl["H"]) for _ in range(S)]
outs = [scs[0].alloc_outputs(wl["H"] * wl["W"]) for _ in range(S)]
lib = _lib.load(); vp = C.c_void_p
org = (C.c_float * 3)(0, 0, 0)
margs = (vp(mesh[0].data_ptr()), vp(mesh[1].data_ptr()), vp(mesh[2].data_ptr()), vp(mesh[3].data_ptr()), mesh[0].numel() // 3, mesh[1].numel() // 3)
N = 2000
probes = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(N)]
for a, b in probes: a.record(); b.record()
pr = [(vp(a.cuda_event), vp(b.cuda_event)) for a, b in probes]
sh = [vp(s.cuda_stream) for s in streams]
os_ = [{k: vp(t.data_ptr()) for k, t in o.items()} for o in outs]
def step(i, probe):
    s = i % S; h = scs[s]._h; o = os_[s]
    lib.lt_scene_set_mesh_dev(h, *margs)
    if probe: lib.lt_scene_set_probe(h, *pr[i])
    lib.lt_scene_render_dev(h, rss[s]._h, org, o["endpoints"], o["endcolors"], o["range"], o["endrem"], o["tri"], 1, sh[s], None)
for probe in (True, False):
    for i in range(100): step(i, False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N): step(i, probe)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"streams {S} tris {f.shape[0]} probes {probe}: enqueue {1e6*(t1-t0)/N:.2f} us/scan, complete {1e6*(t2-t0)/N:.2f} us/scan")
