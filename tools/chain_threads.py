#!/usr/bin/env python3
"""Fusion chain throughput with several volumes in flight (one host thread, stream, volume, mesh and scene each):
does overlapping the latency-bound phases of different scans raise the scan rate?  (one JSON line)"""
import ctypes as C, json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lidar_transfer_amd import _lib
from lidar_transfer_amd.fusion import DeviceMesh, TSDFVolume
from lidar_transfer_amd.laserscan import create_rays
from lidar_transfer_amd.raytracer import RaySet, Scene
from lidar_transfer_amd.synth import WORKLOADS, synth_scene
nthreads = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
wl = WORKLOADS["C2"]; H, W = wl["H"], wl["W"]; dev = torch.device("cuda", 0)
lib = _lib.load()
mesh0 = [torch.from_numpy(x).to(dev) for x in synth_scene(0, wl["tris"])]
rays = torch.from_numpy(create_rays(wl["fov_up"], wl["fov_down"], H, W)).to(dev)
sc0 = Scene(0); rs = RaySet(rays, H); sc0.set_mesh(*mesh0)
o = sc0.render(rs, (0, 0, 0)); torch.cuda.synchronize()
folded = (o["endcolors"][:, 2].reshape(H, W).float() * 65536.0).contiguous()
depth = o["range"].reshape(H, W).contiguous(); remi = o["endrem"].reshape(H, W).contiguous()
org = (C.c_float * 3)(0, 0, 0)
workers = []
for k in range(nthreads):
    st = torch.cuda.Stream(dev)
    workers.append(dict(vol=TSDFVolume(np.array([[-50.0, 50.0], [-50.0, 50.0], [-5.0, 5.0]]), 0.05, wl["fov_up"], wl["fov_down"]),
                        mesh=DeviceMesh(0), sc=Scene(0), st=st, sp=C.c_void_p(st.cuda_stream), out=sc0.alloc_outputs(H * W)))
torch.cuda.synchronize()

def work(wk, reps):
    torch.cuda.set_device(0)
    vol, mesh, sc, sp, out = wk["vol"], wk["mesh"], wk["sc"], wk["sp"], wk["out"]
    for i in range(reps):
        assert lib.lt_tsdf_reset(vol._h, sp) == 0
        assert lib.lt_tsdf_integrate_dev(vol._h, folded.data_ptr(), depth.data_ptr(), remi.data_ptr(), H, W, 1.0, 1, sp) == 0
        assert lib.lt_tsdf_extract_mesh_dev(vol._h, mesh._h, sp, None) == 0
        assert lib.lt_scene_set_mesh(sc._h, mesh._h) == 0
        assert lib.lt_scene_render_dev(sc._h, rs._h, org, out["endpoints"].data_ptr(), out["endcolors"].data_ptr(), out["range"].data_ptr(),
                                       out["endrem"].data_ptr(), out["tri"].data_ptr(), 1, sp, None) == 0
    wk["st"].synchronize()

for wk in workers: work(wk, 2)
t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(wk, n)) for wk in workers]
for t in th: t.start()
for t in th: t.join()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"threads": nthreads, "scans": nthreads * n, "ms_per_scan": round(dt / (nthreads * n) * 1e3, 4),
                  "scans_per_s": round(nthreads * n / dt, 1), "hits": int((workers[0]["out"]["range"] > 0).sum())}))
