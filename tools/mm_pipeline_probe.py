#!/usr/bin/env python3
"""deform('mergemesh') with scans in flight: ms per output scan for 1 .. 6 chains, and where a chain thread's time goes
(inside lt_mergemesh_scan_dev / around it).  python tools/mm_pipeline_probe.py [scans]"""
import gc, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lidar_transfer_amd import _lib
from lidar_transfer_amd.laserscan import create_rays
from lidar_transfer_amd.pipeline import FusionScanPipeline
from lidar_transfer_amd.raytracer import RaySet, Scene
from lidar_transfer_amd.synth import WORKLOADS, synth_scene

n_scans = int(sys.argv[1]) if len(sys.argv) > 1 else 48
wl = WORKLOADS["C2"]; H, W = wl["H"], wl["W"]; dev = torch.device("cuda", 0)
mesh0 = [torch.from_numpy(x).to(dev) for x in synth_scene(0, wl["tris"])]
rays = torch.from_numpy(create_rays(wl["fov_up"], wl["fov_down"], H, W)).to(dev)
rs = RaySet(rays, H); sc = Scene(0); sc.set_mesh(*mesh0)
o = sc.render(rs, (0, 0, 0)); torch.cuda.synchronize()
hit = o["tri"] >= 0
cloud = [(o["endpoints"][hit].double().contiguous(), o["endrem"][hit].contiguous(), o["endcolors"][hit][:, 2].contiguous().to(torch.int32))]
sc.close(); rs.close()
lib = _lib.load()
inside = [0.0, 0]
orig = lib.lt_mergemesh_scan_dev
class Timed:
    def __call__(self, *a):
        t = time.perf_counter(); r = orig(*a); inside[0] += time.perf_counter() - t; inside[1] += 1; return r
for chains in (1, 2, 3, 4, 6):
    bnds = np.array([-50, 50, -50, 50, -5, 5]).reshape(3, 2)
    with FusionScanPipeline(bnds, 0.05, wl["fov_up"], wl["fov_down"], rays, H, chains=chains, device=0, label_image=True,
                            source_hw=(H, W), fixed_volume=False) as pipe:
        for t in [pipe.submit_mergemesh(cloud, inputs_ready=True) for _ in range(4 * chains)]:
            pipe.wait(t)
        bufs = [pipe._chains[0]["scene"].alloc_outputs(H * W, label_image=True) for _ in range(n_scans)]
        torch.cuda.synchronize(); gc.collect(); gc.disable()
        bursts = []  # (three bursts, the median: the process's one-time runtime stall of tens of ms lands in ONE of them)
        for _ in range(3):
            inside[0], inside[1] = 0.0, 0
            lib.lt_mergemesh_scan_dev = Timed()
            t0 = time.perf_counter()
            tk = [pipe.submit_mergemesh(cloud, out=b, inputs_ready=True) for b in bufs]
            t_sub = time.perf_counter() - t0
            for t in tk:
                pipe.wait(t)
            bursts.append((time.perf_counter() - t0, t_sub, inside[0], inside[1]))
            lib.lt_mergemesh_scan_dev = orig
        gc.enable()
        dt, t_sub, inside[0], inside[1] = sorted(bursts)[1]
        print(json.dumps({"chains": chains, "ms_per_scan": round(dt / n_scans * 1e3, 4),
                          "bursts_ms_per_scan": [round(b[0] / n_scans * 1e3, 4) for b in bursts], "submit_ms_total": round(t_sub * 1e3, 3),
                          "native_ms_per_call": round(inside[0] / max(inside[1], 1) * 1e3, 4), "native_calls": inside[1],
                          "stats": pipe._mm_state.stats}))
