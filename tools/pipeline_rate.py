#!/usr/bin/env python3
"""Throughput of lidar_transfer_amd.pipeline.ScanPipeline (the library form of bench.py's step loop) on workload C2:
12 scenes cycled, range + label image of every scan kept."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidar_transfer_amd.laserscan import create_rays
from lidar_transfer_amd.pipeline import ScanPipeline
from lidar_transfer_amd.synth import WORKLOADS, synth_scene
wl = WORKLOADS["C2"]; dev = torch.device("cuda", 0); R = wl["H"] * wl["W"]
rays = torch.from_numpy(create_rays(wl["fov_up"], wl["fov_down"], wl["H"], wl["W"])).to(dev)
meshes = [[torch.from_numpy(x).to(dev) for x in synth_scene(i, wl["tris"])] for i in range(12)]
N = 4000
ranges = torch.empty((N, R), dtype=torch.float32, device=dev); labels = torch.empty((N, R), dtype=torch.int32, device=dev)
with ScanPipeline(rays, wl["H"]) as pipe:
    for k in range(200):
        pipe.submit(*meshes[k % 12], (0.0, 0.0, 0.0), range_out=ranges[k], label_out=labels[k])
    pipe.flush(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(N):
        pipe.submit(*meshes[k % 12], (0.0, 0.0, 0.0), range_out=ranges[k], label_out=labels[k])
    pipe.flush()
    dt = time.perf_counter() - t0
print("ScanPipeline: %.0f scans/s = %.2f Grays/s (%d scans, hits in the last one: %d)" % (N / dt, N * R / dt / 1e9, N, int((ranges[-1] > 0).sum())))
