#!/usr/bin/env python3
"""Debug: per-workgroup wall-clock stamps of k_tsdf_integrate_written (the further observations of a fused scan) on the fusion
chain's default volume.  Needs a library built with LIDARHIP_EXTRA_FLAGS=-DLT_TSDF_STAMP (exported for this process too).
    --pix: the phases of k_tsdf_integrate_pix per workgroup instead (library built with -DLT_PIX_STAMP)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lidar_transfer_amd import _lib
from lidar_transfer_amd.fusion import TSDFVolume
from lidar_transfer_amd.laserscan import create_rays
from lidar_transfer_amd.raytracer import RaySet, Scene
from lidar_transfer_amd.synth import WORKLOADS, synth_scene
wl = WORKLOADS["C2"]; H, W = wl["H"], wl["W"]; dev = torch.device("cuda", 0)
lib = _lib.load()
mesh0 = [torch.from_numpy(x).to(dev) for x in synth_scene(0, wl["tris"])]
rays = torch.from_numpy(create_rays(wl["fov_up"], wl["fov_down"], H, W)).to(dev)
sc = Scene(0); rs = RaySet(rays, H); sc.set_mesh(*mesh0)
o = sc.render(rs, (0, 0, 0)); torch.cuda.synchronize()
folded = (o["endcolors"][:, 2].reshape(H, W).float() * 65536.0).contiguous()
depth = o["range"].reshape(H, W).clone(); remi = o["endrem"].reshape(H, W).clone()
gen = torch.Generator(device=dev); gen.manual_seed(1234)
noise = (torch.rand((H, W), device=dev, generator=gen) - 0.5) * 0.04
depth2 = torch.where(depth == 0, depth, depth + noise).contiguous()
vol = TSDFVolume(np.array([[-50.0, 50.0], [-50.0, 50.0], [-5.0, 5.0]]), 0.05, wl["fov_up"], wl["fov_down"])
sp = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for i in range(3):
    assert lib.lt_tsdf_reset(vol._h, sp) == 0
    for d in (depth, depth2):
        assert lib.lt_tsdf_integrate_dev(vol._h, folded.data_ptr(), d.data_ptr(), remi.data_ptr(), H, W, 1.0, 1, sp) == 0
torch.cuda.synchronize()
if "--pix" in sys.argv:  # k_tsdf_integrate_pix of the FIRST observation (library built with -DLT_PIX_STAMP)
    assert lib.lt_tsdf_reset(vol._h, sp) == 0
    assert lib.lt_tsdf_integrate_dev(vol._h, folded.data_ptr(), depth.data_ptr(), remi.data_ptr(), H, W, 1.0, 1, sp) == 0
    torch.cuda.synchronize()
    n = 1 << 12
    buf = np.zeros(5 * n, np.uint64)
    lib.lt_debug_pix_stamps.argtypes = [C.c_void_p, C.c_int]
    assert lib.lt_debug_pix_stamps(buf.ctypes.data_as(C.c_void_p), n) == 0
    t = buf.reshape(n, 5).astype(np.int64); t = t[(t > 0).all(axis=1)]  # (a workgroup without pairs sets no mark 2)
    t0 = t[:, 0].min()
    u = (t - t0) / 100.0
    print("k_tsdf_integrate_pix: %d workgroups, span %.1f us" % (len(t), u[:, 4].max()))
    for name, a in (("start", u[:, 0]), ("phase A", u[:, 1] - u[:, 0]), ("first chunk's pairs + scan", u[:, 2] - u[:, 1]),
                    ("voxel rounds (+ further chunks)", u[:, 3] - u[:, 2]), ("flush", u[:, 4] - u[:, 3]), ("life", u[:, 4] - u[:, 0]), ("end", u[:, 4])):
        print("  %-32s mean %7.2f p50 %7.2f p90 %7.2f max %7.2f us" % (name, a.mean(), np.percentile(a, 50), np.percentile(a, 90), a.max()))
    sys.exit(0)
n = 1 << 12
buf = np.zeros(5 * n, np.uint64)
lib.lt_debug_tsdf_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.lt_debug_tsdf_stamps(buf.ctypes.data_as(C.c_void_p), n) == 0
t = buf.reshape(n, 5).astype(np.int64); t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
start = (t[:, 0] - t0) / 100.0; end = (t[:, 1] - t0) / 100.0; pro = t[:, 2] / 100.0; ev = t[:, 3] / 100.0
nch = t[:, 4] & 0xFFFF; vox = t[:, 4] >> 16
print("workgroups %d, span %.1f us; written chunks per workgroup: mean %.1f max %d; voxels per chunk: mean %.0f" % (
    len(t), end.max(), nch.mean(), nch.max(), vox.sum() / max(1, nch.sum())))
for name, a in (("start", start), ("life", end - start), ("prologues", pro), ("voxel rounds", ev)):
    print("  %-13s mean %7.2f p50 %7.2f p90 %7.2f max %7.2f us" % (name, a.mean(), np.percentile(a, 50), np.percentile(a, 90), a.max()))
print("  per chunk: prologue %.2f us, voxel rounds %.2f us" % (pro.sum() / nch.sum(), ev.sum() / nch.sum()))
