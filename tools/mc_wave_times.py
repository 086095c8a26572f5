#!/usr/bin/env python3
"""Debug: per-wave wall-clock stamps of k_mc_words / k_mc_compact on the fusion chain's default volume.
Needs a library built with the stamps:  LIDARHIP_EXTRA_FLAGS=-DLT_MC_STAMP=1 (k_mc_words), =2 (k_mc_compact) or =3
(k_mc_emit_batch: a batch's life by section) -- exported for this process too."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lidar_transfer_amd import _lib
from lidar_transfer_amd.fusion import DeviceMesh, TSDFVolume
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
vol = TSDFVolume(np.array([[-50.0, 50.0], [-50.0, 50.0], [-5.0, 5.0]]), 0.05, wl["fov_up"], wl["fov_down"])
mesh = DeviceMesh(0)
sp = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for i in range(4):
    assert lib.lt_tsdf_reset(vol._h, sp) == 0
    assert lib.lt_tsdf_integrate_dev(vol._h, folded.data_ptr(), depth.data_ptr(), remi.data_ptr(), H, W, 1.0, 1, sp) == 0
    assert lib.lt_tsdf_extract_mesh_dev(vol._h, mesh._h, sp, None) == 0
torch.cuda.synchronize()
nw = 1 << 16
buf = np.zeros(4 * nw, np.uint64)
lib.lt_debug_mc_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.lt_debug_mc_stamps(buf.ctypes.data_as(C.c_void_p), nw) == 0
if "-DLT_MC_STAMP=3" in os.environ.get("LIDARHIP_EXTRA_FLAGS", ""):
    # k_mc_emit_batch by section: wall clock (100 MHz) accumulated over all batches of all launches
    t = buf[:8 * (1 << 15)].reshape(-1, 8).astype(np.float64) / 100.0
    t = t[t[:, 0] > 0]
    names = ["records -> LDS", "sign words, compact indices, neighbour records", "-", "vertex list",
             "vertex pass (field samples, attributes, stores drained)", "cell masks, cell list, triangle offsets",
             "triangle list + pass (stores drained)"]
    print("k_mc_emit_batch: %d batches of the last launch; a batch's life by section (us): mean / p90" % len(t))
    for i, nm in enumerate(names):
        if nm != "-":
            print("  %-58s %6.2f %6.2f" % (nm, t[:, i].mean(), np.percentile(t[:, i], 90)))
    life = t[:, [0, 1, 3, 4, 5, 6]].sum(axis=1)
    print("  %-58s %6.2f %6.2f" % ("a batch, start to end", life.mean(), np.percentile(life, 90)))
    print("  a batch's life: p99 %.1f, p99.9 %.1f, max %.1f us; the %d batches above 30 us hold %.1f %% of the wave time" % (
        np.percentile(life, 99), np.percentile(life, 99.9), life.max(), int((life > 30).sum()), 100.0 * life[life > 30].sum() / life.sum()))
    sys.exit(0)
t = buf.reshape(nw, 4).astype(np.int64)
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
start = (t[:, 0] - t0) / 100.0; flags = (t[:, 1] - t[:, 0]) / 100.0; dur = (t[:, 2] - t[:, 0]) / 100.0; end = (t[:, 2] - t0) / 100.0
live = t[:, 3]
print("waves %d, span %.1f us, blocks walked: mean %.2f max %d" % (len(t), end.max(), live.mean(), live.max()))
for name, a in (("start", start), ("stamp ballot", flags), ("duration", dur), ("end", end)):
    print("  %-14s mean %7.2f p50 %7.2f p90 %7.2f p99 %7.2f max %7.2f" % (name, a.mean(), np.percentile(a, 50), np.percentile(a, 90), np.percentile(a, 99), a.max()))
for k in range(0, int(live.max()) + 1):
    sel = live == k
    if sel.any():
        print("  waves with %d blocks: %6d, duration mean %6.2f p90 %6.2f max %6.2f us" % (k, sel.sum(), dur[sel].mean(), np.percentile(dur[sel], 90), dur[sel].max()))
print("  busy fraction of the wave slots over the span (8192 slots): %.3f" % (dur.sum() / (8192 * end.max())))
