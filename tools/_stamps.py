import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lidar_transfer_amd import _lib
from lidar_transfer_amd.laserscan import create_rays
from lidar_transfer_amd.raytracer import RaySet, Scene
from lidar_transfer_amd.synth import WORKLOADS, synth_scene
wl = WORKLOADS["C2"]; dev = torch.device("cuda", 0)
rays = torch.from_numpy(create_rays(wl["fov_up"], wl["fov_down"], wl["H"], wl["W"])).to(dev)
meshes = [[torch.from_numpy(x).to(dev) for x in synth_scene(i, wl["tris"])] for i in range(6)]
sc = Scene(0); rs = RaySet(rays, wl["H"]); out = sc.alloc_outputs(wl["H"] * wl["W"])
for i in range(12):
    sc.set_mesh(*meshes[i % 6]); o = sc.render(rs, (0.0, 0.0, 0.0), out=out, stats=True)
nw = 2048
buf = np.zeros(16 * nw, np.uint64)
lib = _lib.load()
lib.lt_debug_wave_times.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
assert lib.lt_debug_wave_times(sc._h, buf.ctypes.data_as(C.c_void_p), 8 * nw) == 0
t = buf.reshape(nw, 16).astype(np.int64)
print("ms_trace", o["stats"]["ms_trace"])
n = int((t[0] > 0).sum())
d = np.diff(t[:, :n], axis=1)
names = ["start->V ready", "bins", "prefix+cap+barrier", "phase B"]
for k in range(n - 1):
    nm = names[k % 4] if k > 0 else "issue->V0 ready"
    nm = ["V ready (wait)", "bins", "prefix+barrier", "phase B"][k % 4]
    print("blk %d %-16s mean %7.0f p50 %7.0f p90 %7.0f" % (k // 4, nm, d[:, k].mean(), np.percentile(d[:, k], 50), np.percentile(d[:, k], 90)))
print("lifetime mean %.0f" % (t[:, n - 1] - t[:, 0]).mean())
