import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lidar_transfer_amd import _lib
from lidar_transfer_amd.laserscan import create_rays
from lidar_transfer_amd.raytracer import RaySet, Scene
from lidar_transfer_amd.synth import WORKLOADS, synth_scene
wl = WORKLOADS["C2"]; dev = torch.device("cuda", 0); R = wl["H"] * wl["W"]
rays = torch.from_numpy(create_rays(wl["fov_up"], wl["fov_down"], wl["H"], wl["W"])).to(dev)
meshes = [[torch.from_numpy(x).to(dev) for x in synth_scene(i, wl["tris"])] for i in range(12)]
S = 16
workers = [Scene(0) for _ in range(S)]; raysets = [RaySet(rays, wl["H"]) for _ in range(S)]
outs = [workers[0].alloc_outputs(R, label_image=True) for _ in range(S)]
streams = [torch.cuda.Stream(dev) for _ in range(2)]
def batch(b):
    g = b % 2
    ws = workers[g * 8:(g + 1) * 8]
    for k, w in enumerate(ws):
        w.set_mesh(*meshes[(b * 8 + k) % 12])
    Scene.render_batch(ws, raysets[g * 8:(g + 1) * 8], [(0.0, 0.0, 0.0)] * 8, outs=outs[g * 8:(g + 1) * 8], stream=streams[g], label_image=True)
for b in range(40): batch(b)
torch.cuda.synchronize(); t0 = time.perf_counter()
NB = 300
for b in range(NB): batch(b)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("Grays/s %.3f  us/scan %.2f" % (NB * 8 * R / dt / 1e9, dt / NB / 8 * 1e6))
nw = 2048
buf = np.zeros(16 * nw, np.uint64)
lib = _lib.load()
lib.lt_debug_wave_times.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
assert lib.lt_debug_wave_times(workers[0]._h, buf.ctypes.data_as(C.c_void_p), 8 * nw) == 0
t = buf.reshape(nw, 16).astype(np.int64)
n = int((t[0] > 0).sum())
if n > 1:
    d = np.diff(t[:, :n], axis=1)
    for k in range(n - 1):
        nm = ["V ready (wait)", "bins", "prefix+barrier", "phase B"][k % 4]
        print("blk %d %-16s mean %7.0f p50 %7.0f p90 %7.0f" % (k // 4, nm, d[:, k].mean(), np.percentile(d[:, k], 50), np.percentile(d[:, k], 90)))
    print("lifetime mean %.0f; start spread of the 2048 waves: %d cycles" % ((t[:, n - 1] - t[:, 0]).mean(), t[:, 0].max() - t[:, 0].min()))
