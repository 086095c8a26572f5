import numpy as np, torch, sys, os
sys.path.insert(0, os.getcwd())
from lidar_transfer_amd.deform import DeviceDeform
g = np.load("tests/golden/f13_deform_mesh.npz")
r = np.load("gpurun_in/ref_vol_a.npz")
case = "a"
src = (int(g["a_source"][0]), int(g["a_source"][1]), float(g["a_source"][2]), float(g["a_source"][3]))
tgt = (int(g["a_target"][0]), int(g["a_target"][1]), float(g["a_target"][2]), float(g["a_target"][3]))
clouds = [(torch.from_numpy(g["a_points0"]).cuda(), torch.from_numpy(g["a_rem0"]).cuda(), torch.from_numpy(g["a_label0"].astype(np.int32)).cuda())]
dd = DeviceDeform(src, tgt, g["a_bnds"].copy(), float(g["a_voxel"]), fusion="numpy")
got = dd.mesh(clouds)
torch.cuda.synchronize()
tsdf, weight, color, rem = [t.cpu().numpy() for t in dd.vol.get_volume_tensors()]
for name, a, b in (("weight", weight, r["weight"]), ("tsdf", tsdf, r["tsdf"]), ("color", color, r["color"])):
    d = np.argwhere(a.view(np.int32) != b.view(np.int32))
    print(name, "differ", len(d))
    for ix, iy, iz in d[:10]:
        x, y, z = -8.0 + ix * 0.1, -8.0 + iy * 0.1, -3.0 + iz * 0.1
        depth = np.sqrt(x * x + y * y + z * z)
        pitch = np.arcsin(z / depth); yaw = -np.arctan2(y, x)
        print("  voxel", ix, iy, iz, "xyz", repr(x), repr(y), repr(z), "ours", a[ix, iy, iz], "ref", b[ix, iy, iz], "depth", repr(depth),
              "pitch", repr(pitch), "deg", np.degrees(pitch), "yaw", repr(yaw), "proj_x", repr(0.5 * (yaw / np.pi + 1.0) * src[1]),
              "proj_y", repr((1.0 - (pitch + abs(np.radians(src[3]))) / (abs(np.radians(src[3])) + abs(np.radians(src[2])))) * src[0]))
# device f64 atan2 / asin vs numpy on lattice points
xs = torch.arange(-80, 81, dtype=torch.float64, device="cuda") * 0.1
X, Y = torch.meshgrid(xs, xs, indexing="ij")
a_dev = torch.atan2(Y, X).cpu().numpy(); a_np = np.arctan2(Y.cpu().numpy(), X.cpu().numpy())
print("torch.atan2 vs numpy on the lattice: differ", int((a_dev != a_np).sum()), "of", a_np.size)
