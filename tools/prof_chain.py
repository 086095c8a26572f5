#!/usr/bin/env python3
"""Driver for rocprofv3: the fusion chain (reset -> integrate -> marching cubes -> render) on the default volume, N times.
    python tools/prof_chain.py [N [observations per output scan]]      (observations 2.. are noisy copies, as in bench.py)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lidar_transfer_amd import _lib
from lidar_transfer_amd.fusion import DeviceMesh, TSDFVolume
from lidar_transfer_amd.laserscan import create_rays
from lidar_transfer_amd.raytracer import RaySet, Scene
from lidar_transfer_amd.synth import WORKLOADS, synth_scene
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 10
n_obs = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 1
wl = WORKLOADS["C2"]; H, W = wl["H"], wl["W"]; dev = torch.device("cuda", 0)
lib = _lib.load()
mesh0 = [torch.from_numpy(x).to(dev) for x in synth_scene(0, wl["tris"])]
rays = torch.from_numpy(create_rays(wl["fov_up"], wl["fov_down"], H, W)).to(dev)
sc = Scene(0); rs = RaySet(rays, H); sc.set_mesh(*mesh0)
o = sc.render(rs, (0, 0, 0)); torch.cuda.synchronize()
folded = (o["endcolors"][:, 2].reshape(H, W).float() * 65536.0).contiguous()
depth = o["range"].reshape(H, W).clone(); remi = o["endrem"].reshape(H, W).clone()   # (clones: `o` is rendered into again below)
vol = TSDFVolume(np.array([[-50.0, 50.0], [-50.0, 50.0], [-5.0, 5.0]]), 0.05, wl["fov_up"], wl["fov_down"])
mesh = DeviceMesh(0)
sp = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream); org = (C.c_float * 3)(0, 0, 0)
gen = torch.Generator(device=dev); gen.manual_seed(1234)
obs = [(folded, depth)]
for k in range(1, n_obs):
    noise = (torch.rand((H, W), device=dev, generator=gen) - 0.5) * 0.04
    hole = torch.rand((H, W), device=dev, generator=gen) < 0.05
    flip = torch.rand((H, W), device=dev, generator=gen) < 0.02
    obs.append((torch.where(flip, torch.full_like(folded, 50.0 * 65536.0), folded).contiguous(),
                torch.where(hole | (depth == 0), torch.zeros_like(depth), depth + noise).contiguous()))
evs = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(n)]   # phase marks (--phases prints their medians)
for i in range(n):
    evs[i][0].record()
    assert lib.lt_tsdf_reset(vol._h, sp) == 0
    evs[i][1].record()
    if os.environ.get("LT_CHAIN_SEQUENTIAL") == "1":   # one lt_tsdf_integrate_dev per observation (rounds 1-3)
        for f_k, d_k in obs:
            assert lib.lt_tsdf_integrate_dev(vol._h, f_k.data_ptr(), d_k.data_ptr(), remi.data_ptr(), H, W, 1.0, 1, sp) == 0
    else:                                               # all observations of the fresh volume in one fused pass
        vpn = C.c_void_p * len(obs)
        assert lib.lt_tsdf_integrate_multi_dev(vol._h, len(obs), vpn(*[f_k.data_ptr() for f_k, _ in obs]),
                                               vpn(*[d_k.data_ptr() for _, d_k in obs]), vpn(*[remi.data_ptr()] * len(obs)),
                                               H, W, 1.0, 1, sp) == 0
    evs[i][2].record()
    assert lib.lt_tsdf_extract_mesh_dev(vol._h, mesh._h, sp, None) == 0
    evs[i][3].record()
    assert lib.lt_scene_set_mesh(sc._h, mesh._h) == 0
    assert lib.lt_scene_render_dev(sc._h, rs._h, org, o["endpoints"].data_ptr(), o["endcolors"].data_ptr(), o["range"].data_ptr(),
                                   o["endrem"].data_ptr(), o["tri"].data_ptr(), 1, sp, None) == 0
    evs[i][4].record()
torch.cuda.synchronize()
dirty = None
print("verts", mesh.n_verts, "faces", mesh.n_faces)
if "--phases" in sys.argv:
    import hashlib, json
    ms = np.array([[e[k].elapsed_time(e[k + 1]) for k in range(4)] for e in evs[2:]])
    v, f, c, r = mesh.tensors()
    sha = hashlib.sha256(v.cpu().numpy().tobytes() + f.cpu().numpy().tobytes() + c.cpu().numpy().tobytes() + r.cpu().numpy().tobytes()).hexdigest()[:16]
    print(json.dumps({"phase_ms_median": dict(zip(("reset", "integrate", "marching_cubes", "render"), [round(float(x), 4) for x in np.median(ms, axis=0)])),
                      "mc_min": round(float(ms[:, 2].min()), 4), "mesh_sha": sha, "range_sha": hashlib.sha256(o["range"].cpu().numpy().tobytes()).hexdigest()[:16],
                      "env": {k: v_ for k, v_ in os.environ.items() if k.startswith("LIDARHIP_MC")}}))
if os.environ.get("LIDARHIP_DEBUG_TSDF"):
    c3 = (C.c_ulonglong * 8)()
    if lib.lt_debug_tsdf_pix_counts(c3) == 0:
        print("k_tsdf_integrate_pix: pairs %d, candidate voxels %d, written %d" % (c3[0], c3[1], c3[2]))
        nwg = max(1, (H * W + 63) // 64)
        print("  per workgroup (100 MHz wall clock, mean us): phase A %.2f, A + first chunk's pairs + scan %.2f, voxel rounds %.2f (%d chunks)"
              % (c3[4] / nwg / 100.0, c3[5] / nwg / 100.0, c3[6] / max(1, c3[7]) / 100.0, c3[7]))
if "--ranges" in sys.argv:
    # how tight the per-column written ranges (min .. max z, what k_tsdf_integrate_written / reset walk) are around what was written
    tv, wv, _, _ = vol.get_volume_tensors()
    n_written = n_cols = n_range = 0
    zs = torch.arange(tv.shape[2], device=dev)
    for x0 in range(0, tv.shape[0], 100):
        m = (tv[x0:x0 + 100] != 1) | (wv[x0:x0 + 100] != 0)
        anyc = m.any(dim=2)
        lo = torch.where(m, zs, tv.shape[2]).amin(dim=2)
        hi = torch.where(m, zs, -1).amax(dim=2)
        n_written += int(m.sum()); n_cols += int(anyc.sum()); n_range += int((hi - lo + 1)[anyc].sum())
    print("written voxels %d in %d columns; sum of the columns' min..max ranges %d (x %.2f)" % (n_written, n_cols, n_range, n_range / max(1, n_written)))
if "--count" in sys.argv:
    sc.set_device_mesh(mesh)
    st = sc.render(rs, (0, 0, 0), count=True)["stats"]
    print("MC mesh render: candidate bins %d (%.2f per triangle), MT tests %d (%.2f per ray), hits %d" % (
        st["nodes_visited"], st["nodes_visited"] / mesh.n_faces, st["tris_tested"], st["tris_tested"] / (H * W), st["n_hits"]))
    qs = (C.c_int * 2)()
    lib.lt_debug_scatter_queues.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    assert lib.lt_debug_scatter_queues(sc._h, qs) == 0
    print("queued big triangles %d, queued slices %d" % (qs[0], qs[1]))
    v, f, c, r = mesh.tensors()
    tri = v[f.long()]                      # [F, 3, 3]
    d = tri.norm(dim=2).min(dim=1).values
    rho = tri[:, :, :2].norm(dim=2).min(dim=1).values
    print("triangles: nearest vertex distance  min %.3f  p1 %.3f  p50 %.2f ; min horizontal distance  min %.4f  p0.1 %.4f" % (
        d.min(), d.kthvalue(max(1, int(0.01 * d.numel()))).values, d.median(), rho.min(), rho.kthvalue(max(1, int(0.001 * rho.numel()))).values))
    print("triangles within 1 m of the sensor: %d, within 0.3 m of the vertical axis: %d" % (int((d < 1.0).sum()), int((rho < 0.3).sum())))
    area = torch.linalg.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]).norm(dim=1) * 0.5
    print("degenerate (area < 1e-9 m^2): %d of %d" % (int((area < 1e-9).sum()), area.numel()))
    nl = qs[0]
    ids = np.zeros(nl, np.int32)
    lib.lt_debug_scatter_large.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    assert lib.lt_debug_scatter_large(sc._h, ids.ctypes.data_as(C.c_void_p), nl) == 0
    T = tri[torch.from_numpy(ids).long().to(dev)].cpu().numpy().astype(np.float64)
    np.set_printoptions(precision=5, suppress=True, linewidth=200)
    for k in range(min(6, nl)):
        a = T[k]
        az = np.degrees(np.arctan2(a[:, 1], a[:, 0])); el = np.degrees(np.arctan2(a[:, 2], np.hypot(a[:, 0], a[:, 1])))
        print("big face", ids[k], "verts", a.reshape(-1), "az", az, "el", el)
    az_all = np.degrees(np.arctan2(T[:, :, 1], T[:, :, 0]))
    print("big triangles: |azimuth| of their vertices min %.3f ; all near the +-180 seam: %s" % (np.abs(az_all).min(), bool((np.abs(az_all) > 179).all())))
