#!/usr/bin/env python3
"""tests/stress_tsdf_ref.py [n] [seed] -- the reference's own TSDF `integrate` kernel (its source compiled for gfx950,
oracle/_ref/libref_tsdf_integrate{,_plain}.so) against TSDFVolume.integrate and integrate_multi on RANDOM configurations:
field of view, voxel size, volume extent and offset (the sensor on / off a lattice point, inside / outside the volume), image
shape, number and content of the observations (zero / -1 / NaN / infinite depth pixels, 1-4 classes), both branches.  All
four volumes must be bit-identical every time."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(argv=None):
    argv = sys.argv if argv is None else argv
    import torch
    from lidar_transfer_amd.fusion import TSDFVolume
    from oracle import binding as ob
    n = int(argv[1]) if len(argv) > 1 else 40
    rng = np.random.default_rng(int(argv[2]) if len(argv) > 2 else 0)
    dev = torch.device("cuda", 0)
    vp = C.c_void_p
    bad = 0
    ran = 0
    touched_total = 0
    t0 = time.time()
    for k in range(n):
        merge = bool(k % 2 == 0)
        ref = ob.ref_tsdf_lib(merge)
        fu = float(rng.choice([2.0, 3.0, 10.0, 15.0, 30.0, 45.0])); fd = -float(rng.choice([10.0, 16.6, 25.0, 30.0, 50.0]))
        voxel = float(rng.choice([0.05, 0.1, 0.2, 0.25, 0.4]))
        ext = rng.uniform(4, 8, 2) * (voxel / 0.05) ** 0.7; zext = rng.uniform(2, 5)
        off = rng.uniform(-4, 4, 3) * (rng.random() < 0.5)          # half of the cases: the sensor off-centre
        if rng.random() < 0.3:
            off = np.round(off / voxel) * voxel                     # ... on a lattice point
        bnds = np.array([[-ext[0] + off[0], ext[0] + off[0]], [-ext[1] + off[1], ext[1] + off[1]], [-zext + off[2], zext + off[2]]])
        H = int(rng.choice([16, 25, 32, 64, 70])); W = int(rng.choice([97, 128, 256, 301, 512, 1024]))
        n_obs = int(rng.integers(1, 5))
        vol = TSDFVolume(bnds.copy(), voxel, fu, fd, merge=merge)
        fused = TSDFVolume(bnds.copy(), voxel, fu, fd, merge=merge)
        dims_t = tuple(int(x) for x in vol._vol_dim)
        if np.prod(dims_t) > 120e6:
            vol.close(); fused.close(); continue
        ran += 1
        dims = (C.c_int * 3)(*dims_t); org = (C.c_float * 3)(*[float(x) for x in vol._vol_origin])
        # (64 spare floats behind each field: the reference kernel's bounds test is `voxel_idx > N`, so thread N -- one past
        # the end -- runs and, when its point happens to project onto a valid pixel, WRITES element N of all four arrays:
        # into whatever lies behind them.  With back-to-back allocations that is voxel 0 of the next field; here it is the pad.)
        n_vox = int(np.prod(dims_t))
        flat = [torch.ones(n_vox + 64, device=dev)] + [torch.zeros(n_vox + 64, device=dev) for _ in range(3)]
        vr = [t[:n_vox].view(dims_t) for t in flat]
        st = vp(torch.cuda.current_stream().cuda_stream)
        yaw = np.linspace(-np.pi, np.pi, W)
        classes = rng.choice(np.array([0.0, 10.0, 40.0, 50.0, 259.0]), int(rng.integers(1, 5)), replace=False)
        obs = []
        for j in range(n_obs):
            depth = (rng.uniform(3, 12) + rng.uniform(0, 4) * np.sin(rng.integers(1, 5) * yaw + j)[None, :] + 0.3 * rng.random((H, W))).astype(np.float32)
            depth[rng.random((H, W)) < 0.05] = 0.0
            if rng.random() < 0.5:
                depth[:, W // 5:W // 4] = -1.0
            depth[rng.random((H, W)) < 0.002] = np.nan
            depth[rng.random((H, W)) < 0.002] = np.inf
            lab = rng.choice(classes, (H, W)).astype(np.float32)
            obs.append((np.stack([lab, np.zeros_like(lab), np.zeros_like(lab)], 2), depth, rng.random((H, W)).astype(np.float32)))
        for label3, depth, rem in obs:
            c = torch.from_numpy(label3).to(dev)
            folded = torch.floor(c[:, :, 0] * 65536 + c[:, :, 1] * 256 + c[:, :, 2]).contiguous()
            d, r = torch.from_numpy(depth).to(dev), torch.from_numpy(rem).to(dev)
            assert ref.ref_tsdf_integrate(*[vp(t.data_ptr()) for t in vr], dims, org, C.c_float(np.float32(voxel)), C.c_float(np.float32(voxel * 5)),
                                          C.c_float(fu), C.c_float(fd), vp(folded.data_ptr()), vp(d.data_ptr()), vp(r.data_ptr()), H, W, C.c_float(1.0), st) == 0
            vol.integrate(label3, depth, rem, np.eye(4), obs_weight=1.)
        fused.integrate_multi(obs, obs_weight=1.)
        torch.cuda.synchronize()
        touched_total += int(((vr[0] != 1) | (vr[1] != 0)).sum())
        for who, V in (("integrate", vol.get_volume_tensors()), ("integrate_multi", fused.get_volume_tensors())):
            nb = sum(int((a.view(torch.int32) != b.view(torch.int32)).sum()) for a, b in zip(vr, V))
            if nb:
                bad += 1
                if os.environ.get("LT_STRESS_VERBOSE"):
                    names = ("tsdf", "weight", "color", "rem")
                    for fi, (a, b) in enumerate(zip(vr, V)):
                        idx = torch.nonzero(a.view(torch.int32) != b.view(torch.int32))
                        for x, y, z in idx[:4].tolist():
                            px, py, pz = [float(np.float32(org[j]) + np.float32(v) * np.float32(voxel)) for j, v in enumerate((x, y, z))]
                            print(f"    {names[fi]}[{x},{y},{z}] pt ({px:.7g}, {py:.7g}, {pz:.7g}): reference {float(a[x, y, z])!r} product {float(b[x, y, z])!r}; "
                                  f"all fields ref {[float(t[x, y, z]) for t in vr]} prod {[float(t[x, y, z]) for t in V]}")
                print(f"case {k}: {who} differs in {nb} field values (merge {merge}, fov {fu}/{fd}, voxel {voxel}, dims {dims_t}, image {H}x{W}, {n_obs} observations, origin {list(org)})")
        vol.close(); fused.close()
    print(f"{ran} configurations (of {n} drawn; the rest beyond 120 M voxels), {touched_total} voxels touched by the reference kernel: {bad} mismatches ({time.time() - t0:.0f} s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
