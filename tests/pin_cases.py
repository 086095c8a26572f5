"""Seeded inputs shared by the generators of golden fixtures that need the reference's own environment (tests/golden/
make_golden_mc.py: scikit-image, run with the image's /opt/conda interpreter; make_golden_tsdf_cuda.py: pycuda + an NVIDIA GPU;
make_golden_deform_fuzz.py) and by the GPU tests that consume those fixtures (tests/test_pin_f10_f11_gpu.py,
tests/test_deform_gpu.py).  numpy only, deterministic for a seed on any machine (integer lattices, float64 sin/cos rounded to
float32 afterwards)."""
from __future__ import annotations

import numpy as np

from lidar_transfer_amd.synth import synth_cloud

# ---- F10: marching cubes (fusion_lidar.py:403-424) ----------------------------------------------------------------------
#: name -> (H, W, fov_up, fov_down) of the image rendered from the extracted mesh (None: no render, mesh only)
MC_CASES = {"street": (64, 1024, 3.0, -25.0), "blob": (32, 256, 40.0, -40.0), "noise": None}


def mc_case(name):
    """(tsdf, color_vol, rem_vol, voxel_size, vol_origin) of a marching-cubes case; float32 volumes, C order."""
    if name == "street":
        # a truncated signed distance field of a street scene seen from the origin: ground height field, four boxes,
        # a pole -- what `integrate` leaves behind (values in [-1, 1], +1 in free space), 25 cm voxels, labels by object
        vs = np.float32(0.25)
        dims = (160, 160, 32)
        org = np.array([-20.0, -20.0, -4.0], np.float32)
        x, y, z = np.meshgrid(*[org[k] + vs * np.arange(dims[k]) for k in range(3)], indexing="ij")
        ground = z - (-1.73 + 0.15 * np.sin(0.3 * x) * np.cos(0.2 * y))          # > 0 above the ground
        sd, lab = ground.copy(), np.full(dims, 40.0)
        boxes = [(6.0, 3.0, 2.0, 3.0, 2.5, 50.0), (-7.5, -4.0, 3.0, 2.0, 3.5, 50.0), (2.0, -9.0, 1.0, 2.2, 1.6, 10.0),
                 (-3.0, 8.0, 4.0, 1.5, 2.0, 50.0)]
        for cx, cy, hx, hy, top, lb in boxes:
            q = np.stack([np.abs(x - cx) - hx, np.abs(y - cy) - hy, z - top], 0)
            d = np.linalg.norm(np.maximum(q, 0), axis=0) + np.minimum(q.max(0), 0)
            lab = np.where(d < sd, lb, lab)
            sd = np.minimum(sd, d)
        pole = np.maximum(np.hypot(x - 4.0, y + 3.0) - 0.2, z - 3.0)
        lab = np.where(pole < sd, 80.0, lab)
        sd = np.minimum(sd, pole)
        tsdf = np.clip(sd / (5 * float(vs)), -1.0, 1.0).astype(np.float32)
        # what the reference would not have seen stays at the initial 1 (behind surfaces by more than the margin)
        tsdf[sd < -5 * float(vs)] = 1.0
        color = (lab * 65536.0).astype(np.float32)                                # label in the b channel (:261-264)
        rem = (0.5 + 0.4 * np.sin(0.7 * x + 0.3 * y)).astype(np.float32)
        return tsdf, color, rem, vs, org
    if name == "blob":
        shape = (64, 64, 64)
        g = [np.linspace(-1, 1, n) for n in shape]
        x, y, z = np.meshgrid(*g, indexing="ij")
        t = (np.sqrt(x * x + 1.3 * y * y + 0.7 * z * z) - 0.6 + 0.05 * np.sin(9 * x) * np.cos(7 * y)).astype(np.float32)
        t = (-t).astype(np.float32)            # inside-out: free space (positive) around the origin, surface around it
        col = np.full(shape, 40 * 65536, np.float32)
        col[:, :, 32:] = 50 * 65536
        rem = (0.5 + 0.5 * np.sin(3 * z)).astype(np.float32)
        return t, col, rem, np.float32(0.1), np.array([-3.2, -3.2, -3.2], np.float32)
    if name == "noise":                          # all 256 cases, every ambiguous face and interior
        shape = (20, 18, 22)
        rng = np.random.default_rng(2024)
        t = rng.normal(size=shape).astype(np.float32)
        t[rng.random(shape) < 0.05] = 0.0
        col = (rng.integers(0, 260, shape) * 65536 + rng.integers(0, 256, shape) * 256 + rng.integers(0, 256, shape)).astype(np.float32)
        rem = rng.random(shape).astype(np.float32)
        return t, col, rem, np.float32(0.05), np.array([-1.25, 2.5, 0.75], np.float32)
    raise KeyError(name)


# ---- F11: class-aware TSDF integrate (the CUDA kernel, fusion_lidar.py:66-229) ------------------------------------------
TSDF_BOUNDS = np.array([[-16.0, 16.0], [-16.0, 16.0], [-4.0, 4.0]])
TSDF_VOXEL, TSDF_H, TSDF_W, TSDF_FOV = 0.25, 32, 256, (3.0, -25.0)


def tsdf_observations(n=3):
    """n observations (label3 [H,W,3] with the label in channel 0 as laserscan.py:893-895 builds it, depth, remission) of
    a scene with several classes side by side -- so that voxels see DIFFERENT classes in successive observations and the
    kernel's `dist < dist_old` branch (it compares with the WEIGHT volume, :191-228) is exercised."""
    H, W = TSDF_H, TSDF_W
    yaw = np.linspace(-np.pi, np.pi, W)
    out = []
    for k in range(n):
        rng = np.random.default_rng(700 + k)
        depth = (7.0 + 2.5 * np.sin(3 * yaw + 0.4 * k)[None, :] + 0.05 * rng.standard_normal((H, W))).astype(np.float32)
        depth[rng.random((H, W)) < 0.04] = 0.0
        lab = np.where(np.sin(5 * yaw + 0.9 * k)[None, :] + 0.2 * rng.standard_normal((H, W)) > 0, 40.0, 50.0)
        lab[rng.random((H, W)) < 0.05] = 10.0
        lab[rng.random((H, W)) < 0.03] = 0.0
        label3 = np.zeros((H, W, 3), np.float64)
        label3[:, :, 0] = lab
        rem = rng.random((H, W)).astype(np.float32)
        out.append((label3, depth, rem))
    return out


# ---- golden F12b (tests/golden/make_golden_deform_fuzz.py): random configurations of the closest-point adaption -------------
def deform_cp_case(k):
    rng = np.random.default_rng(5000 + k)
    H, W = int(rng.choice([8, 16, 32, 64])), int(rng.choice([64, 256, 500, 1024]))
    fu, fd = float(rng.choice([2.0, 3.0, 10.0, 15.0])), -float(rng.choice([16.6, 25.0, 30.0]))
    nscans = int(rng.integers(1, 4))
    sizes = [int(rng.integers(800, 9000)) for _ in range(nscans)]
    beams = bool(rng.random() < 0.33)
    pf = bool(k % 2)
    return H, W, fu, fd, sizes, beams, pf


def deform_cp_clouds(k):
    H, W, fu, fd, sizes, beams, pf = deform_cp_case(k)
    rng = np.random.default_rng(9000 + k)
    out = []
    for j, n in enumerate(sizes):
        pts, rem, lab = synth_cloud(7000 + 10 * k + j, n, dtype=np.float64, fov_up=fu + 2.0, fov_down=fd - 2.0)
        pts[rng.integers(0, n, 2)] = 0.0
        d = rng.integers(0, n, 20)
        pts[d[:10]] = pts[d[10:]]
        lab = np.where(lab == 0, 40, lab).astype(np.uint32)
        lab[rng.integers(0, n, max(n // 80, 1))] = 0
        out.append((pts, rem.astype(np.float32), lab))
    return out




# ---- golden F15 (tests/golden/make_golden_mc_full.py): marching cubes + ray cast at the DEFAULT volume size -----------------
MC_FULL_DIMS, MC_FULL_VOXEL, MC_FULL_ORIGIN = (2000, 2000, 200), 0.05, (-50.0, -50.0, -5.0)
MC_FULL_SENSOR = (64, 2048, 3.0, -25.0)
_MC_FULL_BOXES = [(9.0, 4.0, 2.5, 3.5, 2.25, 50.0), (-8.5, -6.0, 3.0, 2.0, 3.5, 50.0), (3.0, -11.0, 1.0, 2.25, 1.5, 10.0),
                  (-4.0, 10.0, 4.5, 1.5, 2.0, 50.0), (12.0, -9.0, 0.25, 0.25, 4.0, 80.0), (-13.0, 2.0, 2.0, 5.0, 2.75, 50.0),
                  (5.5, 13.0, 1.0, 2.25, 1.5, 10.0), (-2.0, -14.5, 6.0, 1.0, 3.0, 50.0)]


def mc_full_fields(xp, device=None):
    """(tsdf, color_vol, rem_vol) float32 [2000, 2000, 200] -- the reference's DEFAULT volume (config/lidar_transfer.yaml:6-9:
    +-50 m x +-50 m x +-5 m at 5 cm, 800 M voxels) holding a truncated signed distance field of a street scene: a piecewise
    linear ground, eight boxes / a pole, observed inside +-18 m of the sensor, +1 elsewhere and behind the truncation band
    as `integrate` leaves it.  `xp` is numpy (tests/golden/make_golden_mc_full.py, on the host) or torch (the GPU test, on the
    device): the field is built ONLY from +, -, *, abs, min, max, comparisons on float32 arrays (one rounding per operation in
    either library, no contraction across operations; the 1-D coordinate tables come from numpy on the host either way), so
    both produce the same 2.4 G floats bit for bit without 9.6 GB travelling in a fixture."""
    nx, ny, nz = MC_FULL_DIMS
    vs = MC_FULL_VOXEL
    ax = [(np.float64(MC_FULL_ORIGIN[k]) + np.arange(n, dtype=np.float64) * vs).astype(np.float32)
          for k, n in enumerate(MC_FULL_DIMS)]

    def tri(u, period):                          # triangle wave in [0, 1], exact operations only
        t = u.astype(np.float64) / period
        return np.abs(2.0 * (t - np.floor(t + 0.5))).astype(np.float32)

    if xp is np:
        arr = lambda a: np.ascontiguousarray(a, dtype=np.float32)            # noqa: E731
        full = lambda shape, v: np.full(shape, v, np.float32)                # noqa: E731
    else:
        arr = lambda a: xp.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)   # noqa: E731
        full = lambda shape, v: xp.full(shape, v, dtype=xp.float32, device=device)            # noqa: E731
    x, y, z = arr(ax[0]).reshape(nx, 1, 1), arr(ax[1]).reshape(1, ny, 1), arr(ax[2]).reshape(1, 1, nz)
    gx, gy = arr(tri(ax[0], 21.0)).reshape(nx, 1, 1), arr(tri(ax[1], 31.0)).reshape(1, ny, 1)
    h = (gx * np.float32(0.25)) * gy + np.float32(-1.73)     # ground height [nx, ny, 1]
    sd = z - h                                                # > 0 above the ground  [nx, ny, nz]
    lab = full((nx, ny, nz), 40.0)
    for cx, cy, hx, hy, top, lb in _MC_FULL_BOXES:
        dx = xp.abs(x - np.float32(cx)) - np.float32(hx)
        dy = xp.abs(y - np.float32(cy)) - np.float32(hy)
        d = xp.maximum(xp.maximum(dx, dy), z - np.float32(top))   # Chebyshev box distance
        lab = xp.where(d < sd, np.float32(lb) if xp is np else full((), lb), lab)
        sd = xp.minimum(sd, d)
    tsdf = xp.minimum(xp.maximum(sd * np.float32(4.0), np.float32(-1.0) if xp is np else full((), -1.0)),
                      np.float32(1.0) if xp is np else full((), 1.0))      # sd / (5 voxels) clipped to [-1, 1]
    one = np.float32(1.0) if xp is np else full((), 1.0)
    unseen = (sd < np.float32(-0.25)) | (xp.maximum(xp.abs(x), xp.abs(y)) > np.float32(18.0))
    tsdf = xp.where(unseen, one, tsdf)
    color = lab * np.float32(65536.0)
    rem = (arr(tri(ax[0], 7.0)).reshape(nx, 1, 1) * np.float32(0.5) + arr(tri(ax[1], 5.0)).reshape(1, ny, 1) * np.float32(0.25)) \
        + arr(tri(ax[2], 3.0)).reshape(1, 1, nz) * np.float32(0.125)
    rem = rem + (sd * np.float32(0.0))           # (broadcast to the full shape)
    if xp is np:
        return np.ascontiguousarray(tsdf), np.ascontiguousarray(color), np.ascontiguousarray(rem)
    return tsdf.contiguous(), color.contiguous(), rem.contiguous()


# ---- goldens F13b / F14b (tests/golden/make_golden_deform_mesh_fuzz.py): random configurations of the two mesh adaptions -----
N_DEFORM_MESH_CASES = 16


def deform_mesh_case(k):
    """(adaption, source (H, W, fu, fd), target, n_scans, vol_bnds [3,2] as the YAML would give it, voxel, scene seeds)"""
    rng = np.random.default_rng(7100 + k)
    adaption = "mesh" if k % 2 == 0 else "mergemesh"
    src = (int(rng.choice([16, 24, 32])), int(rng.choice([256, 360, 512])), float(rng.choice([3.0, 10.0, 15.0])),
           -float(rng.choice([20.0, 25.0, 30.0])))
    tgt = src if rng.random() < 0.3 else (int(rng.choice([8, 16, 32])), int(rng.choice([128, 256, 500])),
                                          float(rng.choice([2.0, 10.0])), -float(rng.choice([16.6, 25.0, 30.0])))
    n_scans = int(rng.integers(1, 4))
    voxel = float(rng.choice([0.1, 0.2, 0.25]))
    ext = int(rng.choice([6, 8, 10])) if voxel < 0.2 else int(rng.choice([8, 10, 12]))
    bnds = np.array([[-ext, ext], [-ext + int(rng.integers(0, 3)), ext], [-int(rng.integers(2, 4)), int(rng.integers(1, 4))]])
    if rng.random() < 0.4:
        bnds = bnds.astype(np.float64)       # (a caller that passes floats: no truncation in fusion_lidar.py:36)
    seeds = (int(rng.integers(100, 10000)), int(rng.integers(100, 10000)))
    return adaption, src, tgt, n_scans, bnds, voxel, seeds


def deform_mesh_clouds(seed, n_scans, src, render):
    """Source scans of a seeded street scene as the source sensor sees it from the origin: the hit points of `render(verts,
    faces, colors, rem, H, W, fu, fd) -> (endpoints [R,3] f32, label [R] i32, rem [R] f32, range [R] f32)` -- the reference's
    raytracer in the generator, this library's (bit-identical, goldens F2-F5) in the GPU test -- and noisy, thinned copies."""
    from lidar_transfer_amd.synth import synth_scene
    H, W, fu, fd = src
    v, f, c, r = synth_scene(seed % 64, 20000, bounds=(-12, 12, -12, 12, -3, 3), n_boxes=6, n_poles=6)
    ends, lab, rem, rng_im = render(v, f, c, r, H, W, fu, fd)
    hit = np.asarray(rng_im) > 0
    pts0 = np.asarray(ends, np.float32).reshape(-1, 3)[hit].astype(np.float64)
    lab0 = np.asarray(lab).reshape(-1)[hit].astype(np.uint32)
    rem0 = np.asarray(rem, np.float32).reshape(-1)[hit]
    rng = np.random.default_rng(seed)
    scans = []
    for k in range(n_scans):
        keep = rng.random(len(pts0)) > (0.0 if k == 0 else 0.1)
        p = pts0[keep] * (1.0 + (rng.normal(0, 0.002, (int(keep.sum()), 1)) if k else 0.0))
        lb = lab0[keep].copy()
        if k:
            lb[rng.random(len(lb)) < 0.02] = 50
        p = np.concatenate([p, np.zeros((1, 3))])
        lb = np.concatenate([lb, [40]]).astype(np.uint32)
        rm = np.concatenate([rem0[keep], [0.5]]).astype(np.float32)
        scans.append((np.ascontiguousarray(p), rm, lb))
    return scans


# ---- golden F14c (tests/golden/make_golden_mergemesh_seq.py): SEQUENCES of mergemesh output scans on one bounds array ----------
N_MERGEMESH_SEQ_CASES = 4
MERGEMESH_SEQ_LEN = 6


def mergemesh_seq_case(k):
    """(source, target, vol_bnds [3,2] ints or floats, voxel, scene seed, crop limits per scan): six output scans in a row on ONE
    bounds array (laserscan.py:957-962 clips it scan after scan; fusion_lidar.py:36 rewrites its upper bounds); the clouds are
    cropped by limits that move in and out again, so the bounds shrink at several scans and then stay (they never grow)."""
    rng = np.random.default_rng(9100 + k)
    src = (int(rng.choice([16, 24, 32])), int(rng.choice([256, 360, 512])), float(rng.choice([3.0, 10.0])),
           -float(rng.choice([20.0, 25.0])))
    tgt = src if k % 2 == 0 else (int(rng.choice([16, 32])), int(rng.choice([256, 500])), float(rng.choice([2.0, 10.0])),
                                  -float(rng.choice([25.0, 30.0])))
    voxel = float([0.1, 0.2, 0.25, 0.1][k])
    ext = 10 if voxel < 0.2 else 12
    bnds = np.array([[-ext, ext], [-ext + 1, ext], [-3, 3]])
    if k >= 2:
        bnds = bnds.astype(np.float64) + (0.0 if k == 2 else 0.13)   # (floats; k == 3: not on the voxel lattice)
    seed = int(rng.integers(100, 10000))
    # per scan: (x_max, y_min) the cloud is cropped to -- moving in, out again (the bounds must NOT follow), further in
    limits = [(9.4, -9.6), (7.4, -9.6), (8.6, -6.6), (7.4, -6.6), (5.6, -7.4), (5.6, -4.4)]
    return src, tgt, bnds, voxel, seed, limits


def mergemesh_seq_clouds(seed, src, render, limit):
    """the one source scan of a sequence's output scan: deform_mesh_clouds' first scan, cropped to x < limit[0], y > limit[1]"""
    p, rm, lb = deform_mesh_clouds(seed, 1, src, render)[0]
    keep = (p[:, 0] < limit[0]) & (p[:, 1] > limit[1])
    return [(np.ascontiguousarray(p[keep]), np.ascontiguousarray(rm[keep]), np.ascontiguousarray(lb[keep]))]
