#!/usr/bin/env python3
"""Generate golden fixture F15 -- the tail of the reference's `deform('mesh' | 'mergemesh')` at its DEFAULT volume size:
`TSDFVolume.get_mesh` (auxiliary/fusion_lidar.py:403-424: scikit-image's Lewiner marching cubes over 2000 x 2000 x 200 voxels +
the attribute look-ups) and `throw_rays_at_mesh` (:426-455 -> the C++ raytracer, 64 x 2048 rays from its own `create_rays`) on the
seeded street field of tests/pin_cases.py::mc_full_fields.

    /opt/conda/bin/python3.9 tests/golden/make_golden_mc_full.py        # scikit-image 0.18.3; ~25 GB of host memory, minutes

What runs is the reference itself (make_golden.import_reference, scikit-image NOT stubbed; the raytracer compiled in place by
oracle/Makefile with strict IEEE flags).  `f15_mc_full.npz`: the mesh sizes and SHA-256 of the four arrays `get_mesh` returns
AS THEY ARE (scikit-image's vertex order), and the images the reference renders from them.  Only data is written.  The GPU test
(tests/test_pin_f10_f11_gpu.py) rebuilds the field on the device from the same expressions, extracts, renumbers, hashes, renders."""
import hashlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

import make_golden  # noqa: E402
import pin_cases  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    import skimage
    from skimage import measure
    if not hasattr(measure, "marching_cubes_lewiner"):
        measure.marching_cubes_lewiner = lambda vol, level=0.0, **kw: measure.marching_cubes(vol, level=level, method="lewiner", **kw)
    ls, fl = make_golden.import_reference(stub_skimage=False)
    fl.FUSION_GPU_MODE = 0
    t0 = time.time()
    tsdf, color, rem = pin_cases.mc_full_fields(np)
    print("fields", tsdf.shape, f"{time.time() - t0:.0f} s; below the level: {(tsdf < 0).mean():.4f}", flush=True)
    vol = object.__new__(fl.TSDFVolume)
    vol._tsdf_vol_cpu, vol._color_vol_cpu, vol._rem_vol_cpu = tsdf, color, rem
    vol._voxel_size, vol._vol_origin = float(pin_cases.MC_FULL_VOXEL), np.array(pin_cases.MC_FULL_ORIGIN, np.float32)
    H, W, fu, fd = pin_cases.MC_FULL_SENSOR
    rays = ls.MultiSemLaserScan.create_rays(None, fu, fd, H, W)
    t0 = time.time()
    endpoints, ray_colors, verts, colors, faces, range_image, rem_image = vol.throw_rays_at_mesh(rays, np.zeros(3, np.float32), H, W, None)
    print(f"get_mesh + throw_rays_at_mesh {time.time() - t0:.0f} s: {len(verts)} vertices, {len(faces)} faces, "
          f"{int((range_image > 0).sum())} hits", flush=True)
    _, _, _, _, vrem = vol.get_mesh(None) if False else (None, None, None, None, None)
    # (the remission per vertex is not in the 7-tuple: the same look-up as fusion_lidar.py:418)
    verts_ind = np.round((np.asarray(verts) - vol._vol_origin) / vol._voxel_size).astype(int)
    out = dict(n_verts=len(verts), n_faces=len(faces), skimage_version=np.array(skimage.__version__),
               mesh_sha=np.array([sha(np.asarray(verts, np.float32)), sha(np.asarray(faces, np.int32)), sha(np.asarray(colors, np.uint8))]),
               H=H, W=W, fov_up=fu, fov_down=fd, range=np.asarray(range_image, np.float32),
               label=np.asarray(ray_colors, np.int32).reshape(H, W, 3)[:, :, 2].astype(np.uint8),
               label_max=int(np.asarray(ray_colors).max()), rem=np.asarray(rem_image, np.float32),
               endpoints_sha=np.array(sha(np.asarray(endpoints, np.float32))))
    np.savez_compressed(os.path.join(HERE, "f15_mc_full.npz"), **out)
    print("f15_mc_full.npz", os.path.getsize(os.path.join(HERE, "f15_mc_full.npz")), "bytes")


if __name__ == "__main__":
    main()
