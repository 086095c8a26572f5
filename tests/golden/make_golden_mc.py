#!/usr/bin/env python3
"""Generate golden fixture F10 -- the reference's `TSDFVolume.get_mesh` (auxiliary/fusion_lidar.py:403-424), i.e.
scikit-image's Lewiner marching cubes (:407) + the attribute look-ups (:409-423), and the image the reference's own
raytracer renders from that mesh -- for the seeded volumes of tests/pin_cases.py.

Needs an interpreter with scikit-image 0.18.x (the last series that has `measure.marching_cubes_lewiner`, the name the
reference calls).  The build image has one beside the system Python: /opt/conda/bin/python3.9 (scikit-image 0.18.3, numpy
1.26, no torch -- make_golden.import_reference gives laserscan.py's unused `import torch` an empty module):

    /opt/conda/bin/python3.9 tests/golden/make_golden_mc.py
    git add tests/golden/f10_mc_*.npz

What runs is the reference itself: `auxiliary.fusion_lidar.TSDFVolume.get_mesh` on a volume object whose CPU arrays are
the seeded fields, then its `throw_rays_at_mesh` (:426-455) -> `C_Trace` = the reference's C++ raytracer compiled in place
by oracle/Makefile (strict IEEE flags), with rays from its own `create_rays`.  scikit-image >= 0.19 renamed
`marching_cubes_lewiner(vol, level)` to `marching_cubes(vol, level, method="lewiner")` (same Cython implementation); the
alias below bridges that and is recorded in the fixture.  Only data is written.

Per case `f10_mc_<name>.npz`:
    verts [V,3] f32, faces [F,3] i32, colors [V,3] u8, vrem [V] f32  `get_mesh`'s return values AS THEY ARE (values and order)
    verts_sorted [V,3] f32   world vertices, rows sorted lexicographically (x, y, z) -- the vertex SET; skimage's order is
                             not part of the contract (the order of lt_mc.hip is its own)
    colors_sorted [V,3] u8, rem_sorted [V] f32     attributes in the same order
    n_faces, face_area_sum, skimage_version, used_alias
    range [H,W] f32, label [H,W] i32, rem [H,W] f32   the reference's rendering of ITS mesh (cases with a sensor model)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

import make_golden  # noqa: E402  (the import shims of the other fixtures; skimage NOT stubbed here)
import pin_cases  # noqa: E402


def main():
    try:
        import skimage
        from skimage import measure
    except ImportError:
        raise SystemExit("make_golden_mc.py needs scikit-image 0.18.x (the reference's dependency): /opt/conda/bin/python3.9 has it")
    used_alias = False
    if not hasattr(measure, "marching_cubes_lewiner"):
        used_alias = True
        measure.marching_cubes_lewiner = lambda vol, level=0.0, **kw: measure.marching_cubes(vol, level=level, method="lewiner", **kw)
    ls, fl = make_golden.import_reference(stub_skimage=False)
    fl.FUSION_GPU_MODE = 0   # (get_volume: no device copies)
    create_rays = lambda *a: ls.MultiSemLaserScan.create_rays(None, *a)  # noqa: E731
    for name, sensor in pin_cases.MC_CASES.items():
        tsdf, color, rem, vs, org = pin_cases.mc_case(name)
        vol = object.__new__(fl.TSDFVolume)          # the volume object without its allocations: the seeded fields instead
        vol._tsdf_vol_cpu, vol._color_vol_cpu, vol._rem_vol_cpu = tsdf, color, rem
        vol._voxel_size, vol._vol_origin = float(vs), org.astype(np.float32)
        verts, faces, norms, colors, vrem = vol.get_mesh(None)
        verts = np.asarray(verts, np.float32)
        order = np.lexsort((verts[:, 2], verts[:, 1], verts[:, 0]))
        tri = verts[np.asarray(faces)]
        area = 0.5 * np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1)
        out = dict(verts=verts, faces=np.asarray(faces, np.int32), colors=np.asarray(colors, np.uint8), vrem=np.asarray(vrem, np.float32),
                   verts_sorted=verts[order], colors_sorted=np.asarray(colors)[order], rem_sorted=np.asarray(vrem, np.float32)[order],
                   n_faces=len(faces), face_area_sum=float(area.sum()), skimage_version=np.array(skimage.__version__),
                   used_alias=used_alias, voxel_size=float(vs), vol_origin=org)
        if sensor is not None:
            H, W, fu, fd = sensor
            rays = create_rays(fu, fd, H, W)
            origin = np.zeros(3, np.float32)
            endpoints, ray_colors, _, _, _, range_image, rem_image = vol.throw_rays_at_mesh(rays, origin, H, W, None)
            out.update(H=H, W=W, fov_up=fu, fov_down=fd, range=np.asarray(range_image, np.float32),
                       label=np.asarray(ray_colors, np.int32).reshape(H, W, 3)[:, :, 2].copy(),
                       rem=np.asarray(rem_image, np.float32))
        np.savez_compressed(os.path.join(HERE, f"f10_mc_{name}.npz"), **out)
        print(name, "verts", len(verts), "faces", len(faces), "skimage", skimage.__version__)


if __name__ == "__main__":
    main()
