#!/usr/bin/env python3
"""Generate goldens F13b / F14b -- the reference's own `deform('mesh' | 'mergemesh')` + `write()` (as make_golden_deform_mesh.py
runs them: numpy fusion mode, real scikit-image, the C++ raytracer compiled in place) on 16 random configurations
(tests/pin_cases.py::deform_mesh_case: sensor models, 1-3 source scans, volume bounds given as ints or floats, voxel sizes;
`mergemesh` cases run TWO output scans in a row on one bounds array).  The source clouds are not stored: they are the hit points
of a seeded scene seen through the reference's raytracer, which this library's render reproduces bit for bit, so the GPU test
rebuilds them.

    /opt/conda/bin/python3.9 tests/golden/make_golden_deform_mesh_fuzz.py

`f13b_deform_mesh_fuzz.npz`: per output scan SHA-256 of the bytes of velodyne/N.bin and labels/N.label, of `proj_range` and
`label_image`, the volume's dimensions, the bounds array afterwards, the numbers of written voxels / faces / points, and the two images.  Only data."""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

import make_golden  # noqa: E402
import make_golden_deform_mesh as gm  # noqa: E402
import pin_cases  # noqa: E402


def main():
    from skimage import measure
    if not hasattr(measure, "marching_cubes_lewiner"):
        measure.marching_cubes_lewiner = lambda vol, level=0.0, **kw: measure.marching_cubes(vol, level=level, method="lewiner", **kw)
    ls, fl = make_golden.import_reference(stub_skimage=False)
    assert fl.FUSION_GPU_MODE == 0
    orig_integrate = fl.TSDFVolume.integrate
    fl.TSDFVolume.integrate = lambda self, c, d, r, pose, obs_weight=1.: orig_integrate(self, c, d, r, np.eye(4), obs_weight=obs_weight)
    import auxiliary.raytracer.RayTracerCython as rtc

    def render(v, f, c, r, H, W, fu, fd):
        rays = ls.MultiSemLaserScan.create_rays(None, fu, fd, H, W).reshape(-1)
        n = H * W
        ends, cols = np.zeros(3 * n, np.float32), np.zeros(3 * n, np.int32)
        rng_im, rem_im = np.zeros(n, np.float32), np.zeros(n, np.float32)
        rtc.C_Trace(rays, np.zeros(3, np.float32), np.ascontiguousarray(v.reshape(-1)), np.ascontiguousarray(f.reshape(-1)),
                    np.ascontiguousarray(c.reshape(-1)), np.ascontiguousarray(r), ends, cols, rng_im, rem_im, H, W)
        return ends.reshape(-1, 3), cols.reshape(-1, 3)[:, 2], rem_im, rng_im

    sha = gm.sha
    rec = dict(n_cases=pin_cases.N_DEFORM_MESH_CASES)
    for k in range(pin_cases.N_DEFORM_MESH_CASES):
        adaption, src, tgt, n_scans, bnds, voxel, seeds = pin_cases.deform_mesh_case(k)
        bnds = bnds.copy()
        for step in range(2 if adaption == "mergemesh" else 1):
            clouds = pin_cases.deform_mesh_clouds(seeds[step], n_scans, src, render)
            out = {}
            tag = f"c{k}s{step}"
            gm.run(ls, fl, adaption, gm.sensor("s", *src), gm.sensor("t", *tgt), clouds, bnds, voxel, out, tag)
            rec[f"{tag}_sha"] = np.array([sha(out[f"{tag}_bin"]), sha(out[f"{tag}_label"]), sha(out[f"{tag}_proj_range"]),
                                          sha(out[f"{tag}_label_image"])])
            rec[f"{tag}_vol_dim"] = out[f"{tag}_vol_dim"]
            rec[f"{tag}_bnds_after"] = out[f"{tag}_bnds_after"]
            rec[f"{tag}_counts"] = np.array([out[f"{tag}_n_written"], out[f"{tag}_n_faces"], out[f"{tag}_bin"].size // 16,
                                             int((out[f"{tag}_proj_range"] > 0).sum())])
            rec[f"{tag}_cloud_sha"] = np.array(sha(np.concatenate([c[0].reshape(-1) for c in clouds])))
            # the images themselves (a few KB each): where a boundary voxel (numpy's last-bit arctan2 / arcsin) reaches the mesh the
            # consumer must be able to say HOW MANY pixels it moved
            assert out[f"{tag}_label_image"].max() < 256
            rec[f"{tag}_range"], rec[f"{tag}_limg"] = out[f"{tag}_proj_range"], out[f"{tag}_label_image"].astype(np.uint8)
    np.savez_compressed(os.path.join(HERE, "f13b_deform_mesh_fuzz.npz"), **rec)
    print("f13b_deform_mesh_fuzz.npz", os.path.getsize(os.path.join(HERE, "f13b_deform_mesh_fuzz.npz")), "bytes")


if __name__ == "__main__":
    main()
