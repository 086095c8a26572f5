#!/usr/bin/env python3
"""Generate golden F14c -- SEQUENCES of the reference's own `deform('mergemesh')` + `write()` (as make_golden_deform_mesh.py
runs them: numpy fusion mode, real scikit-image, the C++ raytracer compiled in place): four configurations
(tests/pin_cases.py::mergemesh_seq_case: integer and float bounds arrays, on and off the voxel lattice) x SIX output scans in
a row on ONE bounds array, the clouds cropped by limits that move in and out again -- what pins the order-dependent bounds
bookkeeping (laserscan.py:957-962, fusion_lidar.py:33-37) that round 6 moved onto the device.

    /opt/conda/bin/python3.9 tests/golden/make_golden_mergemesh_seq.py

`f14c_mergemesh_seq.npz`: per output scan SHA-256 of the bytes of velodyne/N.bin and labels/N.label, of `proj_range` and
`label_image`, the volume's dimensions, the bounds array afterwards, counts, the two images.  Only data."""
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
    rec = dict(n_cases=pin_cases.N_MERGEMESH_SEQ_CASES, seq_len=pin_cases.MERGEMESH_SEQ_LEN)
    for k in range(pin_cases.N_MERGEMESH_SEQ_CASES):
        src, tgt, bnds, voxel, seed, limits = pin_cases.mergemesh_seq_case(k)
        bnds = bnds.copy()
        for step, lim in enumerate(limits):
            clouds = pin_cases.mergemesh_seq_clouds(seed, src, render, lim)
            out = {}
            tag = f"q{k}s{step}"
            gm.run(ls, fl, "mergemesh", gm.sensor("s", *src), gm.sensor("t", *tgt), clouds, bnds, voxel, out, tag)
            rec[f"{tag}_sha"] = np.array([sha(out[f"{tag}_bin"]), sha(out[f"{tag}_label"]), sha(out[f"{tag}_proj_range"]),
                                          sha(out[f"{tag}_label_image"])])
            rec[f"{tag}_vol_dim"] = out[f"{tag}_vol_dim"]
            rec[f"{tag}_bnds_after"] = out[f"{tag}_bnds_after"]
            rec[f"{tag}_counts"] = np.array([out[f"{tag}_n_written"], out[f"{tag}_n_faces"], out[f"{tag}_bin"].size // 16,
                                             int((out[f"{tag}_proj_range"] > 0).sum())])
            rec[f"{tag}_cloud_sha"] = np.array(sha(np.concatenate([c[0].reshape(-1) for c in clouds])))
            assert out[f"{tag}_label_image"].max() < 256
            rec[f"{tag}_range"], rec[f"{tag}_limg"] = out[f"{tag}_proj_range"], out[f"{tag}_label_image"].astype(np.uint8)
            print(tag, "vol_dim", out[f"{tag}_vol_dim"], "bnds_after", np.asarray(out[f"{tag}_bnds_after"]).reshape(-1).tolist(),
                  "faces", out[f"{tag}_n_faces"], flush=True)
    np.savez_compressed(os.path.join(HERE, "f14c_mergemesh_seq.npz"), **rec)
    print("f14c_mergemesh_seq.npz", os.path.getsize(os.path.join(HERE, "f14c_mergemesh_seq.npz")), "bytes")


if __name__ == "__main__":
    main()
