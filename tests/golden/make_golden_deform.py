#!/usr/bin/env python3
"""Generate golden fixture F12 -- the reference's OWN loop body for the closest-point adaption, as a whole:
`MultiSemLaserScan.deform('cp', poses, idx)` (auxiliary/laserscan.py:827-861: the scans merged into one cloud, the inverse pose,
`do_range_projection_new` + `do_label_projection_new` + `do_reverse_projection_new` of the TARGET sensor) followed by
`MultiSemLaserScan.write(out_dir, idx)` (:1121-1178: the filters and the per-point `struct.pack` loops) -- on three seeded
source scans (identity poses: pose handling is out of scope), for `preserve_float` False and True.

    python tests/golden/make_golden_deform.py        # needs /root/reference (LT_REFERENCE) -- numpy + torch, no GPU

`f12_deform_cp.npz`: the three input clouds (points float64 as `apply_inv_pose` leaves them, remissions float32, labels uint32),
the sensor models, and per variant the BYTES of `velodyne/NNNNNN.bin` and `labels/NNNNNN.label` plus the images `deform`
leaves on the object (`proj_range`, `label_image`, `index`).  tests/test_deform_gpu.py feeds the clouds to
`DeviceDeform.cp` and compares the bytes.  Only data is written."""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import make_golden  # noqa: E402
from lidar_transfer_amd.synth import synth_cloud  # noqa: E402

COLOR_DICT = {0: [0, 0, 0], 10: [245, 150, 100], 40: [255, 0, 255], 48: [75, 0, 75], 50: [0, 200, 255],
              70: [0, 175, 0], 80: [150, 240, 255]}
SOURCE = dict(name="src", beams=32, fov_hor=360.0, angle_res_hor=360.0 / 512, fov_up=3.0, fov_down=-25.0)
TARGET = dict(name="tgt", beams=16, fov_hor=360.0, angle_res_hor=360.0 / 256, fov_up=10.0, fov_down=-30.0)


def main():
    ls, fl = make_golden.import_reference()
    out = dict(source=np.array([SOURCE["beams"], 512, SOURCE["fov_up"], SOURCE["fov_down"]], np.float64),
               target=np.array([TARGET["beams"], 256, TARGET["fov_up"], TARGET["fov_down"]], np.float64))
    clouds = []
    for k in range(3):
        pts, rem, lab = synth_cloud(40 + k, 6000, dtype=np.float64, fov_up=12.0, fov_down=-32.0)
        pts[5] = 0.0                                   # a depth-0 point: removed by the projection
        lab = np.where(lab == 0, 40, lab).astype(np.uint32)
        lab[::97] = 0                                  # unlabeled points: index > 0 keeps them, label 0 is written
        clouds.append((pts, rem.astype(np.float32), lab))
        out[f"points{k}"], out[f"rem{k}"], out[f"label{k}"] = pts, rem.astype(np.float32), lab
    poses = np.stack([np.eye(4, dtype=np.float32)] * 3)
    for pf in (False, True):
        ms = ls.MultiSemLaserScan(dict(SOURCE), dict(TARGET), 3, 300, [], [], color_dict=COLOR_DICT, transformation=None,
                                  preserve_float=pf, voxel_size=0.1, vol_bnds=None)
        for scan, (pts, rem, lab) in zip(ms.scans, clouds):
            scan.points, scan.remissions, scan.label = pts.copy(), rem.copy(), lab.copy()
            scan.colorize()
            scan.pose = np.eye(4, dtype=np.float32)
        ms.deform("cp", poses, 0)
        with tempfile.TemporaryDirectory() as d:
            os.makedirs(os.path.join(d, "velodyne"))
            os.makedirs(os.path.join(d, "labels"))
            ms.write(d, 5)
            tag = "float" if pf else "int"
            out[f"bin_{tag}"] = np.fromfile(os.path.join(d, "velodyne", "000005.bin"), np.uint8)
            out[f"label_{tag}"] = np.fromfile(os.path.join(d, "labels", "000005.label"), np.uint8)
        out[f"proj_range_{tag}"] = np.asarray(ms.proj_range, np.float32)
        out[f"label_image_{tag}"] = np.asarray(ms.label_image)
        out[f"index_{tag}"] = np.asarray(ms.index)
        print(tag, "points written", out[f"bin_{tag}"].size // 16, "cells filled", int((np.asarray(ms.proj_range) > 0).sum()))
    np.savez_compressed(os.path.join(HERE, "f12_deform_cp.npz"), **out)


if __name__ == "__main__":
    main()
