#!/usr/bin/env python3
"""Generate golden fixture F12b -- the reference's closest-point loop body (`MultiSemLaserScan.deform('cp')` + `write()`,
auxiliary/laserscan.py:827-861, :1121-1178) on 24 random configurations: target image shape and field of view, 1-3 source
scans of random size (regenerated from their seeds by lidar_transfer_amd.synth.synth_cloud: not stored), beam tables for a
third of them, `preserve_float` on and off, depth-0 points, exact duplicates, unlabeled points.

    python tests/golden/make_golden_deform_fuzz.py        # needs /root/reference (LT_REFERENCE)

`f12b_deform_cp_fuzz.npz`: per case the parameters and the SHA-256 of the bytes of velodyne/N.bin and labels/N.label, of the
`index` image and of `back_points` rounded to float32 (the float64 sin / cos of two math libraries differ in the last ulp of
the double; what `write` packs is the float32), plus the number of points written.  tests/test_deform_gpu.py rebuilds the clouds from the
seeds, runs `DeviceDeform.cp` and compares the digests.  Only data is written."""
import hashlib
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

sys.path.insert(0, os.path.join(ROOT, "tests"))
import make_golden  # noqa: E402
from pin_cases import deform_cp_case as case_params, deform_cp_clouds as clouds_of  # noqa: E402

COLOR_DICT = {0: [0, 0, 0], 10: [245, 150, 100], 40: [255, 0, 255], 48: [75, 0, 75], 50: [0, 200, 255],
              70: [0, 175, 0], 80: [150, 240, 255]}


def main():
    ls, _ = make_golden.import_reference()
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()  # noqa: E731
    rec = dict(n_cases=24)
    for k in range(24):
        H, W, fu, fd, sizes, beams, pf = case_params(k)
        tgt = dict(name="t", beams=H, fov_hor=360.0, angle_res_hor=360.0 / W, fov_up=fu, fov_down=fd)
        src = dict(tgt)
        if beams:  # (the reference reads the TARGET's beam angles from the SOURCE config, laserscan.py:744)
            src["beam_angles"] = sorted((np.linspace(fd, fu, H) / 180.0 * np.pi).tolist())
        nscans = len(sizes)
        ms = ls.MultiSemLaserScan(src, tgt, nscans, 300, [], [], color_dict=COLOR_DICT, transformation=None, preserve_float=pf,
                                  voxel_size=0.1, vol_bnds=None)
        assert ms.t_W == W, (ms.t_W, W)
        for scan, (pts, rem, lab) in zip(ms.scans, clouds_of(k)):
            scan.points, scan.remissions, scan.label = pts.copy(), rem.copy(), lab.copy()
            scan.colorize()
            scan.pose = np.eye(4, dtype=np.float32)
        ms.deform("cp", np.stack([np.eye(4, dtype=np.float32)] * nscans), 0)
        with tempfile.TemporaryDirectory() as d:
            os.makedirs(os.path.join(d, "velodyne"))
            os.makedirs(os.path.join(d, "labels"))
            ms.write(d, 1)
            b = np.fromfile(os.path.join(d, "velodyne", "000001.bin"), np.uint8)
            l = np.fromfile(os.path.join(d, "labels", "000001.label"), np.uint8)
        rec[f"sha_bin_{k}"], rec[f"sha_label_{k}"] = sha(b), sha(l)
        rec[f"sha_index_{k}"] = sha(np.asarray(ms.index))
        rec[f"sha_back32_{k}"] = sha(np.asarray(ms.back_points, np.float64).astype(np.float32))
        rec[f"n_written_{k}"] = b.size // 16
        print(k, (H, W, fu, fd), sizes, "beams" if beams else "", "float" if pf else "int", "written", b.size // 16)
    np.savez_compressed(os.path.join(HERE, "f12b_deform_cp_fuzz.npz"), **rec)


if __name__ == "__main__":
    main()
