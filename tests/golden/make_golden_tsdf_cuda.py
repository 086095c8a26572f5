#!/usr/bin/env python3
"""Generate golden fixture F11 -- the reference's CUDA kernel `integrate` (auxiliary/fusion_lidar.py:66-229, source in a
Python string compiled by pycuda) on the multi-class observations of tests/pin_cases.py: the class-aware branch
(`merge == true`, the only one the reference runs, :177) with voxels that see DIFFERENT classes in successive observations.

CANNOT RUN IN THE BUILD IMAGE (no pycuda, no NVIDIA GPU): this is the recipe a maintainer with the reference's
environment runs once; tests/test_pin_f10_f11_gpu.py skips until the file exists.

    pip install pycuda                  # + an NVIDIA GPU and nvcc: the reference's GPU mode (fusion_lidar.py:10-18)
    LT_REFERENCE=/path/to/lidar_transfer python tests/golden/make_golden_tsdf_cuda.py
    git add tests/golden/f11_tsdf_cuda.npz

What runs is the reference itself: `TSDFVolume(vol_bnds, voxel_size, fov_up, fov_down)` in GPU mode, `integrate(label3,
depth, rem, np.eye(4))` per observation exactly as laserscan.py:890-899 calls it, then the four device volumes copied
back (`get_volume` returns tsdf / colour / remission, :395-400; the weight volume is read with the same memcpy).
Only data is written: after every observation the volumes, plus the nvcc version (the FMA contraction of the kernel's
`a * b + c` expressions is nvcc's default -fmad=true -- the thing lt_tsdf.hip's __fmaf_rn pattern assumes)."""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

import make_golden  # noqa: E402
import pin_cases  # noqa: E402


def main():
    try:
        import pycuda.driver as cuda
        import pycuda.autoinit  # noqa: F401
    except Exception as e:  # noqa: BLE001
        raise SystemExit(f"make_golden_tsdf_cuda.py needs pycuda and an NVIDIA GPU (the reference's GPU mode): {e}")
    ls, fl = make_golden.import_reference(stub_skimage=True)   # marching cubes is not on this path
    assert fl.FUSION_GPU_MODE == 1, "the reference fell back to its numpy CPU mode (which has no class-aware branch)"
    vol = fl.TSDFVolume(pin_cases.TSDF_BOUNDS.copy(), pin_cases.TSDF_VOXEL, *pin_cases.TSDF_FOV)
    out = {}
    for k, (label3, depth, rem) in enumerate(pin_cases.tsdf_observations()):
        vol.integrate(label3, depth, rem, np.eye(4), obs_weight=1.)
        tsdf, color, remv = [a.copy() for a in vol.get_volume()]
        weight = np.empty_like(vol._weight_vol_cpu)
        cuda.memcpy_dtoh(weight, vol._weight_vol_gpu)
        out.update({f"tsdf_{k}": tsdf, f"weight_{k}": weight, f"color_{k}": color, f"rem_{k}": remv})
    try:
        nvcc = subprocess.run(["nvcc", "--version"], capture_output=True, text=True).stdout.strip().splitlines()[-1]
    except OSError:
        nvcc = "unknown"
    np.savez_compressed(os.path.join(HERE, "f11_tsdf_cuda.npz"), n_obs=len(pin_cases.tsdf_observations()),
                        nvcc=np.array(nvcc), device=np.array(cuda.Device(0).name()), **out)
    print("written f11_tsdf_cuda.npz;", nvcc)


if __name__ == "__main__":
    main()
