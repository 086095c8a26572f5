#!/usr/bin/env python3
"""Secondary measurements (one JSON line each) for the rows either side of the ray cast:
spherical z-min projection, TSDF integration at the reference's default volume, scan packing.
HIP-event timed on cuda:0, inputs resident in HBM."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from lidar_transfer_amd import _lib  # noqa: E402
from lidar_transfer_amd.fusion import TSDFVolume  # noqa: E402
from lidar_transfer_amd.synth import synth_cloud  # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.load()
st = torch.cuda.current_stream(dev)
sp = C.c_void_p(st.cuda_stream)


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


# ---- z-min projection: 120 k points x 5 scans -> 64 x 2048 (do_range_projection_new of the merged cloud) ------
H, W = 64, 2048
for dtype, nscan in ((np.float32, 1), (np.float64, 5)):
    pts, rem, lab = synth_cloud(0, 120000 * nscan, dtype=dtype)
    n = pts.shape[0]
    tp, tr, tl = [torch.from_numpy(x).to(dev) for x in (pts, rem, lab.astype(np.int32))]
    ft = torch.float32 if dtype == np.float32 else torch.float64
    o = dict(points=torch.empty((n, 3), dtype=ft, device=dev), rem=torch.empty(n, dtype=torch.float32, device=dev),
             label=torch.empty(n, dtype=torch.int32, device=dev), depth=torch.empty(n, dtype=ft, device=dev),
             px=torch.empty(n, dtype=torch.int32, device=dev), py=torch.empty(n, dtype=torch.int32, device=dev),
             xf=torch.empty(n, dtype=ft, device=dev), yf=torch.empty(n, dtype=ft, device=dev))
    img = dict(idx=torch.empty(H * W, dtype=torch.int32, device=dev), rng=torch.empty(H * W, dtype=torch.float32, device=dev),
               xyz=torch.empty(H * W * 3, dtype=torch.float32, device=dev), rem=torch.empty(H * W, dtype=torch.float32, device=dev),
               lab=torch.empty(H * W, dtype=torch.int32, device=dev))
    kept = C.c_int(0)

    def run():
        rc = lib.lt_range_projection_dev(tp.data_ptr(), int(dtype == np.float64), tr.data_ptr(), tl.data_ptr(), n, 3.0,
                                         -25.0, H, W, None, 0, _lib.LT_PROJ_NEW | _lib.LT_PROJ_REMOVE, None, 0,
                                         o["points"].data_ptr(), o["rem"].data_ptr(), o["label"].data_ptr(),
                                         o["depth"].data_ptr(), o["px"].data_ptr(), o["py"].data_ptr(),
                                         o["xf"].data_ptr(), o["yf"].data_ptr(), img["idx"].data_ptr(),
                                         img["rng"].data_ptr(), img["xyz"].data_ptr(), img["rem"].data_ptr(),
                                         img["lab"].data_ptr(), None, None, 0.0, -1.0, 0.0, C.byref(kept), sp)
        assert rc == 0
    ms = timed(run, 50)
    es = 4 if dtype == np.float32 else 8
    alg = n * (3 * es + 8) + n * (3 * es + 8 + es + 8 + 2 * es) + H * W * (8 + 28)
    print(json.dumps({"metric": "z-min spherical projection (do_range_projection_new)", "points": n,
                      "dtype": "f32" if es == 4 else "f64", "image": f"{H}x{W}", "ms": round(ms, 4),
                      "Mpoints_per_s": round(n / ms / 1e3, 1), "algorithmic_GBs": round(alg / ms / 1e6, 1),
                      "kept": kept.value, "reference_python_ms": 510 if nscan == 1 else None}))

# ---- TSDF integration at the reference's default volume (config/lidar_transfer.yaml: 2000 x 2000 x 200, 5 cm) ---
vol = TSDFVolume(np.array([[-50.0, 50.0], [-50.0, 50.0], [-5.0, 5.0]]), 0.05, 3.0, -25.0)
folded = torch.full((64, 2048), 40.0, device=dev)
depth = torch.full((64, 2048), 12.0, device=dev)
remi = torch.full((64, 2048), 0.5, device=dev)
nvox = int(np.prod(vol._vol_dim))


def run_t():
    assert lib.lt_tsdf_integrate_dev(vol._h, folded.data_ptr(), depth.data_ptr(), remi.data_ptr(), 64, 2048, 1.0,
                                     _lib.LT_TSDF_MERGE, sp) == 0
ms = timed(run_t, 5)
print(json.dumps({"metric": "TSDF integrate, class-aware, 2000x2000x200 voxels (4 x 3.2 GB volumes in HBM)",
                  "voxels": nvox, "ms": round(ms, 3), "Gvoxels_per_s": round(nvox / ms / 1e6, 2)}))
vol.close()
