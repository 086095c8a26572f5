#!/usr/bin/env python3
"""SHA-256 of the LBVH's 4-wide node array (Scene.build) on seeded meshes: the pin a change of the sort / hierarchy kernels
must leave untouched (tests/test_trace_gpu.py::test_lbvh_node_array_is_pinned holds the digests).  Nodes never written keep
the fill pattern, so the digest also pins WHICH nodes exist."""
import ctypes as C, hashlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CASES = ((0, 300), (1, 5000), (0, 20000), (2, 200000), (0, 1000000))


def digests():
    import torch
    from lidar_transfer_amd import _lib
    from lidar_transfer_amd.raytracer import Scene
    from lidar_transfer_amd.synth import synth_scene
    lib = _lib.load()
    out = {}
    for seed, tris in CASES:
        v, f, c, r = synth_scene(seed, tris)
        sc = Scene(0)
        sc.set_mesh(*[torch.from_numpy(x).cuda() for x in (v, f, c, r)])
        sc.build()
        torch.cuda.synchronize()
        assert lib.lt_debug_nodes4_fill(sc._h, 0xAB) == 0
        sc.build()
        torch.cuda.synchronize()
        n = f.shape[0]
        a = np.empty((n, 32), np.uint32)
        assert lib.lt_debug_nodes4_get(sc._h, a.ctypes.data_as(C.c_void_p), n) == 0
        sc.status()
        sc.close()
        out[f"s{seed}_{tris}"] = hashlib.sha256(a.tobytes()).hexdigest()
    return out


if __name__ == "__main__":
    print(json.dumps(digests()))
