#!/usr/bin/env python3
"""tests/stress_mc.py [n] [seed] -- marching cubes on the device against the CPU oracle (which returns scikit-image 0.18.3's
arrays: goldens F10 / F10b, tools/mc_lewiner_fuzz.py) on random volumes of the five fuzz kinds and random shapes (nz on both
sides of the 64-bit word boundaries), through ONE mesh object (buffers grown, shrunk, the side table emptied in between):
the same vertices and the same face stream, and after lt_mesh_renumber_dev the same arrays element for element."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))  # (mc_lewiner_fuzz.volume)


def main(argv=None):
    argv = sys.argv if argv is None else argv
    import torch
    from mc_lewiner_fuzz import volume
    from mesh_canon import assert_same_mesh
    from lidar_transfer_amd.fusion import DeviceMesh
    from oracle import binding as ob
    n = int(argv[1]) if len(argv) > 1 else 200
    rng = np.random.default_rng(int(argv[2]) if len(argv) > 2 else 0)
    dev = torch.device("cuda", 0)
    m = DeviceMesh(0)
    faces = cells = bad = 0
    t0 = time.time()
    for k in range(n):
        kind = k % 5
        shape = (int(rng.integers(2, 40)), int(rng.integers(2, 40)), int(rng.choice([2, 7, 63, 64, 65, 100, 128, 129, 200, int(rng.integers(2, 260))])))
        t = np.ascontiguousarray(volume(rng, kind, shape), np.float32)
        col = (rng.integers(0, 260, shape) * 65536 + rng.integers(0, 256, shape)).astype(np.float32)
        rem = rng.random(shape).astype(np.float32)
        vs = float(rng.choice([0.05, 0.1, 0.25]))
        org = rng.uniform(-5, 5, 3).astype(np.float32)
        want = ob.marching_cubes(t, col, rem, vs, org)
        m.extract(*[torch.from_numpy(a).to(dev) for a in (t, col, rem)], vs, [float(x) for x in org])
        got = [a.cpu().numpy() for a in m.tensors()]
        try:
            assert_same_mesh(got, want)
            v, f, c, r = [a.cpu().numpy() for a in m.renumber().tensors()]
            assert np.array_equal(v.view(np.int32), want[0].view(np.int32)) and np.array_equal(f, want[1])
            assert np.array_equal(c, np.asarray(want[2])) and np.array_equal(r.view(np.int32), want[3].view(np.int32))
        except AssertionError as e:
            bad += 1
            print(f"case {k} kind {kind} shape {shape}: {str(e)[:120]}")
        faces += want[1].shape[0]
        cells += (shape[0] - 1) * (shape[1] - 1) * (shape[2] - 1)
    m.close()
    print(f"{n} volumes, {cells} cells, {faces} faces: {bad} mismatches ({time.time() - t0:.0f} s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
