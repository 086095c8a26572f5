#!/opt/conda/bin/python3.9
"""Generate golden fixture F10b -- scikit-image 0.18.3's `measure.marching_cubes_lewiner(vol, level=0)` (the call of
auxiliary/fusion_lidar.py:407) on 60 small seeded volumes that reach every one of Lewiner's sub-cases, the centre vertex,
exact zeros and exact ties of the face / interior tests:

    /opt/conda/bin/python3.9 tests/golden/make_golden_mc_fuzz.py      # needs scikit-image 0.18.x
    git add tests/golden/f10b_lewiner_fuzz.npz

The fixture holds the INPUT volumes (float32, concatenated) with their shapes and, per volume, the number of vertices and
faces scikit-image returned and the SHA-256 of its `verts` (float32) and `faces` (int32) arrays as they are -- values and
order.  tests/test_mc_cpu.py runs the C restatement (oracle/lt_mc_oracle.c) on the same volumes and compares the digests.
Volume kinds: tools/mc_lewiner_fuzz.py (`volume`).  Only data is written."""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    import skimage
    from skimage import measure
    import mc_lewiner_fuzz as fz
    rng = np.random.default_rng(2024)
    vols, shapes, nv, nf, hv, hf = [], [], [], [], [], []
    k = 0
    while len(shapes) < 60:
        kind = k % 5
        k += 1
        shape = tuple(int(s) for s in rng.integers(2, 11, 3))
        vol = np.ascontiguousarray(fz.volume(rng, kind, shape), np.float32)
        if not (vol.min() <= 0 <= vol.max()):
            continue
        try:
            v, f, _, _ = measure.marching_cubes_lewiner(vol, level=0)
        except RuntimeError:      # "No surface found at the given iso value."
            v, f = np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32)
        v, f = np.ascontiguousarray(v, np.float32), np.ascontiguousarray(f, np.int32)
        vols.append(vol.ravel()); shapes.append(shape); nv.append(len(v)); nf.append(len(f))
        hv.append(hashlib.sha256(v.tobytes()).hexdigest()); hf.append(hashlib.sha256(f.tobytes()).hexdigest())
    np.savez_compressed(os.path.join(HERE, "f10b_lewiner_fuzz.npz"), volumes=np.concatenate(vols), shapes=np.array(shapes, np.int32),
                        n_verts=np.array(nv, np.int32), n_faces=np.array(nf, np.int32), sha_verts=np.array(hv), sha_faces=np.array(hf),
                        skimage_version=np.array(skimage.__version__))
    print(len(shapes), "volumes,", sum(nf), "faces, skimage", skimage.__version__)


if __name__ == "__main__":
    main()
