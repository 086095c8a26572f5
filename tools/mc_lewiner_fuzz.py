#!/opt/conda/bin/python3.9
"""tools/mc_lewiner_fuzz.py -- scikit-image's `marching_cubes_lewiner` (the REAL dependency of the reference,
auxiliary/fusion_lidar.py:407) against oracle/lt_mc_oracle.c on random volumes, OUTPUT ARRAYS compared exactly: vertices
(values and order) and faces (values and order).  Needs an interpreter with scikit-image 0.18.x -- in the build image:

    /opt/conda/bin/python3.9 tools/mc_lewiner_fuzz.py [n_volumes] [seed]

Volume kinds: dense uniform noise (every one of the 256 sign patterns and of Lewiner's sub-cases occurs thousands of
times), smooth fields (few ambiguous cells), fields with exact zeros, values on a coarse grid (ties in the asymptotic
decider), TSDF-like clipped fields.  numpy only besides scikit-image; the C library is loaded with ctypes."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)
_lib = None


def _oracle():
    global _lib
    if _lib is None:
        _lib = C.CDLL(os.path.join(ROOT, "oracle", "liblt_oracle.so"))
        _lib.lto_marching_cubes.argtypes = [fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_float, fp, fp, ip, ip, fp, C.c_int, C.c_int, ip, ip]
        _lib.lto_marching_cubes.restype = C.c_int
    return _lib


def ours(vol):
    lib = _oracle()
    vol = np.ascontiguousarray(vol, np.float32)
    zeros = np.zeros_like(vol)
    org = np.zeros(3, np.float32)
    nv, nf = C.c_int(0), C.c_int(0)
    f = lambda a: a.ctypes.data_as(fp)  # noqa: E731
    lib.lto_marching_cubes(f(vol), f(zeros), f(zeros), *vol.shape, 1.0, f(org), None, None, None, None, 0, 0, C.byref(nv), C.byref(nf))
    v = np.zeros((max(nv.value, 1), 3), np.float32)
    fa = np.zeros((max(nf.value, 1), 3), np.int32)
    col = np.zeros((max(nv.value, 1), 3), np.int32)
    rem = np.zeros(max(nv.value, 1), np.float32)
    lib.lto_marching_cubes(f(vol), f(zeros), f(zeros), *vol.shape, 1.0, f(org), f(v), fa.ctypes.data_as(ip), col.ctypes.data_as(ip),
                           f(rem), nv.value, nf.value, C.byref(nv), C.byref(nf))
    return v[:nv.value], fa[:nf.value]


def volume(rng, kind, shape):
    if kind == 0:
        return rng.uniform(-1, 1, shape).astype(np.float32)
    if kind == 1:
        g = np.stack(np.meshgrid(*[np.linspace(-1, 1, s) for s in shape], indexing="ij"))
        c = rng.uniform(-0.5, 0.5, 3)[:, None, None, None]
        return (np.sqrt(((g - c) ** 2).sum(0)) - rng.uniform(0.3, 0.9) + 0.05 * rng.standard_normal(shape)).astype(np.float32)
    if kind == 2:
        v = rng.uniform(-1, 1, shape).astype(np.float32)
        v[rng.random(shape) < 0.15] = 0.0
        return v
    if kind == 3:
        return (rng.integers(-3, 4, shape) / 4.0).astype(np.float32) + np.float32(0.125) * (rng.random(shape) < 0.5)
    v = np.clip(rng.standard_normal(shape) * 0.7, -1, 1).astype(np.float32)
    v[rng.random(shape) < 0.3] = 1.0
    return v


def main():
    from skimage import measure   # (only here: `volume` is imported by tests/test_mc_gpu.py under the system interpreter)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    bad = 0
    cells = 0
    for k in range(n):
        kind = k % 5
        shape = tuple(int(s) for s in rng.integers(2, 14, 3))
        vol = volume(rng, kind, shape)
        if not (vol.min() <= 0 <= vol.max()):
            continue
        try:
            v, f, _, _ = measure.marching_cubes_lewiner(vol, level=0)
        except RuntimeError:      # "No surface found"
            v, f = np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32)
        ov, of = ours(vol)
        cells += (shape[0] - 1) * (shape[1] - 1) * (shape[2] - 1)
        ok = v.shape == ov.shape and f.shape == of.shape and np.array_equal(np.asarray(v, np.float32).view(np.int32), ov.view(np.int32)) \
            and np.array_equal(np.asarray(f, np.int32), of)
        if not ok:
            bad += 1
            if bad <= 5:
                print(f"volume {k} kind {kind} shape {shape}: skimage {v.shape} {f.shape} ours {ov.shape} {of.shape}")
                if v.shape == ov.shape:
                    d = np.argwhere((np.asarray(v, np.float32).view(np.int32) != ov.view(np.int32)).any(1))
                    print("  first differing vertex", d[:3].ravel(), v[d[0, 0]] if len(d) else None, ov[d[0, 0]] if len(d) else None)
                if f.shape == of.shape:
                    d = np.argwhere((np.asarray(f) != of).any(1))
                    print("  first differing face", d[:3].ravel(), f[d[0, 0]] if len(d) else None, of[d[0, 0]] if len(d) else None)
    print(f"{n} volumes, {cells} cells: {bad} differ")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
