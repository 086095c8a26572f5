"""Two indexed meshes up to the NUMBERING of their vertices: the device path numbers vertices by (word of 64 voxels, owner
voxel, edge axis; then the word's centre vertices), scikit-image -- and the CPU oracle that reproduces its arrays -- by first
use in its serial face stream.  Equal means: the same multiset of vertices (position bits, colour, remission bits) and the
same FACE STREAM -- face k of the one mesh has the same three vertices, in the same order, as face k of the other (cells in
(a0, a1, a2) order, a tiling's triangles in table order: the device emits in scikit-image's order; a face's vertex order and
the order of the faces matter to the ray cast: its arithmetic, and which of two faces reports an exact-t tie)."""
import numpy as np


def canon(v, f, c, r):
    v = np.ascontiguousarray(v, np.float32).reshape(-1, 3)
    rec = np.concatenate([v.view(np.int32), np.asarray(c, np.int64).reshape(-1, 3).astype(np.int32),
                          np.ascontiguousarray(r, np.float32).reshape(-1, 1).view(np.int32)], axis=1)
    uniq, inverse, counts = np.unique(rec, axis=0, return_inverse=True, return_counts=True)
    inverse = np.asarray(inverse).reshape(-1)
    f = np.asarray(f, np.int64).reshape(-1, 3)
    fr = inverse[f] if len(f) else np.zeros((0, 3), np.int64)
    return uniq, counts, fr


def assert_same_mesh(got, want, what=""):
    gu, gc, gf = canon(*got)
    wu, wc, wf = canon(*want)
    assert np.asarray(got[0]).reshape(-1, 3).shape == np.asarray(want[0]).reshape(-1, 3).shape, \
        f"{what}vertex count {np.asarray(got[0]).shape} vs {np.asarray(want[0]).shape}"
    assert np.asarray(got[1]).reshape(-1, 3).shape == np.asarray(want[1]).reshape(-1, 3).shape, \
        f"{what}face count {np.asarray(got[1]).shape} vs {np.asarray(want[1]).shape}"
    assert gu.shape == wu.shape and np.array_equal(gu, wu), f"{what}vertex sets differ"
    assert np.array_equal(gc, wc), f"{what}vertex multiplicities differ"
    if not np.array_equal(gf, wf):
        so = lambda a: a[np.lexsort((a[:, 2], a[:, 1], a[:, 0]))]  # noqa: E731
        same_set = np.array_equal(so(gf), so(wf))
        raise AssertionError(f"{what}face streams differ in {int((gf != wf).any(1).sum())} rows (as sets: {'equal' if same_set else 'different'})")
