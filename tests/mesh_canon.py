"""Two indexed meshes as SETS: the device path numbers vertices and faces in its own order (owner voxel / cell order),
scikit-image -- and the CPU oracle that reproduces its arrays -- by first use in a serial face stream.  Equal means: the
same multiset of vertices (position bits, colour, remission bits) and the same multiset of faces as ORDERED triples of such
vertices (a face's vertex order is part of the contract: the ray cast's arithmetic depends on it)."""
import numpy as np


def canon(v, f, c, r):
    v = np.ascontiguousarray(v, np.float32).reshape(-1, 3)
    rec = np.concatenate([v.view(np.int32), np.asarray(c, np.int64).reshape(-1, 3).astype(np.int32),
                          np.ascontiguousarray(r, np.float32).reshape(-1, 1).view(np.int32)], axis=1)
    uniq, inverse, counts = np.unique(rec, axis=0, return_inverse=True, return_counts=True)
    inverse = np.asarray(inverse).reshape(-1)
    f = np.asarray(f, np.int64).reshape(-1, 3)
    fr = inverse[f] if len(f) else np.zeros((0, 3), np.int64)
    order = np.lexsort((fr[:, 2], fr[:, 1], fr[:, 0])) if len(fr) else np.zeros(0, np.int64)
    return uniq, counts, fr[order]


def assert_same_mesh(got, want, what=""):
    gu, gc, gf = canon(*got)
    wu, wc, wf = canon(*want)
    assert np.asarray(got[0]).reshape(-1, 3).shape == np.asarray(want[0]).reshape(-1, 3).shape, \
        f"{what}vertex count {np.asarray(got[0]).shape} vs {np.asarray(want[0]).shape}"
    assert np.asarray(got[1]).reshape(-1, 3).shape == np.asarray(want[1]).reshape(-1, 3).shape, \
        f"{what}face count {np.asarray(got[1]).shape} vs {np.asarray(want[1]).shape}"
    assert gu.shape == wu.shape and np.array_equal(gu, wu), f"{what}vertex sets differ"
    assert np.array_equal(gc, wc), f"{what}vertex multiplicities differ"
    assert np.array_equal(gf, wf), f"{what}face sets differ ({int((gf != wf).any(1).sum())} rows)"
