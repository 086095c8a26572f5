"""Marching cubes (fusion_lidar.py:403-424): the generated case table and the CPU oracle.

PINNED: the oracle returns the arrays of the reference's own get_mesh with the real scikit-image 0.18.3 (golden F10 and the
seeded fuzz fixture F10b, both made with /opt/conda/bin/python3.9).  Besides: the oracle's meshes are closed, consistently
oriented surfaces with their vertices on the iso-surface, and the attribute look-up follows fusion_lidar.py:409-423 (index
rounding, world transform, colour unfolding, uint8 wrap)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_tables_are_what_the_generator_emits():
    """lidar_transfer_amd/csrc/lt_mc_lewiner_table.h is generated DATA: Lewiner's tables decoded from scikit-image 0.18.3's LUT
    file (tools/gen_mc_lewiner.py).  Where that file exists (the build image: /opt/conda) the committed header must be what
    the generator prints today; elsewhere the header's own consistency is checked (CASES covers the 256 indices, every
    tiling row holds edge codes 0..12, the derived flat array is as long as the rows it was made from)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_mc_lewiner as g
    committed = open(os.path.join(ROOT, "lidar_transfer_amd", "csrc", "lt_mc_lewiner_table.h")).read()
    if os.path.exists(g.DEFAULT):
        assert g.render(g.load(g.DEFAULT), g.DEFAULT) == committed
    import re
    m = re.search(r"LT_LWF\[(\d+)\]", committed)
    lens = {n: int(l) for n, l in re.findall(r"#define LT_LWF_(TILING\w+?)_LEN (\d+)", committed)}
    rows = {n: int(np.prod([int(d) for d in re.findall(r"\[(\d+)\]", dims)][:-1]))
            for n, dims in re.findall(r"LT_LW_(TILING\w+?)((?:\[\d+\])+) =", committed)}
    assert set(lens) == set(rows) and int(m.group(1)) == sum(rows[n] * lens[n] for n in rows)
    assert "LT_LW_CASES[256][2]" in committed
    # the emission's one-word-per-triangle copy of the flat rows: code 0 | code 1 << 5 | code 2 << 10
    flat = [int(v) for v in re.search(r"LT_LWF\[\d+\] = \{(.*?)\};", committed, re.S).group(1).replace("\n", "").split(",") if v.strip()]
    packed = [int(v) for v in re.search(r"LT_LWF3\[\d+\] = \{(.*?)\};", committed, re.S).group(1).replace("\n", "").split(",") if v.strip()]
    assert len(flat) == int(m.group(1)) and len(packed) * 3 == len(flat) and max(flat) <= 31
    assert packed == [flat[i] | flat[i + 1] << 5 | flat[i + 2] << 10 for i in range(0, len(flat), 3)]


def _random_field_mesh(oracle, seed, shape=(9, 8, 7), smooth=False):
    rng = np.random.default_rng(seed)
    t = rng.normal(size=shape).astype(np.float32)
    if smooth:
        for ax in range(3):
            t = (t + np.roll(t, 1, ax) + np.roll(t, -1, ax)) / 3
    # (no exact zeros here: a zero sample puts the vertices of several lattice edges on one grid point -- distinct indices at
    # one position, which the edge-matching of the watertightness test cannot see through; zeros are in F10 / F10b)
    col = (rng.integers(0, 260, shape) * 65536).astype(np.float32)
    rem = rng.random(shape).astype(np.float32)
    return t, col, rem, oracle.marching_cubes(t, col, rem, 0.05, np.array([-1.0, 2.0, 0.5], np.float32))


def _edge_use(faces):
    e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]], 0)
    return e


@pytest.mark.parametrize("seed", range(6))
def test_oracle_mesh_is_watertight_on_random_fields(oracle, seed):
    """White noise hits all 256 sign patterns and Lewiner's sub-cases: each directed edge must be matched by exactly one
    opposite directed edge (closed, consistently oriented surface), except on the volume boundary."""
    shape = (9, 8, 7)
    t, col, rem, (v, f, c, r) = _random_field_mesh(oracle, seed, shape)
    assert f.shape[0] > 100 and f.min() >= 0 and f.max() < v.shape[0]
    assert len(np.unique(f)) == v.shape[0]                      # no orphan vertices
    e = _edge_use(f)
    fwd = {}
    for a, b in e:
        fwd[(a, b)] = fwd.get((a, b), 0) + 1
    assert max(fwd.values()) == 1                               # no directed edge twice: consistent orientation
    vox = (v - np.array([-1.0, 2.0, 0.5], np.float32)) / np.float32(0.05)
    on_boundary = np.zeros(v.shape[0], bool)
    for k in range(3):
        on_boundary |= (np.abs(vox[:, k]) < 1e-3) | (np.abs(vox[:, k] - (shape[k] - 1)) < 1e-3)
    for (a, b) in fwd:
        if (b, a) not in fwd:
            assert on_boundary[a] and on_boundary[b], f"open edge {a}-{b} inside the volume"


def test_sphere_is_a_closed_surface_of_genus_zero_on_the_iso_level(oracle):
    n = 24
    g = np.arange(n, dtype=np.float64) - (n - 1) / 2 + 0.123
    x, y, z = np.meshgrid(g, g + 0.2, g - 0.31, indexing="ij")
    t = (np.sqrt(x * x + y * y + z * z) - 8.0).astype(np.float32) / 5          # < 0 inside
    v, f, c, r = oracle.marching_cubes(t, np.zeros_like(t), np.zeros_like(t), 1.0, np.zeros(3, np.float32))
    e = _edge_use(f)
    und = np.sort(e, 1)
    uniq, cnt = np.unique(und, axis=0, return_counts=True)
    assert (cnt == 2).all()                                                     # closed
    assert v.shape[0] - uniq.shape[0] + f.shape[0] == 2                         # Euler characteristic of a sphere
    # vertices: linear interpolation of the field along the lattice edge -> the trilinear field is ~0 there
    p = v.astype(np.float64)
    lo = np.floor(p).astype(int)
    frac = p - lo
    on_axis = np.argmax(frac, 1)
    idx = np.arange(len(p))
    hi = lo.copy()
    hi[idx, on_axis] = np.minimum(hi[idx, on_axis] + 1, n - 1)
    t0, t1 = t[lo[:, 0], lo[:, 1], lo[:, 2]], t[hi[:, 0], hi[:, 1], hi[:, 2]]
    w = frac[idx, on_axis]
    assert np.abs(t0 * (1 - w) + t1 * w).max() < 1e-6
    # normals point to the positive side (outwards)
    tri = p[f]
    nrm = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    cen = tri.mean(1) - np.array([(n - 1) / 2 - 0.123, (n - 1) / 2 - 0.323, (n - 1) / 2 + 0.31])
    assert (np.einsum("ij,ij->i", nrm, cen) > 0).mean() > 0.999
    area = 0.5 * np.linalg.norm(nrm, axis=1).sum()
    assert abs(area / (4 * np.pi * 64) - 1) < 0.03


def test_vertex_attributes_follow_get_mesh(oracle):
    """fusion_lidar.py:409-423 restated with numpy on the oracle's own voxel-space vertices."""
    t, col, rem, (v, f, c, r) = _random_field_mesh(oracle, 11, (7, 9, 6), smooth=True)
    org, vs = np.array([-1.0, 2.0, 0.5], np.float32), 0.05
    # the voxel-space vertices, recomputed with voxel_size 1 and origin 0 (exact)
    v1, f1, _, _ = oracle.marching_cubes(t, col, rem, 1.0, np.zeros(3, np.float32))
    assert np.array_equal(f, f1)
    verts_ind = np.round(v1).astype(int)
    world = v1 * vs + org
    assert world.dtype == np.float32 and np.array_equal(world.view(np.int32), v.view(np.int32))
    rgb_vals = col[verts_ind[:, 0], verts_ind[:, 1], verts_ind[:, 2]]
    colors_b = np.floor(rgb_vals / (256 * 256))
    colors_g = np.floor((rgb_vals - colors_b * 256 * 256) / 256)
    colors_r = rgb_vals - colors_b * 256 * 256 - colors_g * 256
    colors = np.floor(np.asarray([colors_r, colors_g, colors_b])).T.astype(np.int64).astype(np.uint8)
    assert np.array_equal(c, colors.astype(np.int32))
    assert (c[:, 2] == (rgb_vals / 65536).astype(np.int64) % 256).all() and c[:, 2].max() <= 255
    assert np.array_equal(r, rem[verts_ind[:, 0], verts_ind[:, 1], verts_ind[:, 2]])


def test_degenerate_volumes(oracle):
    one = np.ones((4, 5, 6), np.float32)
    v, f, c, r = oracle.marching_cubes(one, one, one, 0.1, np.zeros(3, np.float32))
    assert v.shape == (0, 3) and f.shape == (0, 3)
    thin = -np.ones((1, 5, 6), np.float32)          # no cell at all
    thin[0, 2, 3] = 1.0
    v, f, c, r = oracle.marching_cubes(thin, thin, thin, 0.1, np.zeros(3, np.float32))
    assert f.shape[0] == 0


# ---- row f2 pinned: the reference's own get_mesh (scikit-image 0.18.3's Lewiner marching cubes) -----------------------------
@pytest.mark.parametrize("case", ["blob", "noise", "street"])
def test_lewiner_oracle_returns_the_arrays_of_the_reference_get_mesh(case):
    """Golden F10 (tests/golden/make_golden_mc.py, run with /opt/conda/bin/python3.9: the reference's `TSDFVolume.get_mesh`
    with the REAL scikit-image 0.18.3) holds `get_mesh`'s return values as they are.  The C restatement of Lewiner's
    algorithm as scikit-image runs it (oracle/lt_mc_oracle.c) must return the SAME ARRAYS: vertices bit for bit and in the
    same order, faces with the same indices in the same order, colours, remissions -- smooth surfaces (blob), a street
    scene with label boundaries (street) and dense noise with exact zeros, where every ambiguous sub-case and the centre
    vertex occur (noise)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import pin_cases
    from oracle import binding as ob
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"f10_mc_{case}.npz"))
    assert str(g["skimage_version"]).startswith("0.18") and not bool(g["used_alias"])
    tsdf, color, rem, vs, org = pin_cases.mc_case(case)
    v, f, c, r = ob.marching_cubes(tsdf, color, rem, vs, org)[:4]
    assert v.shape == g["verts"].shape and f.shape == g["faces"].shape
    assert np.array_equal(v.view(np.int32), g["verts"].view(np.int32))
    assert np.array_equal(f, g["faces"])
    assert np.array_equal(np.asarray(c).astype(np.uint8), g["colors"])
    assert np.array_equal(np.asarray(r, np.float32).view(np.int32), g["vrem"].view(np.int32))
    assert f.shape[0] > 10000


def test_lewiner_oracle_equals_scikit_image_on_the_seeded_fuzz_volumes():
    """Golden F10b (tests/golden/make_golden_mc_fuzz.py): 60 small volumes -- dense noise, smooth fields, exact zeros, values
    on a coarse grid (exact ties of the asymptotic decider and of the interior test), clipped TSDF-like fields -- with the
    SHA-256 of the `verts` and `faces` arrays the real scikit-image 0.18.3 returned for them.  The C restatement must
    produce the same bytes: values AND order."""
    import hashlib
    from oracle import binding as ob
    g = np.load(os.path.join(ROOT, "tests", "golden", "f10b_lewiner_fuzz.npz"))
    assert str(g["skimage_version"]).startswith("0.18")
    off = 0
    zero3 = np.zeros(3, np.float32)
    faces_total = 0
    for k, shape in enumerate(g["shapes"]):
        n = int(np.prod(shape))
        vol = g["volumes"][off:off + n].reshape(tuple(int(s) for s in shape))
        off += n
        z = np.zeros_like(vol)
        v, f, _, _ = ob.marching_cubes(vol, z, z, 1.0, zero3)
        assert (len(v), len(f)) == (int(g["n_verts"][k]), int(g["n_faces"][k])), (k, tuple(shape))
        assert hashlib.sha256(np.ascontiguousarray(v, np.float32).tobytes()).hexdigest() == str(g["sha_verts"][k]), (k, "verts")
        assert hashlib.sha256(np.ascontiguousarray(f, np.int32).tobytes()).hexdigest() == str(g["sha_faces"][k]), (k, "faces")
        faces_total += len(f)
    assert faces_total > 20000
