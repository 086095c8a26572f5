"""GPU parity of the post-render kernels (reverse projection, scan packing = write(), compare()) against golden
vectors produced by the reference's own Python (tests/golden/make_golden.py, F7)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(GOLD, "f7_post.npz"))


@pytest.mark.parametrize("pf", [False, True])
def test_reverse_projection_vs_reference_python(g, pf):
    from lidar_transfer_amd.post import do_reverse_projection_new
    px = g["proj_x_float"] if pf else g["proj_x"]
    py = g["proj_y_float"] if pf else g["proj_y"]
    got = do_reverse_projection_new(g["range_image"], px, py, float(g["fov_up"]), float(g["fov_down"]), preserve_float=pf)
    ref = g[f"back_points_{'float' if pf else 'int'}"]
    assert got.shape == ref.shape and got.dtype == np.float64
    # float64 sin/cos of two math libraries: allow a few ulp of the double; the packed float32 must be identical
    assert np.allclose(got, ref, rtol=1e-13, atol=1e-13)
    assert np.array_equal(got.astype(np.float32), ref.astype(np.float32))


def test_write_cp_adaption_bytes(g, tmp_path):
    """`.bin` / `.label` of the closest-point adaption: byte-identical files (laserscan.py:1133-1178)."""
    from lidar_transfer_amd.post import do_reverse_projection_new, write_scan
    back = do_reverse_projection_new(g["range_image"], g["proj_x"], g["proj_y"], float(g["fov_up"]),
                                     float(g["fov_down"]))
    # the reference wrote the files from the preserve_float=True back-points (last call in make_golden)
    back = g["back_points_float"]
    n = write_scan(str(tmp_path), 7, back, g["label_image"], g["proj_remissions"], index=g["index"])
    b = np.fromfile(tmp_path / "velodyne" / "000007.bin", np.uint8)
    l = np.fromfile(tmp_path / "labels" / "000007.label", np.uint8)
    assert n * 16 == b.size and n * 4 == l.size
    assert np.array_equal(b, g["cp_bin_bytes"]) and np.array_equal(l, g["cp_label_bytes"])


def test_write_mesh_adaption_bytes(tmp_path):
    """Mesh adaption: pack the images of a raytraced scan (golden F4) exactly as the reference's write()."""
    from lidar_transfer_amd.post import pack_scan
    g7 = np.load(os.path.join(GOLD, "f7_post.npz"))
    g4 = np.load(os.path.join(GOLD, "f4_50k_64x256.npz"))
    b, l = pack_scan(g4["endpoints"], g4["label"], g4["endrem"])
    assert b.dtype == np.float32 and l.dtype == np.uint32 and b.shape[0] == l.shape[0] == int(g4["n_hits"])
    assert np.array_equal(b.view(np.uint8).reshape(-1), g7["mesh_bin_bytes"])
    assert np.array_equal(l.view(np.uint8).reshape(-1), g7["mesh_label_bytes"])


def test_pack_scan_edge_cases():
    from lidar_transfer_amd.post import pack_scan
    pts = np.array([[1, 2, 3], [0, 0, 0], [1, -1, 0], [4, 5, 6], [7, 8, 9]], np.float64)
    lab = np.array([10, 40, 40, -1, 50], np.int32)
    rem = np.arange(5, dtype=np.float32)
    b, l = pack_scan(pts, lab, rem)          # (0,0,0) dropped; x+y+z == 0 dropped (sic); label -1 dropped
    assert np.array_equal(b, np.array([[1, 2, 3, 0], [7, 8, 9, 4]], np.float32)) and list(l) == [10, 50]
    b, l = pack_scan(pts, lab, rem, index=np.array([0, 3, 3, 3, 9], np.int32))   # 'cp': index > 0 only
    assert list(l) == [50]
    b, l = pack_scan(np.zeros((0, 3)), np.zeros(0, np.int32), np.zeros(0, np.float32))
    assert b.shape == (0, 4) and l.shape == (0,)


def test_compare_vs_reference_python(g):
    from lidar_transfer_amd.post import compare
    r = compare(g["cmp_source_label"], g["cmp_source_color"], g["cmp_target_label"], g["cmp_source_range"],
                g["cmp_target_range"], g["cmp_source_rem"], g["cmp_target_rem"], nclasses=20)
    assert np.array_equal(r["range_diff"].view(np.int32), g["cmp_range_diff"].view(np.int32))
    assert np.array_equal(r["rem_diff"].view(np.int32), g["cmp_rem_diff"].view(np.int32))
    assert abs(r["m_iou"] - float(g["cmp_m_iou"])) < 1e-12
    assert abs(r["m_acc"] - float(g["cmp_m_acc"])) < 1e-12
    # numpy sums the float32 image pairwise in float32; the kernel accumulates in float64
    assert abs(r["MSE"] - float(g["cmp_mse"])) < 1e-6 * max(float(g["cmp_mse"]), 1e-12) + 1e-9


@pytest.mark.parametrize("seed", range(8))
def test_pack_scan_fuzz_vs_numpy_filter(seed):
    """Stable compaction at image scale (up to 128 x 2048 cells) against the reference's filter chain restated
    in numpy (laserscan.py:1138-1154): `index > 0` (cp adaption only), `label >= 0`, `sum(xyz) != 0` -- evaluated
    in the dtype of the points, so a float32 and a float64 image can keep different points."""
    from lidar_transfer_amd.post import pack_scan
    rng = np.random.default_rng(seed)
    n = int(rng.choice([1, 63, 4096, 131072, 262144]))
    dtype = np.float32 if seed % 2 else np.float64
    pts = rng.normal(size=(n, 3)).astype(dtype) * 30
    pts[rng.random(n) < 0.3] = 0                                   # misses
    k = rng.random(n) < 0.05
    pts[k, 2] = -(pts[k, 0] + pts[k, 1])                           # x + y + z == 0 without being (0, 0, 0) (sic)
    lab = rng.integers(-1, 260, n).astype(np.int32)
    rem = rng.uniform(0, 1, n).astype(np.float32)
    index = rng.integers(-1, 50, n).astype(np.int32) if seed % 3 == 0 else None
    b, l = pack_scan(pts, lab, rem, index=index)
    keep = np.ones(n, bool) if index is None else index > 0
    keep &= lab >= 0
    keep &= np.sum(pts, axis=1) != 0
    exp_b = np.concatenate([pts[keep].astype(np.float32), rem[keep][:, None]], axis=1)
    assert b.shape == exp_b.shape and np.array_equal(b.view(np.int32), exp_b.view(np.int32))
    assert np.array_equal(l, lab[keep].astype(np.uint32))


@pytest.mark.parametrize("seed", range(6))
def test_compare_fuzz_incl_negative_labels_vs_oracle(seed):
    """Random label images -- every second seed with negative labels, which the reference treats as the lowest
    classes (labels are renumbered by rank before they index the confusion matrix, laserscan.py:1216-1222) --
    against the numpy restatement pinned to the reference (oracle/compare.py)."""
    from lidar_transfer_amd.post import compare
    from oracle.compare import compare as ocompare
    rng = np.random.default_rng(100 + seed)
    H, W = 16, 256
    vals = np.array([0, 1, 10, 40, 48, 50, 70, 259] + ([-1, -7] if seed % 2 else []), np.int32)
    sl = rng.choice(vals, (H, W)).astype(np.int32)
    tl = np.where(rng.random((H, W)) < 0.8, sl, rng.choice(vals, (H, W))).astype(np.int32)
    sc = rng.random((H, W, 3))
    sc[rng.random((H, W)) < 0.1] = 0
    sr, tr = rng.uniform(0, 80, (H, W)).astype(np.float32), rng.uniform(0, 80, (H, W)).astype(np.float32)
    sm, tm = rng.random((H, W)).astype(np.float32), rng.random((H, W)).astype(np.float32)
    got = compare(sl, sc, tl, sr, tr, sm, tm, nclasses=20)
    want = ocompare(sl, sc, tl, sr, tr, sm, tm, nclasses=20)
    assert np.array_equal(got["source_label"], want["source_label"])
    assert np.array_equal(got["target_label"], want["target_label"])
    assert np.array_equal(got["range_diff"].view(np.int32), want["range_diff"].view(np.int32))
    assert np.array_equal(got["rem_diff"].view(np.int32), want["rem_diff"].view(np.int32))
    assert abs(got["m_iou"] - want["m_iou"]) < 1e-12 and abs(got["m_acc"] - want["m_acc"]) < 1e-12
    assert np.allclose(got["iou"], want["iou"], atol=1e-12)


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_compare_negative_labels_vs_reference_golden(case):
    """Golden F7b: the reference's own compare() on label sets with negative values, where its in-place renumbering
    merges classes (laserscan.py:1216-1222) -- the device histogram + host replay must give its mIoU / accuracy."""
    from lidar_transfer_amd.post import compare
    g = np.load(os.path.join(GOLD, "f7b_compare_negative.npz"))
    r = compare(g[f"{case}_source_label"], g[f"{case}_source_color"], g[f"{case}_target_label"],
                g[f"{case}_source_range"], g[f"{case}_target_range"], g[f"{case}_source_rem"], g[f"{case}_target_rem"],
                nclasses=20)
    assert np.array_equal(r["range_diff"].view(np.int32), g[f"{case}_range_diff"].view(np.int32))
    assert np.array_equal(r["rem_diff"].view(np.int32), g[f"{case}_rem_diff"].view(np.int32))
    assert abs(r["m_iou"] - float(g[f"{case}_m_iou"])) < 1e-12 and abs(r["m_acc"] - float(g[f"{case}_m_acc"])) < 1e-12
