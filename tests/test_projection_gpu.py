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


@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("method", ["old", "new"])
def test_projection_at_baseline_scale_vs_reference(tag, method):
    """120 000 points -> 64 x 2048 (BASELINE.json's projection shape) against the reference's own arrays
    (tests/golden/make_golden.py section F9).

    float64 clouds: every output of `do_range_projection` / `do_range_projection_new` +
    `do_label_projection_new` by SHA-256 (dtype and shape included).  float32 clouds -- what `open_scan` reads from a
    .bin file: numpy's float32 arcsin / arctan2 are not correctly rounded and differ from every other libm (and
    between numpy builds) in the last bit, which moves about one point in 1e5 across a pixel border; there the
    per-point outputs must be identical and the images may differ in a handful of cells."""
    import hashlib
    from lidar_transfer_amd.laserscan import SemLaserScan
    from lidar_transfer_amd.synth import synth_cloud
    g = np.load(os.path.join(GOLD, "f9_range_projection_full.npz"))
    H, W, fu, fd = int(g["H"]), int(g["W"]), float(g["fov_up"]), float(g["fov_down"])
    dtype = np.float32 if tag == "f32" else np.float64
    key = f"{tag}_{method}"
    pts, rem_p, lab = synth_cloud(int(g["seed"]), int(g["n_points"]), dtype=dtype, fov_up=fu, fov_down=fd)
    if method == "new":
        pts[1000:1100] = pts[5000:5100]  # exact depth ties: only the _new loop has a defined rule for them
    pts[7] = 0
    assert hashlib.sha256(pts.tobytes()).digest() == bytes(g[f"{key}_points_sha256"]), "synthetic cloud drifted"
    scan = SemLaserScan(H, W, 300, COLOR_DICT, None, None)
    scan.points, scan.remissions, scan.label = pts.copy(), rem_p.copy(), lab.copy()
    scan.colorize()
    if method == "old":
        scan.do_range_projection(fu, fd, remove=True)
        outs = dict(proj_range=scan.proj_range, proj_remissions=scan.proj_remissions, unproj_range=scan.unproj_range,
                    points_kept=scan.points, proj_idx=scan.proj_idx, proj_xyz=scan.proj_xyz, proj_mask=scan.proj_mask)
        index_name = "proj_idx"
    else:
        scan.do_range_projection_new(fu, fd, remove=True)
        scan.do_label_projection_new()
        outs = dict(proj_range=scan.proj_range, proj_remissions=scan.proj_remissions, unproj_range=scan.unproj_range,
                    points_kept=scan.points, index=scan.index, label_image=scan.label_image, proj_x=scan.proj_x,
                    proj_y=scan.proj_y, proj_label=scan.proj_label)
        index_name = "index"
    per_point = ("unproj_range", "points_kept")
    for name, arr in outs.items():
        arr = np.ascontiguousarray(np.asarray(arr))
        assert str(arr.dtype) == str(g[f"{key}_{name}_dtype"]), f"{name}: dtype {arr.dtype}"
        assert tuple(arr.shape) == tuple(g[f"{key}_{name}_shape"]), f"{name}: shape {arr.shape}"
        if tag == "f64" or name in per_point:
            assert hashlib.sha256(arr.tobytes()).digest() == bytes(g[f"{key}_{name}_sha256"]), f"{name} differs"
    if tag == "f32":
        idx, rng_img = np.asarray(outs[index_name]), np.asarray(outs["proj_range"])
        bad = (idx != g[f"{key}_image_index"]) | (rng_img.view(np.int32) != g[f"{key}_image_range"].view(np.int32))
        assert int(bad.sum()) <= 8, f"{int(bad.sum())} cells differ from the reference"
        assert abs(int((rng_img > 0).sum()) - int(g[f"{key}_filled"])) <= 4
    else:
        assert int((np.asarray(outs["proj_range"]) > 0).sum()) == int(g[f"{key}_filled"])


@pytest.mark.parametrize("seed", range(24))
def test_projection_fuzz_vs_cpu_restatement(seed):
    """Random clouds, image shapes, fields of view, beam tables and removal modes: the HIP projections against
    oracle/projection.py (itself pinned to the reference's arrays in tests/test_oracle_cpu.py).  float64 clouds must
    match exactly; float32 clouds may differ in the few cells numpy's float32 arcsin / arctan2 rounding moves."""
    from oracle import projection as op
    from lidar_transfer_amd.laserscan import SemLaserScan
    rng = np.random.default_rng(1000 + seed)
    H = int(rng.choice([1, 4, 16, 64])); W = int(rng.choice([1, 16, 128, 2048]))
    fu = float(rng.uniform(-5, 40)); fd = float(fu - rng.uniform(1, 70))
    n = int(rng.choice([0, 1, 7, 500, 20000, 60000]))
    dtype = np.float32 if seed % 3 == 0 else np.float64
    remove = bool(seed % 2 == 0)
    method = "new" if seed % 4 < 2 else "old"
    beams = list(np.deg2rad(np.sort(rng.uniform(fd, fu, H))[::-1])) if seed % 5 == 0 and H > 1 else None
    r = rng.uniform(0.5, 80.0, n); yaw = rng.uniform(-np.pi, np.pi, n)
    pitch = np.deg2rad(rng.uniform(fd - 3, fu + 3, n) if remove else rng.uniform(fd, fu, n))
    pts = np.stack([r * np.cos(pitch) * np.cos(yaw), r * np.cos(pitch) * np.sin(yaw), r * np.sin(pitch)], 1).astype(dtype)
    if n >= 500:
        pts[10:60] = pts[100:150]            # exact duplicates: depth ties in one cell
        pts[200:230] *= dtype(2.0)           # same pixel, different depth
        if remove or method == "new":
            pts[5] = 0                       # depth 0 (the old method without `remove` divides by it: not a valid input)
    rem = rng.uniform(0, 1, n).astype(np.float32)
    lab = rng.choice(list(COLOR_DICT.keys()), n).astype(np.uint32)
    scan = SemLaserScan(H, W, 300, COLOR_DICT, None, beams)
    scan.points, scan.remissions, scan.label = pts.copy(), rem.copy(), lab.copy()
    scan.colorize()
    if method == "old":
        scan.do_range_projection(fu, fd, remove=remove)
        scan.do_label_projection()
        got_index, got_range = scan.proj_idx, scan.proj_range
    else:
        scan.do_range_projection_new(fu, fd, remove=remove)
        scan.do_label_projection_new()
        got_index, got_range = scan.index, scan.range_image
    o = op.range_projection(pts, rem, H, W, fu, fd, beam_angles=beams, remove=remove, method=method)
    exp_label = op.label_projection(o["index"], lab[o["kept"]])
    if dtype == np.float64:
        assert np.array_equal(scan.points, pts[o["kept"]])
        assert np.array_equal(scan.unproj_range, o["unproj_range"])
        assert np.array_equal(got_index, o["index"]), f"{int((got_index != o['index']).sum())} cells differ"
        assert np.array_equal(np.asarray(got_range).view(np.int32), o["range"].view(np.int32))
        assert np.array_equal(scan.proj_remissions, o["remission"])
        assert np.array_equal(scan.proj_label, exp_label)
        if method == "old":
            assert np.array_equal(scan.proj_xyz, o["xyz"]) and np.array_equal(scan.proj_mask, o["mask"])
    else:
        assert abs(scan.points.shape[0] - o["kept"].shape[0]) <= 2
        if scan.points.shape[0] == o["kept"].shape[0]:
            bad = (got_index != o["index"]) | (np.asarray(got_range).view(np.int32) != o["range"].view(np.int32))
            assert int(bad.sum()) <= 6, f"{int(bad.sum())} cells differ"


def test_label_beyond_the_colour_table_raises_like_the_reference():
    """`proj_color[mask] = color_lut[label[idx[mask]]]` (laserscan.py:649, :676) raises IndexError for a label the
    look-up table does not hold; the mirror must not map it to some other class."""
    from lidar_transfer_amd.laserscan import SemLaserScan
    from lidar_transfer_amd.synth import synth_cloud
    pts, rem, lab = synth_cloud(3, 5000, dtype=np.float32)
    scan = SemLaserScan(16, 128, 300, COLOR_DICT, None, None)
    scan.points, scan.remissions, scan.label = pts, rem, lab.copy()
    scan.do_range_projection(3.0, -25.0, remove=True)
    scan.do_label_projection()                           # fine: every label is inside the table (size 80 + 1 + 100)
    assert scan.proj_color.shape == (16, 128, 3)
    bad = SemLaserScan(16, 128, 300, COLOR_DICT, None, None)
    bad.points, bad.remissions, bad.label = pts.copy(), rem.copy(), lab.copy()
    bad.label[::7] = 100000
    bad.do_range_projection(3.0, -25.0, remove=True)
    with pytest.raises(IndexError):
        bad.do_label_projection()


# ---- the batched, synchronisation-free entry (lt_range_projection_batch_dev) ---------------------------------------------
def _fuzz_cloud(rng, n, dtype, fd, fu):
    r = rng.uniform(0.5, 80.0, n); yaw = rng.uniform(-np.pi, np.pi, n)
    pitch = np.deg2rad(rng.uniform(fd - 3, fu + 3, n))
    pts = np.stack([r * np.cos(pitch) * np.cos(yaw), r * np.cos(pitch) * np.sin(yaw), r * np.sin(pitch)], 1).astype(dtype)
    if n >= 500:
        pts[10:60] = pts[100:150]
        pts[200:230] *= dtype(2.0)
        pts[5] = 0
    rem = rng.uniform(0, 1, n).astype(np.float32)
    lab = rng.choice(list(COLOR_DICT.keys()), n).astype(np.uint32)
    return pts, rem, lab


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("new", [True, False])
@pytest.mark.parametrize("remove", [True, False])
def test_batch_projection_equals_the_single_cloud_call(dtype, new, remove):
    """Twelve clouds of different sizes (0, 1, 7 ... 130 000 points: more than one group of 8) through ONE
    lt_range_projection_batch_dev call, twice in a row on one projector (the resolve pass re-arms the workspace), against
    the single-cloud call per cloud -- which the golden vectors of the reference's own Python pin (tests above): every
    image bit for bit, incl. the numbering of the kept points, the pixel coordinates of empty cells (numpy's index -1)
    and the folded label image `integrate` consumes."""
    import torch
    from lidar_transfer_amd.laserscan import Projector, SemLaserScan
    rng = np.random.default_rng(7 + 2 * int(new) + int(remove))
    H, W, fu, fd = 32, 512, 4.0, -24.0
    sizes = [0, 1, 7, 500, 20000, 60000, 130000, 3000, 64, 65, 4097, 25000]
    beams = list(np.deg2rad(np.linspace(fu, fd, H))) if (new and remove and dtype == np.float64) else None
    pj = Projector()
    outs = ("idx", "range", "xyz", "rem", "label", "color", "mask", "label_folded", "proj_x", "proj_y", "proj_xf",
            "proj_yf", "n_kept")
    for rep in range(2):
        clouds = [_fuzz_cloud(rng, n, dtype, fd, fu) for n in sizes]
        if not (new or remove):
            for pts, _, _ in clouds:       # the old variant without `remove` divides by depth 0: not a valid input
                if len(pts) > 5:
                    pts[5] = pts[6]
        ref = []
        for pts, rem, lab in clouds:
            s = SemLaserScan(H, W, 300, COLOR_DICT, None, beams)
            s.points, s.remissions, s.label = pts.copy(), rem.copy(), lab.copy()
            s.colorize()
            if new:
                s.do_range_projection_new(fu, fd, remove=remove)
                s.do_label_projection_new()
            else:
                s.do_range_projection(fu, fd, remove=remove)
                s.do_label_projection()
            ref.append(s)
        lut = torch.from_numpy(ref[0].color_lut).cuda()
        dev = [(torch.from_numpy(p).cuda(), torch.from_numpy(r).cuda(), torch.from_numpy(l.astype(np.int32)).cuda())
               for p, r, l in clouds]
        got = pj.project(dev, fu, fd, H, W, new=new, remove=remove, beam_angles=beams, color_lut=lut if new else None,
                         outputs=outs)   # (the old variant's mirror passes no colour table: its colour image is 0)
        torch.cuda.synchronize()
        for k, (s, g) in enumerate(zip(ref, got)):
            g = {n: t.cpu().numpy() for n, t in g.items()}
            o = s._last
            tag = f"rep {rep} cloud {k} (n={sizes[k]})"
            assert int(g["n_kept"][0]) == s.points.shape[0], tag
            assert np.array_equal(g["idx"], o["idx"]), tag
            assert np.array_equal(g["range"].view(np.int32), o["range"].view(np.int32)), tag
            assert np.array_equal(g["rem"].view(np.int32), o["remi"].view(np.int32)), tag
            assert np.array_equal(g["label"], o["labi"]), tag
            assert np.array_equal(g["xyz"].view(np.int32), o["xyz"].view(np.int32)), tag
            assert np.array_equal(g["color"], o["col"]), tag
            assert np.array_equal(g["mask"], o["mask"]), tag
            assert np.array_equal(g["label_folded"], np.floor(o["labi"].astype(np.float32) * 256 * 256)), tag
            if new and s.points.shape[0]:
                assert np.array_equal(g["proj_x"], s.proj_x) and np.array_equal(g["proj_y"], s.proj_y), tag
                assert np.array_equal(g["proj_xf"], s.proj_x_float) and np.array_equal(g["proj_yf"], s.proj_y_float), tag
    pj.close()


def test_batch_projection_on_the_reference_golden_at_baseline_scale():
    """120 000 float64 points -> 64 x 2048 (golden F9, made by the reference's own do_range_projection_new +
    do_label_projection_new): the images of the batched entry by SHA-256, five copies of the cloud in one call."""
    import hashlib
    import torch
    from lidar_transfer_amd.laserscan import Projector
    from lidar_transfer_amd.synth import synth_cloud
    g = np.load(os.path.join(GOLD, "f9_range_projection_full.npz"))
    H, W, fu, fd = int(g["H"]), int(g["W"]), float(g["fov_up"]), float(g["fov_down"])
    pts, rem_p, lab = synth_cloud(int(g["seed"]), int(g["n_points"]), dtype=np.float64, fov_up=fu, fov_down=fd)
    pts[1000:1100] = pts[5000:5100]
    pts[7] = 0
    assert hashlib.sha256(pts.tobytes()).digest() == bytes(g["f64_new_points_sha256"]), "synthetic cloud drifted"
    cl = (torch.from_numpy(pts).cuda(), torch.from_numpy(rem_p).cuda(), torch.from_numpy(lab.astype(np.int32)).cuda())
    pj = Projector()
    got = pj.project([cl] * 5, fu, fd, H, W, new=True, remove=True,
                     outputs=("idx", "range", "rem", "label", "proj_x", "proj_y"))
    torch.cuda.synchronize()
    for o in got:
        arrs = dict(index=o["idx"].cpu().numpy(), proj_range=o["range"].cpu().numpy(),
                    proj_remissions=o["rem"].cpu().numpy(), proj_label=o["label"].cpu().numpy(),
                    proj_x=o["proj_x"].cpu().numpy(), proj_y=o["proj_y"].cpu().numpy())
        for name, a in arrs.items():
            a = np.ascontiguousarray(a)
            assert str(a.dtype) == str(g[f"f64_new_{name}_dtype"]) and tuple(a.shape) == tuple(g[f"f64_new_{name}_shape"]), name
            assert hashlib.sha256(a.tobytes()).digest() == bytes(g[f"f64_new_{name}_sha256"]), f"{name} differs"
    pj.close()
