"""TSDF integration kernel on the GPU.

Parity status: the reference's `integrate` exists only as a CUDA kernel inside a Python string
(auxiliary/fusion_lidar.py:66-229).  tests/test_tsdf_ref_kernel_gpu.py pins the product to that source compiled for
gfx950, bit for bit; here the product is checked against our C restatement of it (oracle/lt_tsdf_oracle.c -- libm's
atan2f / asinf / sqrtf instead of the device library's: boundary voxels and last ulps differ), against the dense HIP
restatement (bit for bit), and the plain-average branch against golden volumes of the reference's own numpy CPU mode (F8)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _images(seed, H, W, fu, fd):
    from lidar_transfer_amd.laserscan import SemLaserScan
    from lidar_transfer_amd.synth import synth_cloud
    pts, rem, lab = synth_cloud(seed, 30000, dtype=np.float64, rmin=3.0, rmax=14.0, fov_up=fu, fov_down=fd)
    sc = SemLaserScan(H, W, 300, {0: [0, 0, 0]})
    sc.points, sc.remissions, sc.label = pts, rem, lab
    sc.do_range_projection(fu, fd, remove=True)
    sc.do_label_projection()
    label3 = np.stack([np.zeros_like(sc.proj_label), np.zeros_like(sc.proj_label), sc.proj_label], 2).astype(np.float32)
    depth = np.where(sc.proj_range > 0, sc.proj_range, 0).astype(np.float32)
    remi = np.where(sc.proj_remissions > 0, sc.proj_remissions, 0).astype(np.float32)
    return label3, depth, remi


def _vs_c_restatement(got, ref, n, what=""):
    """Product volumes against the C restatement.  The C code has libm's atan2f / asinf and a correctly rounded sqrtf where
    the device has its own atan2f / asinf and norm3df (1-ulp v_sqrt_f32): a voxel projecting onto a pixel / fov /
    truncation boundary may read the neighbouring pixel (<= 2e-4 of the volume), and the distance of the others may differ
    by the ulp of the depth.  Weight and class must be bit-identical outside the boundary voxels, tsdf / remission within
    a few ulp of their running averages."""
    flips = (got[1] != ref[1]) | (got[2] != ref[2])
    assert flips.sum() <= 2e-4 * n, f"{what}{flips.sum()} of {n} voxels differ in weight / class"
    ok = ~flips
    for k in (0, 3):
        d = np.abs(got[k][ok].astype(np.float64) - ref[k][ok])
        d = d[np.isfinite(d)]
        far = d > 4e-6
        assert far.sum() <= 2e-4 * n, f"{what}field {k}: {far.sum()} voxels off by more than 4e-6 (max {d.max()})"
    nan_mismatch = np.isnan(got[0][ok]) != np.isnan(ref[0][ok])
    assert not nan_mismatch.any()


@pytest.mark.parametrize("merge", [True, False])
def test_integrate_vs_c_restatement(oracle, merge):
    from lidar_transfer_amd.fusion import TSDFVolume
    H, W, fu, fd = 32, 256, 3.0, -25.0
    bnds = np.array([[-16.0, 16.0], [-16.0, 16.0], [-4.0, 4.0]])
    vol = TSDFVolume(bnds, 0.25, fu, fd, merge=merge)
    dims = tuple(int(x) for x in vol._vol_dim)
    ref = [np.ones(dims, np.float32), np.zeros(dims, np.float32), np.zeros(dims, np.float32),
           np.zeros(dims, np.float32)]
    for seed in (21, 22, 21):  # the third observation repeats the first: exercises the same-class branch
        label3, depth, remi = _images(seed, H, W, fu, fd)
        vol.integrate(label3, depth, remi, np.eye(3), obs_weight=1.)
        folded = np.floor(label3[:, :, 0] * 256 * 256 + label3[:, :, 1] * 256 + label3[:, :, 2]).astype(np.float32)
        oracle.tsdf_integrate(ref, dims, vol._vol_origin, 0.25, fu, fd, folded, depth, remi, 1.0, merge=merge)
    got = [t.cpu().numpy() for t in vol.get_volume_tensors()]
    n = got[0].size
    touched = int((ref[1] > 0).sum() + (ref[0] != 1).sum())
    assert touched > 0.02 * n, "test volume barely touched"
    _vs_c_restatement(got, ref, n)
    vol.close()


def test_wedge_table_follows_the_image_shape(oracle):
    """One volume, observations of three different image shapes in turn (reset between them, and two of the same shape in a
    row without a reset): the pixel-centric integrate rebuilds its wedge table for a new width and its row table for a new
    height; every state against the C restatement of the reference kernel."""
    from lidar_transfer_amd.fusion import TSDFVolume
    fu, fd = 3.0, -25.0
    bnds = np.array([[-16.0, 16.0], [-16.0, 16.0], [-4.0, 4.0]])
    vol = TSDFVolume(bnds, 0.25, fu, fd, merge=True)
    dims = tuple(int(x) for x in vol._vol_dim)
    n = int(np.prod(dims))
    for k, (H, W, repeat) in enumerate([(32, 256, 1), (16, 100, 2), (48, 256, 1), (32, 256, 1)]):
        vol.reset()
        ref = [np.ones(dims, np.float32), np.zeros(dims, np.float32), np.zeros(dims, np.float32), np.zeros(dims, np.float32)]
        for j in range(repeat):
            label3, depth, remi = _images(30 + k + j, H, W, fu, fd)
            vol.integrate(label3, depth, remi, np.eye(3), obs_weight=1.)
            folded = np.floor(label3[:, :, 0] * 256 * 256 + label3[:, :, 1] * 256 + label3[:, :, 2]).astype(np.float32)
            oracle.tsdf_integrate(ref, dims, vol._vol_origin, 0.25, fu, fd, folded, depth, remi, 1.0, merge=True)
        got = [t.cpu().numpy() for t in vol.get_volume_tensors()]
        assert (ref[0] != 1).sum() > 0.005 * n, "test volume barely touched"
        _vs_c_restatement(got, ref, n, f"shape {H}x{W}: ")
    vol.close()


def test_plain_average_branch_vs_reference_numpy_cpu_mode():
    """`merge == false` against the reference's numpy CPU mode (float64 pixel maths, no remissions)."""
    from lidar_transfer_amd.fusion import TSDFVolume
    g = np.load(os.path.join(GOLD, "f8_tsdf_cpu_mode.npz"))
    vol = TSDFVolume(g["bnds"], float(g["voxel"]), float(g["fov_up"]), float(g["fov_down"]), merge=False)
    for _ in range(2):
        vol.integrate(g["label3"], g["depth_im"], g["rem_im"], np.eye(4), obs_weight=1.)
    tsdf, weight, color, rem = [t.cpu().numpy() for t in vol.get_volume_tensors()]
    assert tsdf.shape == g["tsdf"].shape
    n = tsdf.size
    same_w = weight == g["weight"]
    # float32 vs float64 projection: voxels on a pixel / fov / truncation boundary may be classified differently
    assert (~same_w).sum() <= 2e-3 * n, f"{(~same_w).sum()} of {n} weights differ"
    assert (g["weight"] > 0).sum() > 0.02 * n
    assert np.abs(tsdf[same_w] - g["tsdf"][same_w]).max() < 2e-4
    close = np.abs(tsdf - g["tsdf"]) < 1e-3
    assert np.array_equal(color[same_w & close], g["color"][same_w & close])
    vol.close()


def test_numpy_mode_equals_the_reference_numpy_cpu_mode():
    """`mode="numpy"` (LT_TSDF_HOST_MODE: the arithmetic of the reference's numpy branch, fusion_lidar.py:290-388, float64
    voxel projection) against the same golden volumes F8: tsdf, weight and colour bit for bit -- but for the handful of voxels
    that project within an ulp of a pixel boundary, where numpy's own (not correctly rounded) arctan2 / arcsin decide by
    their last bit (tests/test_deform_gpu.py::_check_volumes names them in the composed chains)."""
    from lidar_transfer_amd.fusion import TSDFVolume
    g = np.load(os.path.join(GOLD, "f8_tsdf_cpu_mode.npz"))
    vol = TSDFVolume(g["bnds"], float(g["voxel"]), float(g["fov_up"]), float(g["fov_down"]), mode="numpy")
    for _ in range(2):
        vol.integrate(g["label3"], g["depth_im"], g["rem_im"], np.eye(4), obs_weight=1.)
    tsdf, weight, color, rem = [t.cpu().numpy() for t in vol.get_volume_tensors()]
    n = tsdf.size
    diff = (tsdf.view(np.int32) != g["tsdf"].view(np.int32)) | (weight != g["weight"]) | (color != g["color"])
    assert diff.sum() <= max(2, 1e-5 * n), f"{diff.sum()} of {n} voxels differ"
    assert (g["weight"] > 0).sum() > 0.02 * n and float(np.abs(rem).max()) == 0.0
    vol.close()


def test_class_aware_branch_degenerates_to_the_pinned_average_for_a_single_class():
    """The strongest pin of `merge == true` available without running CUDA: when every pixel carries label 0 -- the
    label the fresh colour volume holds -- the class-aware kernel only ever takes its same-class branch
    (fusion_lidar.py:191-203), whose tsdf / weight arithmetic is the plain running average.  Its volumes must then be
    the golden ones of the reference's numpy CPU mode (F8: tsdf and weight there do not depend on the labels), and
    bit-identical to this library's own `merge == false` kernel on tsdf, weight and remission."""
    from lidar_transfer_amd.fusion import TSDFVolume
    g = np.load(os.path.join(GOLD, "f8_tsdf_cpu_mode.npz"))
    zero3 = np.zeros_like(g["label3"])
    vols = {}
    for merge in (True, False):
        vol = TSDFVolume(g["bnds"], float(g["voxel"]), float(g["fov_up"]), float(g["fov_down"]), merge=merge)
        for _ in range(2):
            vol.integrate(zero3, g["depth_im"], g["rem_im"], np.eye(4), obs_weight=1.)
        vols[merge] = [t.cpu().numpy() for t in vol.get_volume_tensors()]
        vol.close()
    tsdf, weight, color, rem = vols[True]
    for k in (0, 1, 3):  # tsdf, weight, remission: the two kernels agree bit for bit
        assert np.array_equal(vols[True][k].view(np.int32), vols[False][k].view(np.int32)), k
    assert float(np.abs(color).max()) == 0.0
    n = tsdf.size
    same_w = weight == g["weight"]
    assert (~same_w).sum() <= 2e-3 * n, f"{(~same_w).sum()} of {n} weights differ"
    assert (g["weight"] > 0).sum() > 0.02 * n and (weight == 2).sum() > 0.01 * n
    assert np.abs(tsdf[same_w] - g["tsdf"][same_w]).max() < 2e-4


def test_volume_geometry_and_reset():
    from lidar_transfer_amd.fusion import TSDFVolume
    vol = TSDFVolume(np.array([[-1.0, 1.0], [-2.0, 2.0], [0.0, 0.55]]), 0.1, 3.0, -25.0)
    t, w, c, r = vol.get_volume_tensors()
    assert tuple(t.shape) == (20, 40, 6) and float(t.min()) == 1.0 and float(w.max()) == 0.0
    assert np.allclose(vol._vol_bnds[:, 1], [1.0, 2.0, 0.6])
    vol.close()


_AB_SCRIPT = r"""
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, %r)
import torch
from lidar_transfer_amd.fusion import TSDFVolume
merge = sys.argv[2] == "1"
fu, fd = float(sys.argv[3]), float(sys.argv[4])
voxel = float(sys.argv[5]) if len(sys.argv) > 5 else 0.05
H, W = (int(sys.argv[6]), int(sys.argv[7])) if len(sys.argv) > 7 else (32, 256)
rng = np.random.default_rng(5)
yaw = np.linspace(-np.pi, np.pi, W)
half = 15.0 if voxel < 0.1 else 20.0
vol = TSDFVolume(np.array([[-half, half], [-half, half], [-5.0, 5.0]]), voxel, fu, fd, merge=merge)   # 0.05: 600 x 600 x 200 = 72 M voxels
if os.environ.get("LT_TEST_TSDF") == "dense":
    # the A/B partner: the one-thread-per-voxel restatement of the reference kernel (oracle/lt_tsdf_dense.hip, an
    # ORACLE library) writes the volume's fields through their raw pointers; the product only allocates and resets them
    from oracle import binding as ob
    dense = ob.dense_lib()
    def integrate_dense(color_im, depth_im, rem_im, cam_pose=None, obs_weight=1.):
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
        c = dev(color_im)
        folded = torch.floor(c[:, :, 0] * 256 * 256 + c[:, :, 1] * 256 + c[:, :, 2]).contiguous()   # fusion_lidar.py:261-264
        d, r = dev(depth_im), dev(rem_im)
        views = vol.get_volume_tensors()          # strided views of the volume's (tsdf, weight, colour, remission) records
        t, w, cv, rv = [x.contiguous() for x in views]   # the oracle kernel works on four packed arrays, as the reference does
        dims = (C.c_int * 3)(*[int(x) for x in t.shape])
        org = (C.c_float * 3)(*[float(x) for x in vol._vol_origin])
        vp = C.c_void_p
        rc = dense.lt_test_tsdf_integrate_dense(vp(t.data_ptr()), vp(w.data_ptr()), vp(cv.data_ptr()), vp(rv.data_ptr()), dims, org,
                                                C.c_float(np.float32(voxel)), C.c_float(np.float32(voxel * 5)), C.c_float(fu), C.c_float(fd),
                                                vp(folded.data_ptr()), vp(d.data_ptr()), vp(r.data_ptr()), H, W,
                                                C.c_float(obs_weight), 1 if merge else 0, vp(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, rc
        for dst, src in zip(views, (t, w, cv, rv)):
            dst.copy_(src)
        torch.cuda.synchronize()
        vol.touch()   # written without column stamps: reset / extraction must take every column for written
    vol.integrate = integrate_dense
out = {}
for rnd in range(2):          # second round: after a reset the volume must equal a fresh one
    for k in range(2):
        depth = (6.0 + 3.0 * np.sin(3 * yaw + k)[None, :] + 0.2 * rng.random((H, W))).astype(np.float32)
        depth[rng.random((H, W)) < 0.05] = 0.0
        depth[:, 40 * W // 256:60 * W // 256] = 0.0            # image columns without any return
        depth[:, 100 * W // 256:110 * W // 256] = -1.0         # ... and with the reference's "no data" value (laserscan.py:38)
        depth[rng.random((H, W)) < 0.003] = np.nan            # broken pixels: both kernels must treat them alike
        depth[rng.random((H, W)) < 0.003] = np.inf
        lab = rng.choice(np.array([0.0, 40.0, 50.0]), (H, W)).astype(np.float32)
        label3 = np.stack([lab, np.zeros_like(lab), np.zeros_like(lab)], 2)
        rem = rng.random((H, W)).astype(np.float32)
        vol.integrate(label3, depth, rem, np.eye(4))
    if rnd == 0:
        out["first"] = [t.cpu().numpy() for t in vol.get_volume_tensors()]
        vol.reset()
        rng = np.random.default_rng(5)
    else:
        out["second"] = [t.cpu().numpy() for t in vol.get_volume_tensors()]
# the caller writes into the volumes through the raw tensors (a region of the SAME class as some pixels, in columns no
# observation has stamped yet after a reset) and says so: nothing may be taken for fresh from here on
vol.reset()
ts = vol.get_volume_tensors()
rx, ry = (slice(60, 160), slice(380, 520)) if voxel < 0.1 else (slice(20, 60), slice(100, 140))
ts[0][rx, ry, :] = 0.25; ts[1][rx, ry, :] = 2.0; ts[2][rx, ry, :] = 40.0 * 65536.0
vol.touch()
depth = (7.0 + 2.0 * np.cos(2 * yaw)[None, :] + 0.1 * rng.random((H, W))).astype(np.float32)
lab = rng.choice(np.array([0.0, 40.0, 50.0]), (H, W)).astype(np.float32)
vol.integrate(np.stack([lab, np.zeros_like(lab), np.zeros_like(lab)], 2), depth, rng.random((H, W)).astype(np.float32), np.eye(4))
out["third"] = [t.cpu().numpy() for t in vol.get_volume_tensors()]
np.savez(sys.argv[1], **{f"{k}{i}": a for k, v in out.items() for i, a in enumerate(v)})
"""


@pytest.mark.parametrize("merge,fu,fd,voxel,hw", [(True, 10.0, -25.0, 0.05, (32, 256)), (False, 10.0, -25.0, 0.05, (32, 256)),
                                                   (True, 40.0, -50.0, 0.05, (32, 256)), (True, 2.0, -24.8, 0.05, (32, 256)),
                                                   (True, 10.0, -25.0, 0.25, (32, 256)), (False, 10.0, -25.0, 0.25, (32, 256)),
                                                   (True, 15.0, -20.0, 0.25, (25, 301)), (True, 5.0, -30.0, 0.25, (70, 97))])
def test_column_aware_integrate_equals_dense_kernel_bit_for_bit(tmp_path, merge, fu, fd, voxel, hw):
    """The work-saving integrate (per-column image column and dead-column test, conservative sine test, dirty-column
    reset) against the plain one-thread-per-voxel restatement of the reference kernel (oracle/lt_tsdf_dense.hip: an ORACLE
    library, not in liblidarhip.so) on a 72 M-voxel volume -- beyond 2^24
    voxels, where the reference's float voxel index misplaces voxels next to x boundaries -- two observations, a
    reset, the same two observations again: all four fields bit-identical, and the volume after the reset round equals
    the first round.  The first observation of a round is driven by the pixels (k_tsdf_integrate_pix: wedge table, rows'
    z intervals, colour-0 pixels as long runs, the y = dim_y - 1 columns voxel by voxel) and so is the second, after a pass
    over the voxels inside the columns' previously written ranges (k_tsdf_integrate_written) -- or, with LIDARHIP_TSDF_PIX=0
    ("walk"), both by the column walk (band test + candidate queue, then the exact evaluation of every voxel), with =1
    ("pix1") the first by the pixels and the second by the walk; (40, -50) degrees is a field of view for which the band test is
    switched off; zero, NaN and infinite depth pixels and colour 0 (the fresh volume's own) are in the images, and image
    columns holding only the reference's "no data" depth -1 -- with voxel_size 0.25 the truncation margin is 1.25 m, so the
    voxels within 0.25 m of the sensor ARE written through such pixels (depth_diff = -1 - depth >= -trunc_margin).  Image
    shapes 25 x 301 and 70 x 97: the pixel kernel's workgroups of 64 pixels then straddle image columns, and the wedge table
    is built for a width that is not a power of two."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for mode in ("cols", "walk", "pix1", "dense"):
        env = dict(os.environ)
        env.pop("LT_TEST_TSDF", None)
        env.pop("LIDARHIP_TSDF_PIX", None)
        if mode == "dense":
            env["LT_TEST_TSDF"] = "dense"
        elif mode == "walk":   # the column walk for every observation (default: driven by the pixels)
            env["LIDARHIP_TSDF_PIX"] = "0"
        elif mode == "pix1":   # by the pixels on a fresh volume, the column walk for the observations after the first
            env["LIDARHIP_TSDF_PIX"] = "1"
        path = str(tmp_path / f"{mode}.npz")
        r = subprocess.run([sys.executable, "-c", _AB_SCRIPT % root, path, "1" if merge else "0", str(fu), str(fd), str(voxel),
                            str(hw[0]), str(hw[1])],
                           env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[mode] = np.load(path)
    for key in res["cols"].files:
        for other in ("dense", "walk", "pix1"):
            a, b = res["cols"][key], res[other][key]
            assert np.array_equal(a.view(np.int32), b.view(np.int32)), (key, other)
    for i in range(4):
        assert np.array_equal(res["cols"][f"first{i}"].view(np.int32), res["cols"][f"second{i}"].view(np.int32)), i
    rx, ry = (slice(60, 160), slice(380, 520)) if voxel < 0.1 else (slice(20, 60), slice(100, 140))
    assert (res["cols"]["third0"][rx, ry, :] != 0.25).any()   # the touched region was observed
    t = res["cols"]["first0"]
    assert (t < 0).sum() > (10000 if voxel < 0.1 else 1000) and (t != 1).mean() < 0.5
    if voxel > 0.2:   # voxels next to the sensor written through the all -1 image columns (trunc_margin 1.25 m > 1)
        assert (res["dense"]["first1"] != 0).sum() > 0


@pytest.mark.parametrize("fu,fd,voxel,hw,n_obs", [(10.0, -25.0, 0.05, (32, 256), 4), (5.0, -30.0, 0.25, (70, 97), 3),
                                                    (10.0, -25.0, 0.05, (32, 256), 11), (40.0, -50.0, 0.05, (32, 256), 2)])
def test_fused_observations_equal_one_integrate_per_observation(fu, fd, voxel, hw, n_obs):
    """lt_tsdf_integrate_multi_dev -- all observations of a fresh volume in ONE pixel pass, the updates applied in order on
    the voxel's state in registers -- against one lt_tsdf_integrate_dev per observation (itself bit-identical to the
    one-thread-per-voxel restatement of the reference kernel, test above): the four fields bit for bit, the mesh extracted
    from the kept sign bits, and once more after a reset.  The observations differ from each other the way re-projected
    neighbouring scans do -- centimetre noise, holes, other labels (the "other class" branch), label 0 (the fresh volume's
    own class: long runs), no-data columns, NaN / infinite pixels -- plus one observation whose surface lies a metre in front
    of the others' and one a metre behind (bands that do not overlap).  11 observations: more than one fused pass holds
    (8), the rest take the single-observation path on the then non-fresh volume."""
    import torch
    from lidar_transfer_amd.fusion import TSDFVolume
    H, W = hw
    rng = np.random.default_rng(17)
    yaw = np.linspace(-np.pi, np.pi, W)
    half = 15.0 if voxel < 0.1 else 20.0
    bnds = np.array([[-half, half], [-half, half], [-5.0, 5.0]])
    base = (6.0 + 3.0 * np.sin(3 * yaw)[None, :] + 0.2 * rng.random((H, W))).astype(np.float32)
    lab0 = rng.choice(np.array([0.0, 40.0, 50.0]), (H, W), p=[0.1, 0.6, 0.3]).astype(np.float32)
    obs = []
    for k in range(n_obs):
        depth = (base + 0.02 * rng.standard_normal((H, W))).astype(np.float32)
        if k == 1:
            depth = (depth - 1.0).astype(np.float32)       # a surface a metre closer
        if k == 2:
            depth = (depth + 1.0).astype(np.float32)       # ... and a metre further
        depth[rng.random((H, W)) < 0.05] = 0.0
        depth[:, 40 * W // 256:60 * W // 256] = 0.0
        depth[:, 100 * W // 256:110 * W // 256] = -1.0
        depth[rng.random((H, W)) < 0.003] = np.nan
        depth[rng.random((H, W)) < 0.003] = np.inf
        lab = lab0.copy()
        flip = rng.random((H, W)) < 0.05
        lab[flip] = rng.choice(np.array([0.0, 40.0, 50.0, 10.0]), int(flip.sum()))
        label3 = np.stack([lab, np.zeros_like(lab), np.zeros_like(lab)], 2)
        obs.append((label3, depth, rng.random((H, W)).astype(np.float32)))
    seq = TSDFVolume(bnds, voxel, fu, fd)
    fused = TSDFVolume(bnds, voxel, fu, fd)
    for rnd in range(2):
        for o in obs:
            seq.integrate(*o, np.eye(4))
        fused.integrate_multi(obs)
        torch.cuda.synchronize()
        a, b = seq.get_volume_tensors(), fused.get_volume_tensors()
        for name, x, y in zip(("tsdf", "weight", "color", "rem"), a, b):
            same = torch.equal(x.view(torch.int32), y.view(torch.int32))
            assert same, f"round {rnd}: {name} differs in {int((x.view(torch.int32) != y.view(torch.int32)).sum())} voxels"
        ma = [t.cpu().numpy() for t in seq.extract_mesh().tensors()]
        mb = [t.cpu().numpy() for t in fused.extract_mesh().tensors()]
        assert ma[1].shape == mb[1].shape and all(np.array_equal(p, q) for p, q in zip(ma, mb))
        if rnd == 0:
            assert int((a[0] < 0).sum()) > 1000 and int((a[1] > 1.5).sum()) > 1000, "the observations did not overlap"
            seq.reset(); fused.reset()
    seq.close(); fused.close()


def test_host_mode_and_merge_flags_exclude_each_other():
    """include/lidarhip.h: LT_TSDF_HOST_MODE (the reference's numpy branch) has no class-aware update -- both flags at once
    are an argument error, not a silent choice."""
    import ctypes as C
    import torch
    from lidar_transfer_amd import _lib
    from lidar_transfer_amd.fusion import TSDFVolume
    vol = TSDFVolume(np.array([[-2.0, 2.0], [-2.0, 2.0], [-1.0, 1.0]]), 0.1, 3.0, -25.0)
    im = torch.zeros((16, 64), device="cuda")
    lib = _lib.load()
    rc = lib.lt_tsdf_integrate_dev(vol._h, im.data_ptr(), im.data_ptr(), im.data_ptr(), 16, 64, 1.0,
                                   _lib.LT_TSDF_HOST_MODE | _lib.LT_TSDF_MERGE, None)
    assert rc == -1 and b"excludes" in lib.lt_last_error()
    cp = (C.c_void_p * 1)(im.data_ptr())
    rc = lib.lt_tsdf_integrate_multi_dev(vol._h, 1, cp, cp, cp, 16, 64, 1.0, _lib.LT_TSDF_HOST_MODE | _lib.LT_TSDF_MERGE, None)
    assert rc == -1
    vol.close()
