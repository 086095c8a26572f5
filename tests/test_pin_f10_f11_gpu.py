"""The two rows of the scope table that need the reference's own ENVIRONMENT -- marching cubes (SURVEY.md section 8f-2;
scikit-image's Lewiner implementation is the reference's dependency: golden F10 was made with the real scikit-image 0.18.3 of
the build image's /opt/conda interpreter, so these tests RUN) and the
class-aware TSDF update as a CUDA run (8f-1; pinned in this image to the kernel's source compiled for gfx950,
tests/test_tsdf_ref_kernel_gpu.py -- pycuda + an NVIDIA GPU would add CUDA's own last ulps) -- against golden fixtures
made by the REAL reference:

    tests/golden/make_golden_mc.py         -> tests/golden/f10_mc_<case>.npz
    tests/golden/make_golden_tsdf_cuda.py  -> tests/golden/f11_tsdf_cuda.npz

F10 is committed; the F11 test SKIPS until a maintainer with pycuda + an NVIDIA GPU has run its generator once and committed
the file (recipe: INTEGRATION.md, "How marching cubes and the fusion kernel are pinned").
Inputs come from tests/pin_cases.py (seeded, numpy only), shared with the generators."""
import os

import numpy as np
import pytest

import pin_cases  # noqa: E402  (tests/ is on sys.path: conftest)

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _fixture(name):
    path = os.path.join(GOLD, name)
    if not os.path.exists(path):
        pytest.skip(f"{name} not generated yet (needs the reference's scikit-image / pycuda environment: "
                    f"tests/golden/make_golden_*.py)")
    return np.load(path)


def _device_mesh(case):
    import torch
    from lidar_transfer_amd.fusion import DeviceMesh
    dev = torch.device("cuda", 0)
    tsdf, color, rem, vs, org = pin_cases.mc_case(case)
    m = DeviceMesh(0)
    m.extract(*[torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (tsdf, color, rem)], float(vs), org)
    return m, float(vs)


@pytest.mark.parametrize("case", sorted(pin_cases.MC_CASES))
def test_f10_marching_cubes_vertex_set_equals_scikit_image(case):
    """Same vertex SET as `measure.marching_cubes_lewiner` + fusion_lidar.py:409-423 (positions bit for bit: one vertex per
    sign-changing lattice edge and Lewiner's centre vertices, the centre-of-mass rule in double, float32 storage, float32
    world transform), same colours (uint8 wrap) and remissions per vertex, and the same FACE set: every face of the
    reference's get_mesh with its three vertices in the same order.  Element ORDER is not compared (lt_mc.hip has its own)."""
    g = _fixture(f"f10_mc_{case}.npz")
    m, _ = _device_mesh(case)
    v, f, c, r = [t.cpu().numpy() for t in m.tensors()]
    m.close()
    order = np.lexsort((v[:, 2], v[:, 1], v[:, 0]))
    assert v.shape == g["verts_sorted"].shape, (v.shape, g["verts_sorted"].shape)
    assert np.array_equal(v[order].view(np.int32), g["verts_sorted"].view(np.int32))
    assert np.array_equal(c[order].astype(np.uint8), g["colors_sorted"])
    assert np.array_equal(r[order].view(np.int32), g["rem_sorted"].view(np.int32))
    assert f.shape[0] == int(g["n_faces"])
    from mesh_canon import assert_same_mesh
    assert_same_mesh((v, f, c, r), (g["verts"], g["faces"], g["colors"].astype(np.int32), g["vrem"]), f"{case}: ")
    # ... and with scikit-image's vertex numbers (lt_mesh_renumber_dev): the reference's arrays, element for element
    m, _ = _device_mesh(case)
    v, f, c, r = [t.cpu().numpy() for t in m.renumber().tensors()]
    m.close()
    assert np.array_equal(v.view(np.int32), g["verts"].view(np.int32)), "verts"
    assert np.array_equal(f, g["faces"]), "faces"
    assert np.array_equal(c.astype(np.uint8), g["colors"]) and np.array_equal(r.view(np.int32), g["vrem"].view(np.int32))


@pytest.mark.parametrize("case", [k for k, s in sorted(pin_cases.MC_CASES.items()) if s is not None])
def test_f10_render_of_the_device_mesh_equals_render_of_the_scikit_image_mesh(case):
    """The image of OUR mesh through OUR ray cast against the image the reference's raytracer made of scikit-image's mesh:
    the meshes are the same triangles with the same vertex order, so the images are the same up to exact-t ties between
    different faces (which face of two sharing an edge reports the hit depends on the order they are visited in): label
    image identical, range image bit-identical on >= 99.9 % of the pixels and within 1e-5 m everywhere."""
    import torch
    from lidar_transfer_amd.laserscan import create_rays
    from lidar_transfer_amd.raytracer import RaySet, Scene
    g = _fixture(f"f10_mc_{case}.npz")
    H, W = int(g["H"]), int(g["W"])
    m, vs = _device_mesh(case)
    dev = torch.device("cuda", 0)
    rays = torch.from_numpy(create_rays(float(g["fov_up"]), float(g["fov_down"]), H, W)).to(dev)
    rs, sc = RaySet(rays, H), Scene(0)
    sc.set_device_mesh(m)
    o = sc.render(rs, (0.0, 0.0, 0.0), label_image=True)
    torch.cuda.synchronize()
    rng_, lab = o["range"].cpu().numpy().reshape(H, W), o["endcolors"].cpu().numpy().reshape(H, W)
    sc.close(); rs.close(); m.close()
    want_l, want_r = g["label"], g["range"]
    same_bits = rng_.view(np.int32) == np.asarray(want_r, np.float32).view(np.int32)
    print(f"\n{case}: {same_bits.mean():.6f} of the range pixels bit-identical, labels equal {np.mean(lab == want_l):.6f}, "
          f"max |d range| {np.abs(rng_ - want_r).max():.3g}")
    assert np.mean(lab == want_l) >= 0.9999
    assert same_bits.mean() >= 0.999
    assert (rng_ > 0).mean() > 0.5
    assert np.abs(rng_ - want_r)[lab == want_l].max() <= 1e-5


def test_f11_class_aware_integrate_equals_the_cuda_kernel():
    """The four volumes after each of three multi-class observations against the reference's pycuda kernel.  Bit-equal up
    to the documented boundary voxels: asinf / atan2f of two math libraries may put a voxel on the other side of a pixel or
    field-of-view boundary (<= 2e-4 of the voxels, the allowance of the C-restatement test)."""
    from lidar_transfer_amd.fusion import TSDFVolume
    g = _fixture("f11_tsdf_cuda.npz")
    vol = TSDFVolume(pin_cases.TSDF_BOUNDS.copy(), pin_cases.TSDF_VOXEL, *pin_cases.TSDF_FOV, merge=True)
    for k, (label3, depth, rem) in enumerate(pin_cases.tsdf_observations(int(g["n_obs"]))):
        vol.integrate(label3, depth, rem, np.eye(4))
        got = [t.cpu().numpy() for t in vol.get_volume_tensors()]
        for name, a in zip(("tsdf", "weight", "color", "rem"), got):
            want = g[f"{name}_{k}"]
            bad = (a.view(np.int32) != want.view(np.int32)).mean()
            assert bad <= 2e-4, f"observation {k}, {name}: {bad:.2e} of the voxels differ"
        assert (got[0] != 1).sum() > 1000
    vol.close()


@pytest.mark.parametrize("case", sorted(pin_cases.MC_CASES))
def test_pin_cases_run_through_the_device_path_today(oracle, case):
    """Not skipped: the seeded cases the fixtures will be made from already run through lt_mc.hip (bit-equal to the CPU
    oracle, as every other volume) and, where a sensor model is given, produce a meaningful image."""
    import torch
    from lidar_transfer_amd.laserscan import create_rays
    from lidar_transfer_amd.raytracer import RaySet, Scene
    tsdf, color, rem, vs, org = pin_cases.mc_case(case)
    want = oracle.marching_cubes(tsdf, color, rem, vs, org)
    m, _ = _device_mesh(case)
    got = [t.cpu().numpy() for t in m.tensors()]
    from mesh_canon import assert_same_mesh
    assert_same_mesh(got, want)
    assert want[1].shape[0] > 1000
    sensor = pin_cases.MC_CASES[case]
    if sensor is not None:
        H, W, fu, fd = sensor
        rays = torch.from_numpy(create_rays(fu, fd, H, W)).to(torch.device("cuda", 0))
        rs, sc = RaySet(rays, H), Scene(0)
        sc.set_device_mesh(m)
        o = sc.render(rs, (0.0, 0.0, 0.0), label_image=True)
        torch.cuda.synchronize()
        assert float((o["range"] > 0).float().mean()) > 0.6
        assert len(np.unique(o["endcolors"].cpu().numpy())) >= 2
        sc.close(); rs.close()
    m.close()


def test_f15_default_volume_mesh_and_render_equal_the_references_get_mesh_and_raytracer():
    """Golden F15 (tests/golden/make_golden_mc_full.py): the reference's `get_mesh` -- the real scikit-image 0.18.3 over the
    DEFAULT 2000 x 2000 x 200 volume -- and `throw_rays_at_mesh` -- the reference's C++ raytracer, 64 x 2048 rays -- on the
    seeded street field of pin_cases.mc_full_fields (rebuilt here on the device from the same exact float32 operations).
    The device extracts (lt_marching_cubes_dev), renumbers (lt_mesh_renumber_dev) and must hold the reference's three arrays
    element for element (SHA-256); its render of its own mesh must be the reference's image but for rays that run IN a lattice
    plane through the sensor (the seam columns yaw = +-pi: marching-cubes vertices lie exactly on y = 0, those rays hit triangle
    EDGES, where the reference's unpadded slab test / first-visited tie rule may report another triangle than the closest --
    DESIGN.md section 3) -- a handful of pixels, named; labels identical everywhere."""
    import hashlib
    import torch
    from lidar_transfer_amd.fusion import DeviceMesh
    from lidar_transfer_amd.laserscan import create_rays
    from lidar_transfer_amd.raytracer import RaySet, Scene
    g = _fixture("f15_mc_full.npz")
    dev = torch.device("cuda", 0)
    tsdf, color, rem = pin_cases.mc_full_fields(torch, device=dev)
    assert tuple(tsdf.shape) == pin_cases.MC_FULL_DIMS
    m = DeviceMesh(0)
    m.extract(tsdf, color, rem, pin_cases.MC_FULL_VOXEL, pin_cases.MC_FULL_ORIGIN)
    assert (m.n_verts, m.n_faces) == (int(g["n_verts"]), int(g["n_faces"])), (m.n_verts, m.n_faces, int(g["n_verts"]), int(g["n_faces"]))
    H, W = int(g["H"]), int(g["W"])
    rays = torch.from_numpy(create_rays(float(g["fov_up"]), float(g["fov_down"]), H, W)).to(dev)
    rs, sc = RaySet(rays, H), Scene(0)
    sc.set_device_mesh(m)
    o = sc.render(rs, (0.0, 0.0, 0.0), label_image=True)
    torch.cuda.synchronize()
    rng_, lab = o["range"].cpu().numpy().reshape(H, W), o["endcolors"].cpu().numpy().reshape(H, W)
    remi = o["endrem"].cpu().numpy().reshape(H, W)
    sc.close(); rs.close()
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()  # noqa: E731
    v, f, c, r = m.renumber().tensors()
    got_sha = [sha(v.cpu().numpy()), sha(f.cpu().numpy()), sha(c.cpu().numpy().astype(np.uint8))]
    m.close()
    del tsdf, color, rem
    assert got_sha == [str(x) for x in g["mesh_sha"]], "the renumbered mesh arrays differ from scikit-image's"
    assert int(g["label_max"]) < 256 and np.array_equal(lab, g["label"].astype(np.int32)), "labels"
    same = rng_.view(np.int32) == g["range"].view(np.int32)
    bad = np.argwhere(~same)
    print(f"\nF15: {int(g['n_faces'])} faces; {same.mean():.6f} of the {H * W} range pixels bit-identical; differing pixels: {bad.tolist()}")
    assert (g["range"] > 0).mean() > 0.9
    assert len(bad) <= 16
    # every differing pixel is a ray inside a lattice plane through the sensor: direction with a zero (|.| < 1e-12) x or y
    d = create_rays(float(g["fov_up"]), float(g["fov_down"]), H, W).reshape(H, W, 3)
    for row, col in bad:
        assert min(abs(float(d[row, col, 0])), abs(float(d[row, col, 1]))) < 1e-7, (row, col, d[row, col].tolist())
    assert np.array_equal(remi.view(np.int32)[same], g["rem"].view(np.int32)[same])
