"""Marching cubes on the device (replaces TSDFVolume.get_mesh, fusion_lidar.py:403-424).

PINNED to scikit-image 0.18.3 (the reference's dependency): the CPU oracle (oracle/lt_mc_oracle.c) returns the arrays of
the reference's own get_mesh (golden F10, tests/test_mc_cpu.py), and the HIP path is checked against it here -- the same
vertices (positions bit for bit, colours, remissions) and the same faces (each face's vertices in the same order), on
every word-layout corner case of the sign bitmask; element ORDER is the device's own.  And the whole fusion -> mesh ->
range image chain runs without the mesh leaving HBM and equals the host-mesh drop-in call."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _gpu_mesh(t, col, rem, vs, org):
    import torch
    from lidar_transfer_amd.fusion import DeviceMesh
    dev = torch.device("cuda", 0)
    m = DeviceMesh(0)
    m.extract(*[torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (t, col, rem)], vs, org)
    out = [x.cpu().numpy() for x in m.tensors()]
    m.close()
    return out


def _assert_same_mesh(got, want):
    """Device mesh vs the oracle's (= the reference's arrays, golden F10): the same vertices and the same faces, each face's
    vertices in the same order; the order of the ELEMENTS is the device's own (tests/mesh_canon.py)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from mesh_canon import assert_same_mesh
    assert_same_mesh(got, want)


@pytest.mark.parametrize("shape", [(9, 8, 7), (5, 6, 64), (4, 5, 65), (6, 3, 130), (3, 4, 200), (1, 9, 70), (7, 1, 70),
                                   (2, 2, 2), (2, 2, 1), (33, 17, 129)])
def test_white_noise_equals_oracle_on_every_word_layout(oracle, shape):
    """White noise reaches all 256 cases; the shapes put the z rows on every side of the 64-bit word boundary
    (7, 64, 65, 130, 200 = the default volume's nz) and degenerate the other axes."""
    rng = np.random.default_rng(sum(shape))
    t = rng.normal(size=shape).astype(np.float32)
    t[rng.random(shape) < 0.05] = 0.0
    col = (rng.integers(0, 260, shape) * 65536 + rng.integers(0, 256, shape) * 256 + rng.integers(0, 256, shape))
    col = col.astype(np.float32)
    rem = rng.random(shape).astype(np.float32)
    org = np.array([-1.25, 2.5, 0.75], np.float32)
    want = oracle.marching_cubes(t, col, rem, 0.05, org)
    got = _gpu_mesh(t, col, rem, 0.05, org)
    _assert_same_mesh(got, want)
    if min(shape) >= 2 and np.prod(shape) > 100:
        assert want[1].shape[0] > 10


@pytest.mark.parametrize("kind", [0, 1, 2, 3, 4])
def test_fuzz_volume_kinds_equal_oracle(oracle, kind):
    """The volume kinds of tools/mc_lewiner_fuzz.py (on which the oracle equals the real scikit-image, 1 000 volumes) through
    the device path: dense noise, smooth fields, exact zeros, values on a coarse grid (exact ties of the asymptotic decider
    and of the interior test -- scikit-image's `+ eps` in the denominators decides them), clipped TSDF-like fields."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from mc_lewiner_fuzz import volume
    rng = np.random.default_rng(900 + kind)
    faces = 0
    for rep in range(4):
        shape = tuple(int(x) for x in rng.integers(3, 24, 2)) + (int(rng.integers(3, 140)),)
        t = np.ascontiguousarray(volume(rng, kind, shape), np.float32)
        col = (rng.integers(0, 260, shape) * 65536).astype(np.float32)
        rem = rng.random(shape).astype(np.float32)
        org = np.array([0.5, -1.0, 2.0], np.float32)
        want = oracle.marching_cubes(t, col, rem, 0.05, org)
        _assert_same_mesh(_gpu_mesh(t, col, rem, 0.05, org), want)
        faces += want[1].shape[0]
    assert faces > 1000


def test_more_ambiguous_cells_than_the_queue_holds_grow_the_buffers_and_start_over(oracle):
    """White noise makes a third of all cells one of Lewiner's ambiguous cases: 546 000 cells queue ~180 000 of them, more
    than the 65 536 the side buffers start with -- the extraction sees the overflow at its one host synchronisation, grows
    queue and table and runs again; a second extraction into the same mesh object (buffers now large enough, table slots
    emptied by k_mc_clear) and a small volume afterwards give the oracle's meshes too."""
    import torch
    from lidar_transfer_amd.fusion import DeviceMesh
    dev = torch.device("cuda", 0)
    m = DeviceMesh(0)
    org = np.zeros(3, np.float32)
    for k, shape in enumerate([(70, 60, 130), (70, 60, 130), (9, 8, 7)]):
        rng = np.random.default_rng(100 + k)
        t = rng.normal(size=shape).astype(np.float32)
        t[rng.random(shape) < 0.02] = 0.0
        col = (rng.integers(0, 260, shape) * 65536).astype(np.float32)
        rem = rng.random(shape).astype(np.float32)
        want = oracle.marching_cubes(t, col, rem, 0.1, org)
        m.extract(*[torch.from_numpy(a).to(dev) for a in (t, col, rem)], 0.1, (0.0, 0.0, 0.0))
        _assert_same_mesh([a.cpu().numpy() for a in m.tensors()], want)
        assert k == 2 or want[1].shape[0] > 1_000_000
    m.close()


@pytest.mark.parametrize("p_neg", [0.0005, 0.004, 0.02, 0.08, 0.3])
def test_speckled_fields_cross_the_sparse_and_dense_emission_paths(oracle, p_neg):
    """The emission kernel lists a batch of 8 active words with a lane per vertex / cell when the batch holds <= 128 of them
    and with a pass per word otherwise (k_mc_emit_batch): fields that are +1 with isolated negative voxels at five
    densities put batches on both sides of both thresholds, next to each other in one volume."""
    shape = (24, 20, 200)
    rng = np.random.default_rng(int(p_neg * 1e4))
    t = np.ones(shape, np.float32)
    t[rng.random(shape) < p_neg] = -0.5
    t[:, :, 100:] = np.where(rng.random((24, 20, 100)) < 4 * p_neg, -0.25, 1.0)   # denser upper half: mixed batches
    col = (rng.integers(0, 260, shape) * 65536).astype(np.float32)
    rem = rng.random(shape).astype(np.float32)
    org = np.zeros(3, np.float32)
    want = oracle.marching_cubes(t, col, rem, 0.1, org)
    got = _gpu_mesh(t, col, rem, 0.1, org)
    _assert_same_mesh(got, want)
    assert want[0].shape[0] > 50


def test_smooth_field_and_reuse_of_one_mesh_object(oracle):
    import torch
    from lidar_transfer_amd.fusion import DeviceMesh
    dev = torch.device("cuda", 0)
    m = DeviceMesh(0)
    for k, shape in enumerate([(40, 37, 200), (12, 50, 90), (64, 64, 64)]):   # shrinking and growing workspaces
        g = [np.linspace(-1, 1, n) for n in shape]
        x, y, z = np.meshgrid(*g, indexing="ij")
        t = (np.sqrt(x * x + 1.3 * y * y + 0.7 * z * z) - 0.6 + 0.05 * np.sin(9 * x + k) * np.cos(7 * y)).astype(np.float32)
        col = np.full(shape, 40 * 65536, np.float32)
        rem = (0.5 + 0.5 * np.sin(3 * z)).astype(np.float32)
        want = oracle.marching_cubes(t, col, rem, 0.1, np.zeros(3, np.float32))
        m.extract(*[torch.from_numpy(a).to(dev) for a in (t, col, rem)], 0.1, (0.0, 0.0, 0.0), timed=True)
        assert m.n_verts == want[0].shape[0] and m.n_faces == want[1].shape[0] and m.last_ms[0] > 0
        _assert_same_mesh([a.cpu().numpy() for a in m.tensors()], want)
        assert want[1].shape[0] > 2000
    m.close()


def _fused_volume(n_obs=2):
    """A street scene observed by a 64 x 1024 sensor, rendered with the library's own ray cast and fused into a
    25.6 m x 25.6 m x 6.4 m volume at 10 cm (256 x 256 x 64 voxels)."""
    import torch
    from lidar_transfer_amd.fusion import TSDFVolume
    from lidar_transfer_amd.laserscan import create_rays
    from lidar_transfer_amd.raytracer import RaySet, Scene
    from lidar_transfer_amd.synth import synth_scene
    dev = torch.device("cuda", 0)
    H, W, fu, fd = 64, 1024, 15.0, -25.0
    v, f, c, r = synth_scene(5, 60000, bounds=(-14, 14, -14, 14, -5, 5), n_boxes=12, n_poles=8)
    sc = Scene(0)
    sc.set_mesh(*[torch.from_numpy(a).to(dev) for a in (v, f, c, r)])
    rays = torch.from_numpy(create_rays(fu, fd, H, W)).to(dev)
    rs = RaySet(rays, H)
    vol = TSDFVolume(np.array([[-12.8, 12.8], [-12.8, 12.8], [-3.2, 3.2]]), 0.1, fu, fd)
    for k in range(n_obs):
        o = sc.render(rs, (0.0, 0.0, 0.0))
        torch.cuda.synchronize()
        lab = o["endcolors"][:, 2].reshape(H, W).float()
        # the reference images are indexed [row = pitch from the top, column = azimuth], like create_rays' rays
        label3 = torch.stack([lab, torch.zeros_like(lab), torch.zeros_like(lab)], 2)
        # range-image convention of do_range_projection: proj_x = 0.5 * (yaw / pi + 1) with yaw = -atan2(y, x); the
        # rays of create_rays run the same way (column w looks along yaw_w), so the rendered image can be fused as is
        vol.integrate(label3, o["range"].reshape(H, W), o["endrem"].reshape(H, W), np.eye(4))
    rs.close(); sc.close()
    return vol, (H, W, fu, fd)


def test_fused_volume_mesh_equals_oracle_and_renders_in_hbm(oracle):
    import torch
    from lidar_transfer_amd.fusion import MeshVolume
    from lidar_transfer_amd.laserscan import create_rays
    vol, (H, W, fu, fd) = _fused_volume()
    tsdf, weight, color, rem = [t.cpu().numpy() for t in vol.get_volume_tensors()]
    assert (tsdf < 0).sum() > 1000, "the fused volume has no surface"
    want = oracle.marching_cubes(tsdf, color, rem, np.float32(0.1), vol._vol_origin)
    mesh = vol.extract_mesh(timed=True)
    got = [a.cpu().numpy() for a in mesh.tensors()]
    _assert_same_mesh(got, want)
    assert got[1].shape[0] > 20000 and set(np.unique(got[2][:, 2])) <= {0, 10, 40, 50, 80}
    # get_mesh: the reference's 5-tuple (fusion_lidar.py:424)
    v, f, norms, colors, r = vol.get_mesh(None)
    assert v.dtype == np.float32 and f.dtype == np.int32 and colors.dtype == np.uint8 and colors.shape == v.shape
    # ... with scikit-image's vertex numbers (lt_mesh_renumber_dev): the oracle's arrays (= the reference's, golden F10)
    assert np.array_equal(v.view(np.int32), want[0].view(np.int32)) and np.array_equal(f, want[1])
    assert np.array_equal(colors, np.asarray(want[2]).astype(np.uint8)) and np.array_equal(r.view(np.int32), want[3].view(np.int32))
    # the whole chain on the device == the drop-in host call on the downloaded mesh
    HT, WT = 32, 512
    rays = create_rays(10.0, -30.0, HT, WT)
    org = np.array([0.3, -0.2, 0.1], np.float32)
    tup = vol.throw_rays_at_mesh(rays, org, HT, WT, None)
    ref = MeshVolume(v, f, colors, r).throw_rays_at_mesh(rays, org, HT, WT, None)
    names = ["ray_endpoints", "ray_colors", "verts", "colors", "faces", "range_image", "rem_image"]
    for name, a, b in zip(names, tup, ref):
        assert a.shape == b.shape and a.dtype == b.dtype, name
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), name
    assert tup[5].shape == (HT, WT) and (tup[5] > 0).mean() > 0.5
    # the rendered surface is where the fused observations put it: the range image of the mesh seen from the fusion
    # origin reproduces the fused depth within a voxel for the bulk of the rays
    tup0 = vol.throw_rays_at_mesh(create_rays(fu, fd, H, W), np.zeros(3, np.float32), H, W, None)
    vol.close()


def test_default_sized_volume_extraction_is_sparse_work():
    """2000 x 2000 x 200 voxels (config/lidar_transfer.yaml: +-50 m, +-50 m, +-5 m at 5 cm; 4 x 3.2 GB fields): one
    observation fused, mesh extracted -- closed-surface bookkeeping holds (every vertex is referenced, every face
    indexes valid vertices) and the float field is streamed once."""
    import torch
    from lidar_transfer_amd.fusion import TSDFVolume
    dev = torch.device("cuda", 0)
    if torch.cuda.get_device_properties(dev).total_memory < 40 * 2**30:
        pytest.skip("needs 13 GB of volumes")
    vol = TSDFVolume(np.array([[-50.0, 50.0], [-50.0, 50.0], [-5.0, 5.0]]), 0.05, 3.0, -25.0)
    H, W = 64, 2048
    yaw = torch.linspace(-np.pi, np.pi, W, device=dev)
    depth = (14.0 + 6.0 * torch.sin(3 * yaw))[None, :].repeat(H, 1).contiguous()
    lab = torch.full((H, W), 40.0, device=dev)
    label3 = torch.stack([lab, torch.zeros_like(lab), torch.zeros_like(lab)], 2)
    vol.integrate(label3, depth, torch.full((H, W), 0.5, device=dev), np.eye(4))
    mesh = vol.extract_mesh(timed=True)
    v, f, c, r = mesh.tensors()
    assert f.shape[0] > 100000 and int(f.min()) == 0 and int(f.max()) == v.shape[0] - 1
    assert torch.unique(f).numel() == v.shape[0]
    # label 40 on the observed surface; 0 where the nearest voxel of a vertex of the BACK sheet (the jump from -1 to
    # the untouched +1 behind the truncation band, which the reference's volume has just the same) was never observed
    assert bool(((c[:, 2] == 40) | (c[:, 2] == 0)).all()) and float((c[:, 2] == 40).float().mean()) > 0.4
    assert bool(torch.isfinite(v).all())
    lo = torch.tensor([-50.0, -50.0, -5.0], device=dev)
    assert bool((v >= lo).all()) and bool((v <= -lo).all())
    ms_signs, ms_rest = mesh.last_ms
    print(f"default volume: {v.shape[0]} verts, {f.shape[0]} faces, sign pass {ms_signs:.3f} ms, rest {ms_rest:.3f} ms")
    assert ms_signs < 5.0 and ms_rest < 5.0
    vol.close()


def test_mesh_after_reset_and_reintegration_equals_oracle(oracle):
    """The volume keeps the sign bit of every voxel current while it integrates and clears them for the columns a reset
    re-initialises; marching cubes reads those bits instead of the float field.  After reset + one more observation the
    mesh must again be the oracle's mesh of the downloaded volume (and differ from the mesh before the reset), also
    after the fields were modified behind the library's back (`touch`)."""
    import torch
    vol, (H, W, fu, fd) = _fused_volume(n_obs=2)
    before = [a.cpu().numpy() for a in vol.extract_mesh().tensors()]
    vol.reset()
    t, w, c, r = vol.get_volume_tensors()
    assert float(t.min()) == 1.0 and float(w.max()) == 0.0 and float(c.max()) == 0.0
    assert vol.extract_mesh().n_faces == 0
    dev = t.device
    yaw = torch.linspace(-np.pi, np.pi, W, device=dev)
    depth = (7.0 + 2.0 * torch.sin(2 * yaw))[None, :].repeat(H, 1).contiguous()
    lab = torch.full((H, W), 50.0, device=dev)
    label3 = torch.stack([lab, torch.zeros_like(lab), torch.zeros_like(lab)], 2)
    vol.integrate(label3, depth, torch.full((H, W), 0.25, device=dev), np.eye(4))
    tsdf, weight, color, rem = [a.cpu().numpy() for a in vol.get_volume_tensors()]
    want = oracle.marching_cubes(tsdf, color, rem, np.float32(0.1), vol._vol_origin)
    got = [a.cpu().numpy() for a in vol.extract_mesh().tensors()]
    _assert_same_mesh(got, want)
    assert got[1].shape[0] > 5000 and got[1].shape != before[1].shape
    # a caller writes into the field directly: the library must be told, then everything is re-read
    t, w, c, r = vol.get_volume_tensors()
    t[100:110, 100:110, 20:30] = -0.5
    vol.touch()
    tsdf = t.cpu().numpy()
    want = oracle.marching_cubes(tsdf, color, rem, np.float32(0.1), vol._vol_origin)
    got = [a.cpu().numpy() for a in vol.extract_mesh().tensors()]
    _assert_same_mesh(got, want)
    vol.close()


def test_fusion_chains_in_flight_equal_the_single_chain():
    """Output scans are independent (lidar_deform.py:393-462): three chains -- volume, mesh, scene, HIP stream and host thread
    each -- run reset -> integrate (two observations) -> marching cubes -> render concurrently (tools/chain_pipeline.py, what
    bench.py reports as fusion_chain.pipelined); every chain's final range / label image equals the single chain's."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import chain_pipeline
    rec = chain_pipeline.run(chains=3, n=4, n_obs=2, device=0, workload="C1", warm=1, voxel=0.2)
    assert rec["verified"] is True
    assert rec["mesh_faces"] > 10000


def test_fusion_scan_pipeline_equals_the_step_by_step_api():
    """FusionScanPipeline (two chains, three-channel label images, a non-zero render origin, label-image output) against
    TSDFVolume.integrate -> throw_rays_at_mesh_device for the same observations; tickets are handed out once."""
    import torch
    from lidar_transfer_amd.fusion import TSDFVolume
    from lidar_transfer_amd.laserscan import create_rays
    from lidar_transfer_amd.pipeline import FusionScanPipeline
    from lidar_transfer_amd.raytracer import RaySet, Scene
    from lidar_transfer_amd.synth import synth_scene
    dev = torch.device("cuda", 0)
    H, W, fu, fd = 64, 1024, 15.0, -25.0
    v, f, c, r = synth_scene(7, 60000, bounds=(-14, 14, -14, 14, -5, 5), n_boxes=12, n_poles=8)
    sc = Scene(0)
    sc.set_mesh(*[torch.from_numpy(a).to(dev) for a in (v, f, c, r)])
    rays = torch.from_numpy(create_rays(fu, fd, H, W)).to(dev)
    rs = RaySet(rays, H)
    scans = []  # per output scan: two observations from slightly different poses' renders
    for k in range(4):
        obs = []
        for org in ((0.0, 0.0, 0.0), (0.05 * k, -0.03, 0.0)):
            o = sc.render(rs, org)
            torch.cuda.synchronize()
            lab = o["endcolors"][:, 2].reshape(H, W).float()
            obs.append((torch.stack([lab, torch.zeros_like(lab), torch.zeros_like(lab)], 2),
                        o["range"].reshape(H, W).clone(), o["endrem"].reshape(H, W).clone()))
        scans.append(obs)
    bnds = np.array([[-12.8, 12.8], [-12.8, 12.8], [-3.2, 3.2]])
    HT, WT = 32, 512
    rays_t = torch.from_numpy(create_rays(10.0, -30.0, HT, WT)).to(dev)
    origin = (0.3, -0.2, 0.1)
    # step by step
    vol = TSDFVolume(bnds, 0.1, fu, fd)
    rs_t = RaySet(rays_t, HT)
    want = []
    for obs in scans:
        vol.reset()
        for cim, dim, rim in obs:
            vol.integrate(cim, dim, rim, np.eye(4))
        o = vol.throw_rays_at_mesh_device(rs_t, origin, label_image=True)
        torch.cuda.synchronize()
        want.append({k: o[k].clone() for k in ("range", "endcolors", "endrem", "endpoints", "tri")} | {"nf": o["mesh"].n_faces})
    vol.close(); rs_t.close()
    with FusionScanPipeline(bnds, 0.1, fu, fd, rays_t, HT, chains=2, label_image=True) as pipe:
        tickets = [pipe.submit(obs, origin) for obs in scans]
        for t, w in zip(tickets, want):
            got = pipe.wait(t)
            assert got["n_faces"] == w["nf"] and w["nf"] > 10000
            for k in ("range", "endcolors", "endrem", "endpoints", "tri"):
                assert torch.equal(got[k], w[k]), (t, k)
        with pytest.raises(KeyError):
            pipe.wait(tickets[0])
        with pytest.raises(ValueError):
            pipe.submit([(np.zeros((H, W)), scans[0][0][1], scans[0][0][2])])
    rs.close(); sc.close()


def test_bare_lt_fusion_scan_dev_symbol_edge_cases():
    """The native per-scan entry point through ctypes: no observation at all (a fresh volume: no surface, every ray a miss,
    outputs written as misses), NULL handles / NULL image arrays (error code + message, nothing launched), and a scan
    whose only observation is empty (all depths 0: the reference leaves at depth_value == 0)."""
    import ctypes as C
    import torch
    from lidar_transfer_amd import _lib
    from lidar_transfer_amd.fusion import DeviceMesh, TSDFVolume
    from lidar_transfer_amd.laserscan import create_rays
    from lidar_transfer_amd.raytracer import RaySet, Scene
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    H, W = 16, 128
    rays = torch.from_numpy(create_rays(10.0, -20.0, H, W)).to(dev)
    rs = RaySet(rays, H)
    vol = TSDFVolume(np.array([[-6.4, 6.4], [-6.4, 6.4], [-1.6, 1.6]]), 0.1, 10.0, -20.0)
    mesh, sc = DeviceMesh(0), Scene(0)
    out = sc.alloc_outputs(H * W)
    for v in out.values():
        v.fill_(7)
    st = torch.cuda.current_stream(dev)
    sp = C.c_void_p(st.cuda_stream)
    org = (C.c_float * 3)(0, 0, 0)
    vp = C.c_void_p
    none = (vp * 1)()

    def call(vol_h, n_obs, cp, dp, rp):
        return lib.lt_fusion_scan_dev(vol_h, mesh._h, sc._h, rs._h, n_obs, cp, dp, rp, H, W, 1.0, _lib.LT_TSDF_MERGE, org,
                                      out["endpoints"].data_ptr(), out["endcolors"].data_ptr(), out["range"].data_ptr(),
                                      out["endrem"].data_ptr(), out["tri"].data_ptr(), _lib.LT_TRACE_WRITE_MISSES, sp, 1)
    assert call(vol._h, 0, None, None, None) == 0
    assert mesh.n_faces == 0 and mesh.n_verts == 0
    assert int((out["range"] != 0).sum()) == 0 and int((out["tri"] != -1).sum()) == 0
    assert call(None, 0, None, None, None) != 0 and b"lt_fusion_scan_dev" in lib.lt_last_error()
    assert call(vol._h, 1, None, None, None) != 0
    z = torch.zeros((H, W), device=dev)
    cp, dp, rp = (vp * 1)(z.data_ptr()), (vp * 1)(z.data_ptr()), (vp * 1)(z.data_ptr())
    for v in out.values():
        v.fill_(7)
    assert call(vol._h, 1, cp, dp, rp) == 0
    assert mesh.n_faces == 0 and int((out["range"] != 0).sum()) == 0
    del none
    mesh.close(); sc.close(); vol.close(); rs.close()