"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bar (BASELINE.json north_star): hit triangle / label bit-exact, range within 1e-4 m.  Because the
HIP kernels reproduce the reference's float32 operation order (no FMA), we hold them to MORE than
that against our oracle: every output bit-identical to the brute-force closest hit.
"""
import os

import numpy as np
import pytest

from lidar_transfer_amd.laserscan import create_rays
from lidar_transfer_amd.synth import synth_scene

pytestmark = pytest.mark.gpu


def _run_ctrace(rays, origin, v, f, c, r, H, W, with_stats=False):
    from lidar_transfer_amd.raytracer import C_Trace
    n = H * W
    out = dict(endpoints=np.zeros(3 * n, np.float32), endcolors=np.zeros(3 * n, np.int32),
               range=np.zeros(n, np.float32), endrem=np.zeros(n, np.float32), tri=np.full(n, -1, np.int32))
    stats = {} if with_stats else None
    C_Trace(rays.reshape(-1), origin, v.reshape(-1), f.reshape(-1), c.reshape(-1), r.reshape(-1), out["endpoints"],
            out["endcolors"], out["range"], out["endrem"], H, W, tri_image=out["tri"], stats=stats)
    out["endpoints"] = out["endpoints"].reshape(-1, 3)
    out["endcolors"] = out["endcolors"].reshape(-1, 3)
    out["stats"] = stats
    return out


def _assert_bits(a, b, what):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    assert a.shape == b.shape, what
    bad = np.nonzero(a.view(np.int32).reshape(-1) != b.view(np.int32).reshape(-1))[0]
    assert bad.size == 0, f"{what}: {bad.size} of {a.size} elements differ, first at {bad[:5]}"


@pytest.mark.parametrize("ntri,H,W,seed", [(2000, 16, 64, 0), (2000, 16, 64, 1), (50000, 64, 256, 0)])
def test_ctrace_vs_bruteforce_bitexact(oracle, ntri, H, W, seed):
    v, f, c, r = synth_scene(seed, ntri)
    rays = create_rays(3, -25, H, W)
    org = np.zeros(3, np.float32)
    got = _run_ctrace(rays, org, v, f, c, r, H, W)
    ref = oracle.oracle_trace(rays, org, v, f, c, r, H, mode=oracle.MODE_BRUTE, norm=oracle.NORM_SSE_TABLE)
    for k in ("tri", "endcolors", "range", "endrem", "endpoints"):
        _assert_bits(got[k], ref[k], k)


def test_ctrace_overlapping_scene_ties(oracle):
    """Coincident double surfaces: equal-t ties must resolve to the lower face index."""
    v, f, c, r = synth_scene(3, 20000, allow_overlap=True)
    H, W = 32, 128
    rays = create_rays(3, -25, H, W)
    org = np.zeros(3, np.float32)
    got = _run_ctrace(rays, org, v, f, c, r, H, W)
    ref = oracle.oracle_trace(rays, org, v, f, c, r, H, mode=oracle.MODE_BRUTE, norm=oracle.NORM_SSE_TABLE)
    for k in ("tri", "endcolors", "range", "endrem", "endpoints"):
        _assert_bits(got[k], ref[k], k)


def test_ctrace_offset_origin_and_unnormalised_rays(oracle):
    v, f, c, r = synth_scene(5, 30000)
    H, W = 16, 96
    rays = create_rays(10, -30, H, W) * np.float32(2.5)  # normalisation happens inside (Vector3.h:73-89)
    org = np.array([1.5, -2.25, 0.4], np.float32)
    got = _run_ctrace(rays, org, v, f, c, r, H, W)
    ref = oracle.oracle_trace(rays, org, v, f, c, r, H, mode=oracle.MODE_BRUTE, norm=oracle.NORM_SSE_TABLE)
    for k in ("tri", "endcolors", "range", "endrem", "endpoints"):
        _assert_bits(got[k], ref[k], k)


def test_ctrace_200k_vs_reference_bvh_restatement(oracle):
    """C1-sized case against the restatement of the reference's own BVH (same normalisation)."""
    v, f, c, r = synth_scene(0, 200000)
    H, W = 64, 1024
    rays = create_rays(3, -25, H, W)
    org = np.zeros(3, np.float32)
    got = _run_ctrace(rays, org, v, f, c, r, H, W, with_stats=True)
    ref = oracle.oracle_trace(rays, org, v, f, c, r, H, mode=oracle.MODE_REF_BVH, norm=oracle.NORM_SSE_TABLE)
    # the reference's unpadded slab test may cull a hit that is closer by an ulp; everything else is exact
    diff = np.nonzero(got["tri"] != ref["tri"])[0]
    assert diff.size <= 8, diff.size
    assert np.all(got["range"][diff] <= ref["range"][diff] + 1e-4)
    same = got["tri"] == ref["tri"]
    _assert_bits(got["range"][same], ref["range"][same], "range")
    _assert_bits(got["endrem"][same], ref["endrem"][same], "endrem")
    assert np.array_equal(got["endcolors"][same], ref["endcolors"][same])
    assert got["stats"]["n_hits"] == int((got["tri"] >= 0).sum())
    assert got["stats"]["stack_overflows"] == 0


def test_ctrace_edge_cases(oracle):
    from lidar_transfer_amd.raytracer import C_Trace
    org = np.zeros(3, np.float32)
    rays = create_rays(3, -25, 4, 8)
    # empty mesh: nothing is written
    out = _run_ctrace(rays, org, np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32),
                      np.zeros((0, 3), np.int32), np.zeros(0, np.float32), 4, 8)
    assert not out["range"].any() and (out["tri"] == -1).all()
    # 1, 2, 5 triangles (root-is-leaf and tiny trees)
    for ntri in (1, 2, 5):
        v = np.array([[5, -5, -5], [5, 5, -5], [5, 0, 5]], np.float32)
        vs = np.concatenate([v + np.float32(k) * np.array([1, 0, 0], np.float32) for k in range(ntri)])
        fs = np.arange(3 * ntri, dtype=np.int32).reshape(-1, 3)
        cs = np.tile(np.array([[7, 8, 40]], np.int32), (3 * ntri, 1))
        rm = np.linspace(0.1, 0.9, 3 * ntri).astype(np.float32)
        rr = np.array([[1, 0, 0], [1, 0.1, 0.1], [-1, 0, 0], [1, 0.9, 0.9]], np.float32)
        got = _run_ctrace(rr, org, vs, fs, cs, rm, 2, 2)
        ref = oracle.oracle_trace(rr, org, vs, fs, cs, rm, 2, mode=oracle.MODE_BRUTE, norm=oracle.NORM_SSE_TABLE)
        for k in ("tri", "endcolors", "range", "endrem", "endpoints"):
            _assert_bits(got[k], ref[k], f"{k} ntri={ntri}")
    # outputs are untouched for misses (RayTracer.cpp:73)
    n = 4
    ep = np.full(3 * n, 9, np.float32); ec = np.full(3 * n, 9, np.int32)
    rg = np.full(n, 9, np.float32); rm2 = np.full(n, 9, np.float32)
    C_Trace(rr.reshape(-1), org, vs.reshape(-1), fs.reshape(-1), cs.reshape(-1), rm, ep, ec, rg, rm2, 2, 2)
    assert rg[2] == 9 and ec[6] == 9 and ep[6] == 9 and rm2[2] == 9 and rg[0] != 9
    # dtype / contiguity errors mirror Cython's typed memoryviews
    with pytest.raises(ValueError):
        C_Trace(rr.reshape(-1).astype(np.float64), org, vs.reshape(-1), fs.reshape(-1), cs.reshape(-1), rm, ep, ec,
                rg, rm2, 2, 2)
    with pytest.raises(ValueError):
        C_Trace(rr, org, vs.reshape(-1), fs.reshape(-1), cs.reshape(-1), rm, ep, ec, rg, rm2, 2, 2)
    # bad face index -> error status
    bad = fs.copy(); bad[0, 0] = 10 ** 6
    with pytest.raises(RuntimeError):
        C_Trace(rr.reshape(-1), org, vs.reshape(-1), bad.reshape(-1), cs.reshape(-1), rm, ep, ec, rg, rm2, 2, 2)


def test_scene_api_device_resident(oracle):
    import torch
    from lidar_transfer_amd.raytracer import Scene
    dev = torch.device("cuda", 0)
    v, f, c, r = synth_scene(2, 50000)
    H, W = 32, 256
    rays = create_rays(3, -25, H, W)
    sc = Scene(0)
    tv, tf, tc, tr = [torch.from_numpy(x).to(dev) for x in (v, f, c, r)]
    sc.set_mesh(tv, tf, tc, tr)
    bst = sc.build(stats=True)
    assert bst["n_faces"] == f.shape[0] and bst["ms_build"] > 0
    out = sc.trace(torch.from_numpy(rays).to(dev), (0.0, 0.0, 0.0), H, count=True)
    ref = oracle.oracle_trace(rays, np.zeros(3, np.float32), v, f, c, r, H, mode=oracle.MODE_LBVH,
                              norm=oracle.NORM_SSE_TABLE)
    for k in ("tri", "endcolors", "range", "endrem", "endpoints"):
        _assert_bits(out[k].cpu().numpy(), ref[k], k)
    if os.environ.get("LIDARHIP_TRACE") == "binary":
        # identical tree -> identical work counters as the CPU model of the binary structure
        assert out["stats"]["nodes_visited"] == ref["stats"]["nodes_popped"]
        assert out["stats"]["tris_tested"] == ref["stats"]["tris_tested"]
    else:  # 4-wide nodes: fewer, fatter node visits than the binary model
        assert 0 < out["stats"]["nodes_visited"] < ref["stats"]["nodes_popped"]
        assert out["stats"]["tris_tested"] > 0
    assert out["stats"]["stack_overflows"] == 0 and out["stats"]["n_hits"] == int((ref["tri"] >= 0).sum())
    # rebuild with another mesh in the same workspace, then trace twice (determinism)
    v2, f2, c2, r2 = synth_scene(9, 20000)
    t2 = [torch.from_numpy(x).to(dev) for x in (v2, f2, c2, r2)]
    sc.set_mesh(*t2)
    sc.build()
    a = sc.trace(torch.from_numpy(rays).to(dev), (0.0, 0.0, 0.0), H)
    b = sc.trace(torch.from_numpy(rays).to(dev), (0.0, 0.0, 0.0), H)
    torch.cuda.synchronize()
    for k in ("tri", "range", "endrem"):
        assert torch.equal(a[k], b[k])
    ref2 = oracle.oracle_trace(rays, np.zeros(3, np.float32), v2, f2, c2, r2, H, mode=oracle.MODE_BRUTE,
                               norm=oracle.NORM_SSE_TABLE)
    _assert_bits(a["tri"].cpu().numpy(), ref2["tri"], "tri")
    _assert_bits(a["range"].cpu().numpy(), ref2["range"], "range")
    sc.status()
    sc.close()


def test_exact_normalisation_mode(oracle):
    """LT_TRACE_NORM_EXACT: correctly rounded 1/sqrt seed (vendor independent)."""
    import torch
    from lidar_transfer_amd.raytracer import Scene
    dev = torch.device("cuda", 0)
    v, f, c, r = synth_scene(4, 20000)
    H, W = 16, 128
    rays = create_rays(3, -25, H, W) * np.float32(1.7)
    sc = Scene(0)
    sc.set_mesh(*[torch.from_numpy(x).to(dev) for x in (v, f, c, r)])
    sc.build()
    out = sc.trace(torch.from_numpy(rays).to(dev), (0.0, 0.0, 0.0), H, exact_normalize=True)
    ref = oracle.oracle_trace(rays, np.zeros(3, np.float32), v, f, c, r, H, mode=oracle.MODE_BRUTE,
                              norm=oracle.NORM_EXACT)
    for k in ("tri", "endcolors", "range", "endrem", "endpoints"):
        _assert_bits(out[k].cpu().numpy(), ref[k], k)
    sc.close()


def test_amd_normalisation_mode(oracle):
    """LT_TRACE_NORM_AMD: the RSQRTSS seed of an AMD host replayed from the 2 x 4096 table measured on the MI355X
    box's EPYC -- both strategies against the oracle with that table, and, when this very host is an AMD CPU, against
    the REAL reference (oracle/_ref/libref_strict.so, compiled from /root/reference) executing the instruction."""
    import torch
    from lidar_transfer_amd.raytracer import RaySet, Scene
    dev = torch.device("cuda", 0)
    v, f, c, r = synth_scene(4, 20000)
    H, W = 16, 128
    # un-normalised rays of many lengths: every length picks its own table entry
    rays = np.ascontiguousarray(create_rays(3, -25, H, W) *
                                np.random.default_rng(3).uniform(0.5, 4.0, (H * W, 1)).astype(np.float32))
    org = np.zeros(3, np.float32)
    sc = Scene(0)
    sc.set_mesh(*[torch.from_numpy(x).to(dev) for x in (v, f, c, r)])
    sc.build()
    trays = torch.from_numpy(rays).to(dev)
    out_l = sc.trace(trays, (0.0, 0.0, 0.0), H, exact_normalize="amd")
    rs = RaySet(trays, H, exact_normalize="amd")
    out_s = sc.render(rs, (0.0, 0.0, 0.0))
    ref = oracle.oracle_trace(rays, org, v, f, c, r, H, mode=oracle.MODE_BRUTE, norm=oracle.NORM_AMD_TABLE)
    intel = oracle.oracle_trace(rays, org, v, f, c, r, H, mode=oracle.MODE_BRUTE, norm=oracle.NORM_SSE_TABLE)
    assert not np.array_equal(ref["range"].view(np.int32), intel["range"].view(np.int32))  # the modes do differ
    for k in ("tri", "endcolors", "range", "endrem", "endpoints"):
        _assert_bits(out_l[k].cpu().numpy(), ref[k], "lbvh " + k)
        _assert_bits(out_s[k].cpu().numpy(), ref[k], "scatter " + k)
    if "authenticamd" in open("/proc/cpuinfo").read(4096).lower() and oracle.ref_available("strict"):
        real = oracle.ref_trace(rays, org, v, f, c, r, H, kind="strict")
        for k in ("endcolors", "range", "endrem", "endpoints"):
            _assert_bits(out_s[k].cpu().numpy(), real[k].reshape(out_s[k].shape), "scatter vs real reference on this AMD host: " + k)
    rs.close(); sc.close()


def test_bare_lt_ctrace_symbol_on_golden_three_triangles():
    """The 14-parameter drop-in symbol itself (RayTracer.cpp:116-124 signature), called through ctypes on golden F2
    (made by the real reference): outputs bit-identical, misses untouched."""
    import ctypes as C
    from lidar_transfer_amd import _lib
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "f2_three_triangles.npz"))
    lib = _lib.load()
    rays = np.ascontiguousarray(g["rays"], np.float32).reshape(-1)
    org = np.ascontiguousarray(g["origin"], np.float32).reshape(-1)
    v = np.ascontiguousarray(g["verts"], np.float32).reshape(-1)
    f = np.ascontiguousarray(g["faces"], np.int32).reshape(-1)
    c = np.ascontiguousarray(g["colors"], np.int32).reshape(-1)
    r = np.ascontiguousarray(g["rem"], np.float32).reshape(-1)
    H = int(g["H"])
    n = rays.size // 3
    # a sentinel instead of zeros: the reference leaves rays that hit nothing untouched (RayTracer.cpp:73)
    ep = np.full(3 * n, -5.0, np.float32); ec = np.full(3 * n, -5, np.int32)
    rg = np.full(n, -5.0, np.float32); rm = np.full(n, -5.0, np.float32)
    fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)
    rc = lib.lt_ctrace(rays.ctypes.data_as(fp), org.ctypes.data_as(fp), v.ctypes.data_as(fp), f.ctypes.data_as(ip),
                       c.ctypes.data_as(ip), r.ctypes.data_as(fp), n, v.size // 3, f.size // 3, H,
                       ep.ctypes.data_as(fp), ec.ctypes.data_as(ip), rg.ctypes.data_as(fp), rm.ctypes.data_as(fp))
    assert rc == 0, lib.lt_last_error()
    want = g["range"].reshape(-1)
    hit = want > 0
    assert hit.any() and (~hit).any()
    _assert_bits(rg[hit], want[hit], "range")
    _assert_bits(rm[hit], g["endrem"].reshape(-1)[hit], "endrem")
    _assert_bits(ec.reshape(-1, 3)[hit], g["endcolors"].reshape(-1, 3)[hit], "endcolors")
    _assert_bits(ep.reshape(-1, 3)[hit], g["endpoints"].reshape(-1, 3)[hit], "endpoints")
    assert (rg[~hit] == -5.0).all() and (rm[~hit] == -5.0).all() and (ec.reshape(-1, 3)[~hit] == -5).all() \
        and (ep.reshape(-1, 3)[~hit] == -5.0).all()


# ---- golden vectors generated from the REAL reference (tests/golden/make_golden.py) -----------------
import glob
import os

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["f2_three_triangles", "f3_demo_geometry"])
def test_golden_small_scenes_bitexact(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    H, W = int(g["H"]), int(g["W"])
    got = _run_ctrace(g["rays"], g["origin"], g["verts"], g["faces"], g["colors"], g["rem"], H, W)
    _assert_bits(got["range"], g["range"].reshape(-1), "range")
    _assert_bits(got["endrem"], g["endrem"].reshape(-1), "endrem")
    _assert_bits(got["endpoints"], g["endpoints"], "endpoints")
    assert np.array_equal(got["endcolors"], g["endcolors"])


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "f[45]_*.npz"))),
                         ids=lambda p: os.path.basename(p)[:-4])
def test_golden_synthetic_scenes(path):
    """HIP path vs the compiled reference's outputs on seeded scenes.

    Hit triangle and label must be identical and the range bit-identical (stronger than the 1e-4 m
    of the north star), except on rays where the reference itself is tree-dependent: it keeps the
    first-visited triangle among exact t ties and can cull a hit that is closer by an ulp with its
    unpadded slab test (BVH.cpp:41, :59; BBox.cpp:52-100).  Those rays are recognised by
    |t_gpu - t_ref| <= 2 ulp with t_gpu <= t_ref, and they are only allowed on the scene that was
    built with coincident surfaces.
    """
    g = np.load(path)
    v, f, c, r = synth_scene(int(g["seed"]), int(g["ntri"]), allow_overlap=bool(g["overlap"]))
    import hashlib
    assert hashlib.sha256(v.tobytes()).digest() == bytes(g["verts_sha256"]), "synthetic scene generator drifted"
    H, W = int(g["H"]), int(g["W"])
    rays = create_rays(g["fov"][0], g["fov"][1], H, W)
    got = _run_ctrace(rays, g["origin"], v, f, c, r, H, W)
    idx = g["sample_idx"] if "sample_idx" in g else np.arange(H * W)
    tri, rg, lab = got["tri"][idx], got["range"][idx], got["endcolors"][idx, 2]
    assert int((got["tri"] >= 0).sum()) == int(g["n_hits"]) or bool(g["overlap"])
    diff = np.nonzero(tri != g["tri"])[0]
    if not bool(g["overlap"]):
        assert diff.size == 0, f"{diff.size} hit triangles differ from the reference"
        assert np.array_equal(lab, g["label"])
        _assert_bits(rg, g["range"], "range")
        _assert_bits(got["endrem"][idx], g["endrem"], "endrem")
        _assert_bits(got["endpoints"][idx], g["endpoints"], "endpoints")
        if "range_sha256" in g:  # full-image checksums of the 200 k / 1 M triangle cases
            assert hashlib.sha256(got["range"].tobytes()).digest() == bytes(g["range_sha256"])
            assert hashlib.sha256(got["tri"].tobytes()).digest() == bytes(g["tri_sha256"])
            assert hashlib.sha256(got["endcolors"][:, 2].astype(np.int32).tobytes()).digest() == \
                bytes(g["label_sha256"])
    else:
        same = tri == g["tri"]
        _assert_bits(rg[same], g["range"][same], "range")
        assert np.array_equal(lab[same], g["label"][same])
        ulp = np.spacing(np.abs(g["range"][diff]).astype(np.float32))
        assert np.all(rg[diff] <= g["range"][diff]) and np.all(g["range"][diff] - rg[diff] <= 2 * ulp)
        assert diff.size <= 0.02 * idx.size


def test_throw_rays_at_mesh_tuple_matches_reference_layout():
    """fusion.throw_rays_at_mesh returns the reference's 7-tuple (fusion_lidar.py:453-455) and, on the
    golden scene, the reference's images."""
    from lidar_transfer_amd.fusion import MeshVolume, unpack_deform
    g = np.load(os.path.join(GOLD, "f4_2k_16x64.npz"))
    v, f, c, r = synth_scene(int(g["seed"]), int(g["ntri"]))
    H, W = int(g["H"]), int(g["W"])
    rays = create_rays(g["fov"][0], g["fov"][1], H, W)
    vol = MeshVolume(v, f, c.astype(np.uint8), r)  # get_mesh hands out uint8 colours (fusion_lidar.py:422-423)
    ep, rc, verts, colors, faces, rg, rm = vol.throw_rays_at_mesh(rays, np.zeros(3, np.float32), H, W, None)
    assert ep.shape == (H * W, 3) and ep.dtype == np.float32 and rc.shape == (H * W, 3) and rc.dtype == np.int32
    assert rg.shape == (H, W) and rm.shape == (H, W) and verts is v and faces is f
    _assert_bits(rg.reshape(-1), g["range"], "range")
    _assert_bits(rm.reshape(-1), g["endrem"], "endrem")
    assert np.array_equal(rc[:, 2], g["label"])
    assert np.array_equal(vol.last_tri_image.reshape(-1), g["tri"])
    lut = np.arange(300 * 3, dtype=np.float32).reshape(300, 3)
    label_image, proj_color = unpack_deform(rc, lut, H, W)
    assert label_image.shape == (H, W) and proj_color.shape == (H, W, 3)


# ---- the two strategies: LBVH build + quad traversal vs single-origin triangle scatter ---------------------
def _both_strategies(v, f, c, r, rays, origin, H):
    import torch
    from lidar_transfer_amd.raytracer import RaySet, Scene
    dev = torch.device("cuda", 0)
    sc = Scene(0)
    sc.set_mesh(*[torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (v, f, c, r)])
    trays = torch.from_numpy(np.ascontiguousarray(rays)).to(dev)
    rs = RaySet(trays, H)
    a = sc.render(rs, origin, count=True)
    sc.build()
    b = sc.trace(trays, origin, H, count=True)
    torch.cuda.synchronize()
    a = {k: (x.cpu().numpy() if hasattr(x, "cpu") else x) for k, x in a.items()}
    b = {k: (x.cpu().numpy() if hasattr(x, "cpu") else x) for k, x in b.items()}
    rs.close()
    sc.close()
    return a, b


def test_scatter_equals_lbvh_equals_bruteforce_on_adversarial_soup(oracle):
    """Random triangle soup around the sensor: triangles pierced by the vertical axis, edge-on triangles,
    triangles a few centimetres from the origin, huge triangles; rays in arbitrary (non-grid) directions
    including straight up / down and the coordinate axes."""
    rng = np.random.default_rng(42)
    n = 3000
    cen = rng.normal(size=(n, 3)) * rng.choice([0.05, 0.5, 5.0, 40.0], size=(n, 1))
    size = rng.choice([0.01, 0.1, 1.0, 20.0], size=(n, 1, 1))
    tri = cen[:, None, :] + rng.normal(size=(n, 3, 3)) * size
    tri[:200, :, 1] = 0.0                      # edge-on: in the plane y = 0 through the sensor
    tri[200:300, :, 2] = tri[200:300, :1, 2]   # horizontal triangles (many contain the vertical axis)
    tri[300:320] = tri[300:320] * [1, 1, 0] + [0, 0, 3.0]
    v = np.ascontiguousarray(tri.reshape(-1, 3).astype(np.float32))
    f = np.arange(3 * n, dtype=np.int32).reshape(-1, 3)
    c = np.stack([rng.integers(0, 255, 3 * n), rng.integers(0, 255, 3 * n), rng.integers(0, 260, 3 * n)], 1) \
        .astype(np.int32)
    r = rng.uniform(0, 1, 3 * n).astype(np.float32)
    H, W = 8, 512
    rays = rng.normal(size=(H * W, 3)).astype(np.float32)
    rays[:6] = [[0, 0, 1], [0, 0, -1], [1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0]]
    rays[6:200, 1] = 0.0                       # rays inside the plane of the edge-on triangles
    for origin in ((0.0, 0.0, 0.0), (0.3, -0.2, 0.1)):
        a, b = _both_strategies(v, f, c, r, rays, origin, H)
        ref = oracle.oracle_trace(rays, np.asarray(origin, np.float32), v, f, c, r, H, mode=oracle.MODE_BRUTE,
                                  norm=oracle.NORM_SSE_TABLE)
        for k in ("tri", "endcolors", "range", "endrem", "endpoints"):
            _assert_bits(a[k], ref[k], f"scatter {k} origin={origin}")
            _assert_bits(b[k], ref[k], f"lbvh {k} origin={origin}")
        assert a["stats"]["n_hits"] == b["stats"]["n_hits"] == int((ref["tri"] >= 0).sum()) > 100


def _adversarial_soup(rng, n):
    cen = rng.normal(size=(n, 3)) * rng.choice([0.05, 0.5, 5.0, 40.0], size=(n, 1))
    size = rng.choice([0.01, 0.1, 1.0, 20.0], size=(n, 1, 1))
    tri = cen[:, None, :] + rng.normal(size=(n, 3, 3)) * size
    k = n // 15
    tri[:k, :, 1] = 0.0                          # edge-on: in the plane y = 0 through the sensor
    tri[k:2 * k, :, 2] = tri[k:2 * k, :1, 2]     # horizontal triangles (many contain the vertical axis)
    tri[2 * k:3 * k, 1, :2] = tri[2 * k:3 * k, 0, :2]   # vertical walls: two vertices above each other
    tri[3 * k:4 * k, :, 0] = -np.abs(tri[3 * k:4 * k, :, 0])  # behind the sensor: straddle azimuth +-pi
    tri[3 * k:4 * k, 0, 1] = -np.abs(tri[3 * k:4 * k, 0, 1]) - 1e-3
    tri[3 * k:4 * k, 1, 1] = np.abs(tri[3 * k:4 * k, 1, 1]) + 1e-3
    # beyond the reference's initial t = 999999999.f (BVH.cpp:20): must stay misses
    nf = min(8, n - 4 * k)
    far = rng.normal(size=(nf, 1, 3)); far /= np.linalg.norm(far, axis=2, keepdims=True)
    tri[4 * k:4 * k + nf] = far * 3e9 + rng.normal(size=(nf, 3, 3)) * 1e9
    v = np.ascontiguousarray(tri.reshape(-1, 3).astype(np.float32))
    f = np.arange(3 * n, dtype=np.int32).reshape(-1, 3)
    c = rng.integers(0, 256, (3 * n, 3)).astype(np.int32)
    r = rng.uniform(0, 1, 3 * n).astype(np.float32)
    return v, f, c, r


@pytest.mark.parametrize("H,W,fov_up,fov_down,seed", [(64, 2048, 3.0, -25.0, 1), (16, 301, 15.0, -15.0, 2),
                                                      (1, 720, 0.0, -10.0, 3), (32, 1, 10.0, -30.0, 4),
                                                      (128, 1024, 45.0, -45.0, 5), (5, 7, 2.0, -2.0, 6)])
def test_scatter_on_sensor_grids_vs_adversarial_soup(oracle, H, W, fov_up, fov_down, seed):
    """Regular sensor grids are where the scatter strategy selects candidates most tightly (rays sit at the bin
    centres, a triangle only visits the columns / rows whose ray can lie inside its padded angular bounds):
    check that selection against nasty geometry -- slivers, walls, triangles through the vertical axis, across
    the +-pi seam, centimetres from the sensor -- for several grid shapes and origins, bit for bit against the
    brute-force oracle and the LBVH strategy."""
    rng = np.random.default_rng(seed)
    v, f, c, r = _adversarial_soup(rng, 2500)
    rays = create_rays(fov_up, fov_down, H, W)
    for origin in ((0.0, 0.0, 0.0), (0.37, -0.21, 0.13)):
        a, b = _both_strategies(v, f, c, r, rays, origin, H)
        ref = oracle.oracle_trace(rays, np.asarray(origin, np.float32), v, f, c, r, H, mode=oracle.MODE_BRUTE,
                                  norm=oracle.NORM_SSE_TABLE)
        for k in ("tri", "endcolors", "range", "endrem", "endpoints"):
            _assert_bits(a[k], ref[k], f"scatter {k} {H}x{W} origin={origin}")
            _assert_bits(b[k], ref[k], f"lbvh {k} {H}x{W} origin={origin}")
        assert a["stats"]["n_hits"] == int((ref["tri"] >= 0).sum()) > 0


@pytest.mark.parametrize("sign", [1.0, -1.0])
def test_near_vertical_rays_vs_triangles_grazing_the_axis(oracle, sign):
    """Triangles high above / far below the sensor whose edges pass the vertical axis at 1 mm .. 10 cm, under a
    dense cone of near-vertical rays (an arbitrary ray set -- no LiDAR looks there): the azimuth error of the
    float Moller-Trumbore slop scales with slop / rho, not slop / distance, and the azimuth pad must cover it."""
    rng = np.random.default_rng(7 if sign > 0 else 8)
    n = 4000
    z = sign * rng.uniform(5.0, 60.0, (n, 1))
    rho = rng.choice([1e-3, 3e-3, 1e-2, 3e-2, 1e-1], size=(n, 1))
    phi = rng.uniform(-np.pi, np.pi, (n, 1))
    # an edge tangent to the circle of radius rho around the axis, third vertex further out
    tx, ty = -np.sin(phi), np.cos(phi)
    half = rng.uniform(0.05, 2.0, (n, 1))
    p = np.concatenate([rho * np.cos(phi), rho * np.sin(phi)], 1)
    a = np.concatenate([p - half * np.concatenate([tx, ty], 1), z], 1)
    b = np.concatenate([p + half * np.concatenate([tx, ty], 1), z + rng.normal(size=(n, 1)) * 0.2], 1)
    out = rng.uniform(0.1, 3.0, (n, 1))
    c3 = np.concatenate([p * (1 + out / rho), z + rng.normal(size=(n, 1)) * 0.2], 1)
    tri = np.stack([a, b, c3], 1)
    v = np.ascontiguousarray(tri.reshape(-1, 3).astype(np.float32))
    f = np.arange(3 * n, dtype=np.int32).reshape(-1, 3)
    c = rng.integers(0, 256, (3 * n, 3)).astype(np.int32)
    r = rng.uniform(0, 1, 3 * n).astype(np.float32)
    H, W = 16, 1024
    th = rng.uniform(0, 2 * np.pi, H * W)
    # ray offsets from the axis chosen so that at the triangles' heights they land within ~1e-4 .. 0.2 m of it
    off = 10.0 ** rng.uniform(-5.5, -2.0, H * W)
    rays = np.stack([off * np.cos(th), off * np.sin(th), np.full(H * W, sign)], 1).astype(np.float32)
    for origin in ((0.0, 0.0, 0.0), (1e-3, -2e-3, 0.5)):
        a_, b_ = _both_strategies(v, f, c, r, rays, origin, H)
        ref = oracle.oracle_trace(rays, np.asarray(origin, np.float32), v, f, c, r, H, mode=oracle.MODE_BRUTE,
                                  norm=oracle.NORM_SSE_TABLE)
        for k in ("tri", "endcolors", "range", "endrem", "endpoints"):
            _assert_bits(a_[k], ref[k], f"scatter {k} origin={origin}")
            _assert_bits(b_[k], ref[k], f"lbvh {k} origin={origin}")
        assert int((ref["tri"] >= 0).sum()) > 1000


@pytest.mark.parametrize("H,W", [(5, 8), (16, 64)])
def test_non_finite_rays_and_origin_are_misses_on_a_deep_tree(oracle, H, W):
    """Zero-length, NaN and unnormalised rays, and a NaN origin, on the adversarial soup (a deep, overlapping
    tree).  fminf / fmaxf drop NaN operands, so a NaN ray used to pass the slab test of the UNUSED child slots of
    the 4-wide nodes and follow their sentinel reference out of the node array (GPU memory fault).  Such rays
    can never be accepted by the triangle test: both strategies must report a miss, the other rays must be
    unaffected."""
    rng = np.random.default_rng(1)
    v, f, c, r = _adversarial_soup(rng, 3000)
    rays = (create_rays(10, -20, H, W) * rng.uniform(0.1, 50.0, (H * W, 1))).astype(np.float32)
    rays[0, 1] = np.nan
    rays[3] = 0.0
    rays[H * W - 1, 2] = np.inf
    origin = (0.7, -1.0, 2.7)
    a, b = _both_strategies(v, f, c, r, rays, origin, H)
    ref = oracle.oracle_trace(rays, np.asarray(origin, np.float32), v, f, c, r, H, mode=oracle.MODE_BRUTE,
                              norm=oracle.NORM_SSE_TABLE)
    for k in ("tri", "endcolors", "range", "endrem", "endpoints"):
        _assert_bits(a[k], ref[k], f"scatter {k}")
        _assert_bits(b[k], ref[k], f"lbvh {k}")
    assert a["tri"][0] == a["tri"][3] == a["tri"][H * W - 1] == -1 and int((ref["tri"] >= 0).sum()) > 0
    a, b = _both_strategies(v, f, c, r, rays, (np.nan, 0.0, 0.0), H)
    assert np.all(a["tri"] == -1) and np.all(b["tri"] == -1) and np.all(a["range"] == 0) and np.all(b["range"] == 0)


def test_broken_meshes_non_finite_and_huge_vertices(oracle):
    """A mesh with NaN / infinite vertices, vertices flung 1e19 ... 1e30 m away (the sliver towards such a vertex
    CAN be hit: only one factor of every product is huge) and degenerate faces.  The scatter strategy's angular
    bounds overflow on such triangles (x^2 > FLT_MAX) and used to drop them; now they are tested against every
    ray, and triangles with a non-finite vertex -- which the triangle test can never accept -- are skipped."""
    rng = np.random.default_rng(5)
    v, f, c, r = synth_scene(4, 20000)
    v = v.copy(); f = f.copy()
    vals = [np.nan, np.inf, -np.inf, 1e30, -1e30, 1e19, -3e18, 1e-30, 5e17]
    for k, val in enumerate(vals * 3):
        v[rng.integers(0, v.shape[0]), k % 3] = val
    fd = rng.integers(0, f.shape[0], 50)
    f[fd, 1] = f[fd, 0]
    rays = create_rays(3, -25, 16, 128)
    for origin in ((0.0, 0.0, 0.0), (0.3, -0.2, 0.1)):
        a, b = _both_strategies(v, f, c, r, rays, origin, 16)
        ref = oracle.oracle_trace(rays, np.asarray(origin, np.float32), v, f, c, r, 16, mode=oracle.MODE_BRUTE,
                                  norm=oracle.NORM_SSE_TABLE)
        for k in ("tri", "endcolors", "range", "endrem", "endpoints"):
            _assert_bits(a[k], ref[k], f"scatter {k} origin={origin}")
            _assert_bits(b[k], ref[k], f"lbvh {k} origin={origin}")
    hit_faces = set(ref["tri"][ref["tri"] >= 0].tolist())
    huge = np.nonzero(np.abs(np.nan_to_num(v, nan=0.0, posinf=0.0, neginf=0.0)).max(axis=1) > 1e17)[0]
    assert any(set(f[t].tolist()) & set(huge.tolist()) for t in hit_faces), "no sliver towards a far vertex was hit"


def test_label_image_output_is_colour_channel_two():
    """LT_TRACE_LABEL_IMAGE: the colour output shrinks to the [n_rays] semantic-label image that `deform` unpacks
    (`label_image = ray_colors[:, :, 2]`, laserscan.py:912) -- in both strategies and in the batch call; everything
    else is unchanged."""
    import torch
    from lidar_transfer_amd.raytracer import RaySet, Scene
    dev = torch.device("cuda", 0)
    v, f, c, r = synth_scene(21, 30000)
    rays = torch.from_numpy(create_rays(3.0, -25.0, 32, 256)).to(dev)
    sc = Scene(0)
    sc.set_mesh(*[torch.from_numpy(x).to(dev) for x in (v, f, c, r)])
    rs, rs2 = RaySet(rays, 32), RaySet(rays, 32)
    full = sc.render(rs, (0.2, 0.1, 0.0))
    lab = sc.render(rs, (0.2, 0.1, 0.0), label_image=True)
    sc.build()
    lab_b = sc.trace(rays, (0.2, 0.1, 0.0), 32, label_image=True)
    lab_c = Scene.render_batch([sc], [rs2], [(0.2, 0.1, 0.0)], label_image=True)[0]
    for o in (lab, lab_b, lab_c):
        assert o["endcolors"].shape == (32 * 256,)
        assert torch.equal(o["endcolors"], full["endcolors"][:, 2])
        for k in ("tri", "range", "endrem", "endpoints"):
            assert torch.equal(o[k].view(torch.int32), full[k].view(torch.int32))
    assert int((lab["endcolors"] != 0).sum()) > 1000
    rs.close(); rs2.close(); sc.close()


def test_render_batch_equals_separate_renders():
    """lt_scene_render_batch_dev: several scans -- different meshes (one of them empty, one low-poly whose triangles
    all go through the big-triangle queue), different ray models and origins -- with three launches for all of
    them; every image must be what a separate lt_scene_render_dev call produces, bit for bit."""
    import torch
    from lidar_transfer_amd.raytracer import RaySet, Scene
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(3)
    meshes = [synth_scene(11, 60000), synth_scene(12, 3000), _adversarial_soup(rng, 2500),
              (np.zeros((3, 3), np.float32), np.zeros((0, 3), np.int32), np.zeros((3, 3), np.int32),
               np.zeros(3, np.float32)),
              synth_scene(13, 250000)]
    models = [(3.0, -25.0, 64, 512), (10.0, -30.0, 32, 301), (15.0, -15.0, 16, 1024), (3.0, -25.0, 8, 64),
              (3.0, -25.0, 64, 2048)]
    origins = [(0.0, 0.0, 0.0), (1.5, -2.25, 0.4), (0.3, -0.2, 0.1), (0.0, 0.0, 0.0), (-4.0, 3.0, 0.2)]
    scenes, raysets, single = [], [], []
    for (v, f, c, r), (fu, fd, H, W), org in zip(meshes, models, origins):
        sc = Scene(0)
        sc.set_mesh(*[torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (v, f, c, r)])
        rs = RaySet(torch.from_numpy(create_rays(fu, fd, H, W)).to(dev), H)
        single.append({k: t.clone() for k, t in sc.render(rs, org).items()})
        scenes.append(sc)
        raysets.append(rs)
    outs = Scene.render_batch(scenes, raysets, origins)
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(single, outs)):
        for k in ("tri", "endcolors", "range", "endrem", "endpoints"):
            assert torch.equal(a[k].view(torch.int32), b[k].view(torch.int32)), f"scan {i}: {k} differs"
    assert int((outs[0]["tri"] >= 0).sum()) > 1000 and int((outs[3]["tri"] >= 0).sum()) == 0
    # a second batch on the same objects (queues and cells were re-armed), in another order and size
    outs2 = Scene.render_batch(scenes[::-1][:3], raysets[::-1][:3], origins[::-1][:3])
    for a, b in zip(single[::-1][:3], outs2):
        assert torch.equal(a["range"].view(torch.int32), b["range"].view(torch.int32))
    with pytest.raises(RuntimeError):
        Scene.render_batch([scenes[0], scenes[0]], [raysets[0], raysets[1]], origins[:2])   # one scene, two scans
    for sc in scenes:
        sc.status()
    for rs in raysets:
        rs.close()
    for sc in scenes:
        sc.close()


def test_one_rayset_shared_by_scenes_and_streams():
    """A ray set is read-only during renders (the z-min image of a scan in flight belongs to its scene): ONE ray
    set must serve several scenes in one batch call and renders running concurrently on different streams, each
    image identical to the one a private ray set gives."""
    import torch
    from lidar_transfer_amd.raytracer import RaySet, Scene
    dev = torch.device("cuda", 0)
    H, W = 32, 1024
    rays = torch.from_numpy(create_rays(3.0, -25.0, H, W)).to(dev)
    shared = RaySet(rays, H)
    torch.cuda.synchronize()
    meshes = [synth_scene(40 + i, 40000 + 15000 * i) for i in range(6)]
    origins = [(0.1 * i, -0.2 * i, 0.05 * i) for i in range(6)]
    scenes, want = [], []
    for (v, f, c, r), org in zip(meshes, origins):
        sc = Scene(0)
        sc.set_mesh(*[torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (v, f, c, r)])
        private = RaySet(rays, H)
        want.append({k: t.clone() for k, t in sc.render(private, org).items()})
        torch.cuda.synchronize()
        private.close()
        scenes.append(sc)
    # (a) one batch call, the same ray set six times
    outs = Scene.render_batch(scenes, [shared] * 6, origins)
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(want, outs)):
        for k in ("tri", "endcolors", "range", "endrem", "endpoints"):
            assert torch.equal(a[k].view(torch.int32), b[k].view(torch.int32)), f"batch, scan {i}: {k} differs"
    # (b) single renders of all scenes in flight at once on their own streams, several rounds
    streams = [torch.cuda.Stream(dev) for _ in scenes]
    bufs = [sc.alloc_outputs(H * W) for sc in scenes]
    for _ in range(5):
        for sc, st, org, o in zip(scenes, streams, origins, bufs):
            sc.render(shared, org, out=o, stream=st)
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(want, bufs)):
        for k in ("tri", "endcolors", "range", "endrem", "endpoints"):
            assert torch.equal(a[k].view(torch.int32), b[k].view(torch.int32)), f"streams, scan {i}: {k} differs"
    assert int((bufs[0]["tri"] >= 0).sum()) > 1000
    # (c) one scene, ray sets of different sizes one after the other (its cells grow and stay armed)
    small = RaySet(torch.from_numpy(create_rays(3.0, -25.0, 8, 128)).to(dev), 8)
    big = RaySet(torch.from_numpy(create_rays(3.0, -25.0, 64, 2048)).to(dev), 64)
    a1 = {k: t.clone() for k, t in scenes[0].render(small, origins[0]).items()}
    b1 = {k: t.clone() for k, t in scenes[0].render(big, origins[0]).items()}
    a2 = scenes[0].render(small, origins[0])
    b2 = scenes[0].render(big, origins[0])
    for x, y in ((a1, a2), (b1, b2)):
        for k in ("tri", "range"):
            assert torch.equal(x[k].view(torch.int32), y[k].view(torch.int32))
    torch.cuda.synchronize()
    for rs in (shared, small, big):
        rs.close()
    for sc in scenes:
        sc.close()


def test_scan_pipeline_equals_single_renders():
    """lidar_transfer_amd.pipeline.ScanPipeline (the reference's batch loop body, kept fed): 37 scans -- meshes of
    different sizes, one of them empty, moving origin -- submitted one after the other, rendered in batches of 8 with
    two batches in flight; every image must be the one a lone Scene.render gives."""
    import torch
    from lidar_transfer_amd.pipeline import ScanPipeline
    from lidar_transfer_amd.raytracer import RaySet, Scene
    dev = torch.device("cuda", 0)
    H, W = 32, 512
    rays = torch.from_numpy(create_rays(3.0, -25.0, H, W)).to(dev)
    base = [synth_scene(60 + i, 20000 + 9000 * i) for i in range(5)]
    base.append((np.zeros((3, 3), np.float32), np.zeros((0, 3), np.int32), np.zeros((3, 3), np.int32),
                 np.zeros(3, np.float32)))
    meshes = [[torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in m] for m in base]
    n = 37
    origins = [(0.05 * k, -0.03 * k, 0.01 * (k % 5)) for k in range(n)]
    ranges = torch.full((n, H * W), -7.0, dtype=torch.float32, device=dev)
    labels = torch.full((n, H * W), -7, dtype=torch.int32, device=dev)
    tris = torch.full((n, H * W), -7, dtype=torch.int32, device=dev)
    with ScanPipeline(rays, H, batch=8, in_flight=2) as pipe:
        for k in range(n):
            pipe.submit(*meshes[k % len(meshes)], origins[k], range_out=ranges[k], label_out=labels[k], tri_out=tris[k])
        pipe.flush()
        assert pipe.n_submitted == n
        pipe.status()
        # a second round through the same pool (slots, cells and queues re-armed), other batch geometry
        r2 = torch.empty((5, H * W), dtype=torch.float32, device=dev)
        for k in range(5):
            pipe.submit(*meshes[k], origins[k], range_out=r2[k])
        pipe.flush()
    sc = Scene(0)
    rs = RaySet(rays, H)
    for k in range(n):
        sc.set_mesh(*meshes[k % len(meshes)])
        want = sc.render(rs, origins[k], label_image=True)
        assert torch.equal(want["range"].view(torch.int32), ranges[k].view(torch.int32)), f"scan {k}: range"
        assert torch.equal(want["endcolors"], labels[k]), f"scan {k}: label"
        assert torch.equal(want["tri"], tris[k]), f"scan {k}: tri"
        if k < 5:
            assert torch.equal(want["range"].view(torch.int32), r2[k].view(torch.int32))
    assert int((tris[0] >= 0).sum()) > 1000 and int((tris[5] >= 0).sum()) == 0
    rs.close(); sc.close()


def test_scan_pipeline_holds_meshes_until_their_batch_completed():
    """The caller drops its mesh tensors right after submit() and immediately scribbles over freshly allocated
    tensors of the same sizes (torch's caching allocator hands freed blocks straight back to the allocating stream):
    the pipeline must keep every mesh referenced until the side stream has finished the batch that reads it."""
    import torch
    from lidar_transfer_amd.pipeline import ScanPipeline
    from lidar_transfer_amd.raytracer import RaySet, Scene
    dev = torch.device("cuda", 0)
    H, W = 32, 512
    rays = torch.from_numpy(create_rays(3.0, -25.0, H, W)).to(dev)
    base = [synth_scene(160 + i, 150000) for i in range(3)]
    n = 40
    ranges = torch.zeros((n, H * W), dtype=torch.float32, device=dev)
    tris = torch.zeros((n, H * W), dtype=torch.int32, device=dev)
    with ScanPipeline(rays, H, batch=8, in_flight=2) as pipe:
        for k in range(n):
            mesh = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in base[k % 3]]
            pipe.submit(*mesh, (0.0, 0.0, 0.0), range_out=ranges[k], tri_out=tris[k])
            sizes = [(t.shape, t.dtype) for t in mesh]
            del mesh
            junk = [torch.full(sh, 1e30 if dt == torch.float32 else 0x7fffffff, dtype=dt, device=dev) for sh, dt in sizes]
            del junk
        pipe.flush()
        pipe.status()
    sc, rs = Scene(0), RaySet(rays, H)
    for k in range(3):
        sc.set_mesh(*[torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in base[k]])
        want = sc.render(rs, (0.0, 0.0, 0.0))
        for kk in range(k, n, 3):
            assert torch.equal(want["range"].view(torch.int32), ranges[kk].view(torch.int32)), f"scan {kk}"
            assert torch.equal(want["tri"], tris[kk]), f"scan {kk}"
    rs.close(); sc.close()


def test_handles_release_their_device_memory():
    """Scene / ray-set handles own device memory (LBVH workspace, z-min cells, queues, bin grid): creating, using and
    destroying them in a loop must leave the free device memory where it was."""
    import torch
    from lidar_transfer_amd.raytracer import RaySet, Scene
    dev = torch.device("cuda", 0)
    v, f, c, r = synth_scene(3, 30000)
    mesh = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (v, f, c, r)]
    rays = torch.from_numpy(create_rays(3.0, -25.0, 32, 512)).to(dev)

    def cycle():
        sc = Scene(0)
        rs = RaySet(rays, 32)
        sc.set_mesh(*mesh)
        a = sc.render(rs, (0.0, 0.0, 0.0))
        sc.build()
        b = sc.trace(rays, (0.0, 0.0, 0.0), 32)
        assert torch.equal(a["tri"], b["tri"])
        del a, b
        rs.close()
        sc.close()

    for _ in range(3):
        cycle()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free0 = torch.cuda.mem_get_info(dev)[0]
    for _ in range(40):
        cycle()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free1 = torch.cuda.mem_get_info(dev)[0]
    assert free0 - free1 < 8 << 20, f"{(free0 - free1) >> 20} MiB of device memory lost in 40 create/destroy cycles"


@pytest.mark.parametrize("wl,seed,origin", [("C1", 5, (0.0, 0.0, 0.0)), ("C2", 2, (0.0, 0.0, 0.0)),
                                            ("C3", 1, (1.5, -2.25, 0.4)), ("C4", 0, (0.0, 0.0, 0.0))])
def test_scatter_equals_lbvh_at_baseline_sizes(wl, seed, origin):
    """BASELINE.json configurations at full size (0.2 M - 2.5 M triangles, up to 128x2048 rays): the two
    independent strategies must produce identical images, bit for bit."""
    from lidar_transfer_amd.synth import WORKLOADS
    w = WORKLOADS[wl]
    v, f, c, r = synth_scene(seed, w["tris"])
    rays = create_rays(w["fov_up"], w["fov_down"], w["H"], w["W"])
    a, b = _both_strategies(v, f, c, r, rays, origin, w["H"])
    for k in ("tri", "endcolors", "range", "endrem", "endpoints"):
        _assert_bits(a[k], b[k], f"{wl} {k}")
    hits = int((a["tri"] >= 0).sum())
    assert hits == a["stats"]["n_hits"] == b["stats"]["n_hits"] and hits > 0.5 * w["H"] * w["W"]
    assert b["stats"]["stack_overflows"] == 0
    # size-independent properties: a hit point lies on its ray at the reported range, inside the scene bounds
    ok = a["tri"] >= 0
    d = a["endpoints"][ok] - np.asarray(origin, np.float32)
    assert np.allclose(np.linalg.norm(d, axis=1), a["range"][ok], rtol=2e-6)
    assert np.all(np.abs(a["endpoints"][ok]) <= np.abs(v).max() + 1e-3)
    assert np.all(a["range"][~ok] == 0) and np.all(a["endcolors"][~ok] == 0)


def test_binary_one_ray_per_lane_path_in_subprocess(tmp_path):
    """LIDARHIP_TRACE=binary (binary nodes, one ray per lane: the A/B twin of the quad traversal) is selected
    once per process, so it is exercised in a child process: same images as the default path."""
    import subprocess
    import sys
    code = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
from lidar_transfer_amd.laserscan import create_rays
from lidar_transfer_amd.raytracer import Scene
from lidar_transfer_amd.synth import synth_scene
dev = torch.device("cuda", 0)
v, f, c, r = synth_scene(6, 40000)
rays = create_rays(3, -25, 32, 256)
sc = Scene(0)
sc.set_mesh(*[torch.from_numpy(x).to(dev) for x in (v, f, c, r)])
sc.build()
o = sc.trace(torch.from_numpy(rays).to(dev), (0.0, 0.0, 0.0), 32, count=True)
np.savez(sys.argv[1], tri=o["tri"].cpu().numpy(), range=o["range"].cpu().numpy(), nodes=o["stats"]["nodes_visited"])
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for mode in ("binary", "quad"):
        env = dict(os.environ)
        env.pop("LIDARHIP_TRACE", None)
        if mode == "binary":
            env["LIDARHIP_TRACE"] = "binary"
        p = str(tmp_path / f"{mode}.npz")
        subprocess.run([sys.executable, "-c", code, p], check=True, env=env, timeout=300)
        outs[mode] = np.load(p)
    assert np.array_equal(outs["binary"]["tri"], outs["quad"]["tri"])
    _assert_bits(outs["binary"]["range"], outs["quad"]["range"], "range")
    assert int(outs["binary"]["nodes"]) > int(outs["quad"]["nodes"]) > 0   # binary visits ~2x as many (thinner) nodes


def test_wide_addressing_variants_in_subprocess(tmp_path):
    """The scatter kernels have a 64-bit addressing variant for arrays of 4 GB and more (> 357 M triangles); force
    it on an ordinary scene (LIDARHIP_FORCE_WIDE=1, read once per process, hence the subprocess) and compare with
    the 32-bit variant of this process."""
    import subprocess
    import sys
    import torch
    from lidar_transfer_amd.raytracer import RaySet, Scene
    v, f, c, r = synth_scene(31, 40000)
    rays = create_rays(3.0, -25.0, 32, 512)
    dev = torch.device("cuda", 0)
    sc = Scene(0)
    sc.set_mesh(*[torch.from_numpy(x).to(dev) for x in (v, f, c, r)])
    rs = RaySet(torch.from_numpy(rays).to(dev), 32)
    a = sc.render(rs, (0.1, 0.2, 0.0))
    np.savez(tmp_path / "narrow.npz", **{k: t.cpu().numpy() for k, t in a.items()})
    code = f"""
import sys, numpy as np, torch
sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})
from lidar_transfer_amd.raytracer import RaySet, Scene
from lidar_transfer_amd.laserscan import create_rays
from lidar_transfer_amd.synth import synth_scene
v, f, c, r = synth_scene(31, 40000)
dev = torch.device("cuda", 0)
sc = Scene(0); sc.set_mesh(*[torch.from_numpy(x).to(dev) for x in (v, f, c, r)])
rs = RaySet(torch.from_numpy(create_rays(3.0, -25.0, 32, 512)).to(dev), 32)
a = sc.render(rs, (0.1, 0.2, 0.0), count=True)
b = Scene.render_batch([sc], [rs], [(0.1, 0.2, 0.0)])[0]
g = np.load({str(tmp_path / "narrow.npz")!r})
for o in (a, b):
    for k in ("tri", "range", "endpoints", "endcolors", "endrem"):
        assert np.array_equal(o[k].cpu().numpy().view(np.int32), g[k].view(np.int32)), k
print("WIDE_OK", int((a["tri"] >= 0).sum()))
"""
    env = dict(os.environ, LIDARHIP_FORCE_WIDE="1")
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert res.returncode == 0 and "WIDE_OK" in res.stdout, res.stdout + res.stderr
    rs.close()
    sc.close()


def test_five_instruction_reciprocal_is_the_ieee_division_for_every_float():
    """The triangle test multiplies by inv_a = 1.0f / a (Triangle.h:35).  The kernels compute it with two Newton
    steps on v_rcp_f32 for 2^-64 <= |a| <= 2^64 (lt_internal.h: lt_rcp_ieee) instead of the 10-instruction IEEE
    division sequence; this compares the two for all 2^32 bit patterns."""
    import ctypes as C
    from lidar_transfer_amd import _lib
    lib = _lib.load()
    bad, fast = C.c_ulonglong(1), C.c_ulonglong(0)
    lib.lt_debug_verify_rcp.argtypes = [C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    lib.lt_debug_verify_rcp.restype = C.c_int
    assert lib.lt_debug_verify_rcp(C.byref(bad), C.byref(fast)) == 0
    assert bad.value == 0, f"{bad.value} of 2^32 floats differ"
    assert fast.value == 2 * 128 * (1 << 23) + 2   # 128 binades of each sign, plus +-2^64


def test_tiny_far_triangles_are_not_taken_for_pierced(oracle):
    """Marching cubes emits micrometre-sized triangles where the field is ~0 at a grid point.  Far from the sensor
    their three sub-areas are all of the order of the rounding tolerance -- neither clearly of mixed sign nor clearly
    zero -- and the angular bounds used to fall back to "every azimuth" (70 000 candidate bins per triangle).
    Parity with the brute-force oracle, and the candidate count stays that of a small triangle."""
    rng = np.random.default_rng(21)
    n = 3000
    rho = rng.uniform(5.0, 60.0, (n, 1))
    phi = rng.uniform(-np.pi, np.pi, (n, 1))
    cen = np.concatenate([rho * np.cos(phi), rho * np.sin(phi), rng.uniform(-2.0, 0.5, (n, 1))], 1)
    size = 10.0 ** rng.uniform(-5.5, -3.0, (n, 1, 1))           # 3 um .. 1 mm
    tri = cen[:, None, :] + rng.normal(size=(n, 3, 3)) * size
    tri[: n // 3] = cen[: n // 3, None, :] + np.eye(3)[None] * size[: n // 3]   # the axis-aligned corner triangles of MC
    big = synth_scene(2, 4000)                                    # plus an ordinary scene behind them
    v = np.ascontiguousarray(np.concatenate([tri.reshape(-1, 3).astype(np.float32), big[0]]))
    f = np.ascontiguousarray(np.concatenate([np.arange(3 * n, dtype=np.int32).reshape(-1, 3), big[1] + 3 * n]))
    c = np.ascontiguousarray(np.concatenate([rng.integers(0, 256, (3 * n, 3)).astype(np.int32), big[2]]))
    r = np.ascontiguousarray(np.concatenate([rng.uniform(0, 1, 3 * n).astype(np.float32), big[3]]))
    H, W = 64, 2048
    rays = create_rays(3.0, -25.0, H, W)
    a, b = _both_strategies(v, f, c, r, rays, (0.0, 0.0, 0.0), H)
    ref = oracle.oracle_trace(rays, np.zeros(3, np.float32), v, f, c, r, H, mode=oracle.MODE_BRUTE, norm=oracle.NORM_SSE_TABLE)
    for k in ("tri", "endcolors", "range", "endrem", "endpoints"):
        _assert_bits(a[k], ref[k], "scatter " + k)
        _assert_bits(b[k], ref[k], "lbvh " + k)
    # the tiny triangles alone: a handful of candidate bins each (it was the whole circle: ~70 000)
    nt = 3 * n
    a2, _ = _both_strategies(v[:nt], f[:n], c[:nt], r[:nt], rays, (0.0, 0.0, 0.0), H)
    assert a2["stats"]["nodes_visited"] < 8 * n, a2["stats"]["nodes_visited"] / n


def test_lbvh_on_degenerate_morton_trees(oracle):
    """The LBVH build's queue of wide nodes (k_hierarchy4 -> k_hierarchy4_big) on trees that are not balanced: thousands
    of identical triangles (one Morton key: the tree is decided by the index tie-break alone and every upper node is
    wide), and a staircase of clusters at x = 2^-k, 600 copies each (a chain of wide nodes, one per key bit)."""
    rng = np.random.default_rng(77)
    one = np.array([[4.0, -1.0, -1.0], [4.0, 1.0, -1.0], [4.0, 0.0, 1.5]])
    same = np.repeat(one[None], 5000, axis=0)
    stairs = []
    for k in range(14):
        base = one * [1.0, 0.2, 0.2] + [30.0 * 2.0 ** -k, 0.0, 0.0]
        stairs.append(np.repeat(base[None], 600, axis=0) + rng.normal(size=(600, 1, 3)) * 1e-4)
    for tri in (same, np.concatenate(stairs), np.concatenate([same] + stairs)):
        n = len(tri)
        v = np.ascontiguousarray(tri.reshape(-1, 3).astype(np.float32))
        f = np.arange(3 * n, dtype=np.int32).reshape(-1, 3)
        c = rng.integers(0, 256, (3 * n, 3)).astype(np.int32)
        r = rng.uniform(0, 1, 3 * n).astype(np.float32)
        H, W = 8, 128
        spread = np.where(np.arange(H * W)[:, None] % 2 == 0, 0.25, 0.03)   # (the stairs subtend +-0.05 rad)
        rays = (np.array([1.0, 0.0, 0.0]) + rng.normal(size=(H * W, 3)) * spread * [0.0, 1.0, 1.0]).astype(np.float32)
        a, b = _both_strategies(v, f, c, r, rays, (0.0, 0.0, 0.0), H)
        ref = oracle.oracle_trace(rays, np.zeros(3, np.float32), v, f, c, r, H, mode=oracle.MODE_BRUTE,
                                  norm=oracle.NORM_SSE_TABLE)
        for k in ("tri", "endcolors", "range", "endrem", "endpoints"):
            _assert_bits(a[k], ref[k], f"scatter {k} n={n}")
            _assert_bits(b[k], ref[k], f"lbvh {k} n={n}")
        assert int((ref["tri"] >= 0).sum()) > 100


def test_lbvh_node_array_is_pinned():
    """The 4-wide node array Scene.build writes on five seeded meshes (300 ... 1 M triangles), by SHA-256 (digests recorded
    with round 5's three-kernels-per-digit sort and segment-tree hierarchy, tools/nodes4_digest.py): the sort is stable and
    the hierarchy deterministic, so ANY correct re-implementation of either (round 6: the one-pass-per-digit sort) must
    reproduce the array bit for bit -- including which nodes are never written."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import nodes4_digest
    want = {"s0_300": "c8bf9696b448e94b8dc430eb52d5cf92e5302685ab91e8eead9a87d04cf33c8c",
            "s1_5000": "050f4c60ecd95455e9e499ef8b16aee968f124f95211c71cc3d2bc2a2e036f49",
            "s0_20000": "4861dd6d0265c2b8e6cb07a85cbc585e5bf77149fadfdb4c3e19912a574ed83e",
            "s2_200000": "2bb99f9066fb393c76f5a6e89abf28107704d146479a964f6d2edeee3f7845f5",
            "s0_1000000": "59e63c6b0565fc31ee67834ad4543ba88ed0c8b7bf9cb6d69e6e5634960ef79b"}
    assert nodes4_digest.digests() == want
