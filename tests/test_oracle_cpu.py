"""CPU suite (no GPU): pins the oracle, the host logic and the C-ABI surface.

* the C restatement (oracle/lt_oracle.c) against the REAL reference compiled into oracle/_ref
  (when that build is present) and against the golden vectors generated from it;
* the brute-force / LBVH-model definitions the HIP path is held to, against the goldens;
* host mirrors (create_rays, synthetic inputs) against goldens;
* liblidarhip.so loads and exports every symbol include/lidarhip.h declares (no compute calls).
"""
import glob
import hashlib
import os
import re
import sys

import numpy as np
import pytest

from lidar_transfer_amd.laserscan import create_rays
from lidar_transfer_amd.synth import soup, synth_scene

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(HERE, "golden")


def _bits_equal(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.int32), b.view(np.int32))


def _host_rsqrt_is_table(oracle):
    """True when this host's RSQRTSS equals the table the goldens were made with (Intel CPUs)."""
    rng = np.random.default_rng(0)
    r = (rng.normal(size=(4096, 3)) * rng.uniform(1e-3, 1e3, (4096, 1))).astype(np.float32)
    return _bits_equal(oracle.normalize_rays(r, oracle.NORM_SSE), oracle.normalize_rays(r, oracle.NORM_SSE_TABLE))


# ---- restatement == compiled reference ------------------------------------------------------------------
@pytest.mark.parametrize("ntri,H,W,seed,org", [(2000, 16, 64, 0, (0, 0, 0)), (50000, 32, 128, 1, (0, 0, 0)),
                                               (30000, 16, 96, 5, (1.5, -2.25, 0.4)),
                                               (200000, 64, 256, 2, (0, 0, 0))])
def test_restatement_equals_compiled_reference(oracle, ntri, H, W, seed, org, capfd):
    if not oracle.ref_available("strict"):
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    v, f, c, r = synth_scene(seed, ntri)
    sv, sf, sc, sr = soup(v, f, c, r)
    rays = create_rays(3, -25, H, W)
    org = np.asarray(org, np.float32)
    ref = oracle.ref_trace(rays, org, sv, sf, sc, sr, H, kind="strict")
    capfd.readouterr()  # the reference printf()s
    got = oracle.oracle_trace(rays, org, sv, sf, sc, sr, H, mode=oracle.MODE_REF_BVH, norm=oracle.NORM_SSE)
    for k in ("range", "endrem", "endpoints", "endcolors"):
        assert _bits_equal(got[k], ref[k]), k
    hit = got["tri"] >= 0
    assert np.array_equal(got["tri"][hit], ref["endcolors"][hit, 0])  # soup trick == our tri output
    # indexed mesh and soup are the same geometry
    got2 = oracle.oracle_trace(rays, org, v, f, c, r, H, mode=oracle.MODE_REF_BVH, norm=oracle.NORM_SSE)
    assert _bits_equal(got2["range"], got["range"]) and np.array_equal(got2["tri"], got["tri"])


def test_rsqrt_table_replays_this_host_when_intel(oracle):
    import platform
    if "intel" not in (platform.processor() + open("/proc/cpuinfo").read(4096)).lower():
        pytest.skip("table was measured on an Intel CPU")
    assert _host_rsqrt_is_table(oracle)


def test_rsqrt_amd_table_replays_this_host_when_amd(oracle):
    """The 2 x 4096 table measured on the MI355X box's EPYC (oracle/rsqrt_amd_table.h) is what an AMD host's
    RSQRTSS returns (skipped on other vendors -- it runs with the CPU suite on the GPU box, not here)."""
    if "authenticamd" not in open("/proc/cpuinfo").read(4096).lower():
        pytest.skip("table was measured on an AMD CPU")
    rng = np.random.default_rng(1)
    r = (rng.normal(size=(1 << 16, 3)) * 10.0 ** rng.uniform(-6, 6, (1 << 16, 1))).astype(np.float32)
    assert _bits_equal(oracle.normalize_rays(r, oracle.NORM_SSE), oracle.normalize_rays(r, oracle.NORM_AMD_TABLE))


def test_the_two_vendor_tables_differ_only_in_the_last_bits(oracle):
    rng = np.random.default_rng(2)
    r = (rng.normal(size=(4096, 3)) * rng.uniform(1e-3, 1e3, (4096, 1))).astype(np.float32)
    a, b = oracle.normalize_rays(r, oracle.NORM_SSE_TABLE), oracle.normalize_rays(r, oracle.NORM_AMD_TABLE)
    e = oracle.normalize_rays(r, oracle.NORM_EXACT)
    assert not _bits_equal(a, b)                       # vendor specific ...
    assert np.abs(a - b).max() < 3e-7 and np.abs(a - e).max() < 3e-7 and np.abs(b - e).max() < 3e-7  # ... in the last ulps


# ---- restatement vs golden vectors -------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["f2_three_triangles", "f3_demo_geometry"])
def test_restatement_vs_golden_small(oracle, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    got = oracle.oracle_trace(g["rays"], g["origin"], g["verts"], g["faces"], g["colors"], g["rem"], int(g["H"]),
                              mode=oracle.MODE_REF_BVH, norm=oracle.NORM_SSE_TABLE)
    assert _bits_equal(got["range"], g["range"].reshape(-1))
    assert _bits_equal(got["endrem"], g["endrem"].reshape(-1))
    assert _bits_equal(got["endpoints"], g["endpoints"])
    assert np.array_equal(got["endcolors"], g["endcolors"])
    if name == "f2_three_triangles":
        rg = g["range"].reshape(-1)
        assert abs(rg[0] - 2.0) < 1e-6 and rg[2] == 0.0 and rg[3] == 0.0  # small triangle first; misses stay 0
        assert g["endcolors"][0, 2] == 70 and g["endcolors"][1, 2] == 40   # colour of vertex 0 of the hit face


SCENES = sorted(glob.glob(os.path.join(GOLD, "f[45]_*.npz")))


def _load_scene(g):
    v, f, c, r = synth_scene(int(g["seed"]), int(g["ntri"]), allow_overlap=bool(g["overlap"]))
    assert hashlib.sha256(v.tobytes()).digest() == bytes(g["verts_sha256"]), "synthetic scene generator drifted"
    assert hashlib.sha256(f.tobytes()).digest() == bytes(g["faces_sha256"])
    H, W = int(g["H"]), int(g["W"])
    return v, f, c, r, H, W, create_rays(g["fov"][0], g["fov"][1], H, W)


@pytest.mark.parametrize("path", [p for p in SCENES if "1m" not in p and "2m5" not in p], ids=lambda p: os.path.basename(p)[:-4])
def test_restatement_vs_golden_scenes(oracle, path):
    """Reference-BVH restatement reproduces the reference's images bit for bit (incl. its tie order)."""
    g = np.load(path)
    v, f, c, r, H, W, rays = _load_scene(g)
    got = oracle.oracle_trace(rays, g["origin"], v, f, c, r, H, mode=oracle.MODE_REF_BVH, norm=oracle.NORM_SSE_TABLE)
    idx = g["sample_idx"] if "sample_idx" in g else np.arange(H * W)
    assert np.array_equal(got["tri"][idx], g["tri"])
    assert np.array_equal(got["endcolors"][idx, 2], g["label"])
    assert _bits_equal(got["range"][idx], g["range"])
    assert _bits_equal(got["endrem"][idx], g["endrem"])
    assert _bits_equal(got["endpoints"][idx], g["endpoints"])
    assert int((got["tri"] >= 0).sum()) == int(g["n_hits"])


@pytest.mark.parametrize("path", [p for p in SCENES if "1m" not in p and "2m5" not in p and "200k" not in p],
                         ids=lambda p: os.path.basename(p)[:-4])
def test_gpu_definition_vs_golden_scenes(oracle, path):
    """The tree-independent definition the HIP kernels implement (closest accepted triangle, ties to the
    lower face index; here via the LBVH model) equals the reference wherever the reference itself is not
    tree dependent -- everywhere on scenes without coincident surfaces."""
    g = np.load(path)
    v, f, c, r, H, W, rays = _load_scene(g)
    got = oracle.oracle_trace(rays, g["origin"], v, f, c, r, H, mode=oracle.MODE_LBVH, norm=oracle.NORM_SSE_TABLE)
    diff = np.nonzero(got["tri"] != g["tri"])[0]
    if not bool(g["overlap"]):
        assert diff.size == 0
        assert _bits_equal(got["range"], g["range"])
        assert np.array_equal(got["endcolors"][:, 2], g["label"])
    else:
        assert 0 < diff.size <= 0.02 * H * W
        ulp = np.spacing(np.abs(g["range"][diff]).astype(np.float32))
        assert np.all(got["range"][diff] <= g["range"][diff])
        assert np.all(g["range"][diff] - got["range"][diff] <= 2 * ulp)


@pytest.mark.parametrize("ntri,H,W,seed,overlap", [(2000, 16, 64, 0, False), (20000, 16, 64, 3, True),
                                                   (50000, 32, 64, 1, False)])
def test_lbvh_model_equals_bruteforce(oracle, ntri, H, W, seed, overlap):
    v, f, c, r = synth_scene(seed, ntri, allow_overlap=overlap)
    rays = create_rays(3, -25, H, W)
    org = np.zeros(3, np.float32)
    a = oracle.oracle_trace(rays, org, v, f, c, r, H, mode=oracle.MODE_LBVH, norm=oracle.NORM_EXACT)
    b = oracle.oracle_trace(rays, org, v, f, c, r, H, mode=oracle.MODE_BRUTE, norm=oracle.NORM_EXACT)
    for k in ("tri", "range", "endrem", "endpoints", "endcolors"):
        assert _bits_equal(a[k], b[k]), k
    assert a["stats"]["max_stack"] < 32


def test_oracle_edge_cases(oracle):
    org = np.zeros(3, np.float32)
    rays = create_rays(3, -25, 4, 8)
    e = oracle.oracle_trace(rays, org, np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32),
                            np.zeros((0, 3), np.int32), np.zeros(0, np.float32), 4)
    assert not e["range"].any() and (e["tri"] == -1).all()
    # one triangle, ray exactly through a vertex / edge / inside / behind
    v = np.array([[5, -1, -1], [5, 1, -1], [5, 0, 1]], np.float32)
    f = np.array([[0, 1, 2]], np.int32)
    c = np.array([[1, 2, 3]] * 3, np.int32)
    r = np.array([0.3, 0.6, 0.9], np.float32)
    rr = np.array([[1, 0, 0], [-1, 0, 0], [5, -1, -1], [5, 0, -1]], np.float32)
    for mode in (oracle.MODE_REF_BVH, oracle.MODE_BRUTE, oracle.MODE_LBVH):
        o = oracle.oracle_trace(rr, org, v, f, c, r, 1, mode=mode, norm=oracle.NORM_EXACT)
        assert o["tri"][0] == 0 and abs(o["range"][0] - 5.0) < 1e-5
        assert o["tri"][1] == -1
        assert abs(o["endrem"][0] - 0.6) < 1e-6 and tuple(o["endcolors"][0]) == (1, 2, 3)


# ---- host mirrors ---------------------------------------------------------------------------------------------
def test_create_rays_vs_golden():
    g = np.load(os.path.join(GOLD, "f1_create_rays.npz"))
    for name in "abcdef":
        fu, fd, H, W = g[f"{name}_args"]
        r = create_rays(fu, fd, int(H), int(W))
        assert r.dtype == np.float32 and r.shape == (int(H) * int(W), 3) and r.flags["C_CONTIGUOUS"]
        if f"{name}_rays" in g:
            assert _bits_equal(r, g[f"{name}_rays"])
        else:
            assert hashlib.sha256(r.tobytes()).digest() == bytes(g[f"{name}_sha256"])
            assert _bits_equal(r[:64], g[f"{name}_head"]) and _bits_equal(r[::997], g[f"{name}_stride997"])
        # quirk kept: linspace(0, 360, W) makes column 0 and column W-1 the same direction
        rr = r.reshape(int(H), int(W), 3)
        assert np.allclose(rr[:, 0], rr[:, -1], atol=1e-6)
        assert np.all(np.abs(np.linalg.norm(r.astype(np.float64), axis=1) - 1) < 1e-6)


def test_synth_scene_is_deterministic_and_mesh_like():
    a = synth_scene(11, 30000)
    b = synth_scene(11, 30000)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    v, f, c, r = a
    assert v.dtype == np.float32 and f.dtype == np.int32 and c.dtype == np.int32 and r.dtype == np.float32
    assert f.min() >= 0 and f.max() < v.shape[0] and c.shape == (v.shape[0], 3) and r.shape == (v.shape[0],)
    assert 0.8 * 30000 < f.shape[0] < 1.25 * 30000


# ---- C ABI surface ----------------------------------------------------------------------------------------------
def test_library_exports_every_declared_symbol():
    import ctypes
    from lidar_transfer_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "lidarhip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(lt_[a-z_0-9]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    lib = _lib.load()
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} declared in lidarhip.h but not exported"
    assert sorted(_lib.SYMBOLS) == declared
    assert b"gfx950" in lib.lt_version()
    assert ctypes.sizeof(_lib.Stats) == 80  # lt_stats layout (entries_culled added in round 3)


def test_header_compiles_as_plain_c_and_its_structs_match_the_ctypes_mirrors(tmp_path):
    """include/lidarhip.h is the boundary: a C compiler must take it as it is (no C++ / HIP types), and the structs the Python
    side fills (lt_stats, lt_cloud, lt_proj_images incl. round 5's `bnds`) must have the sizes of their ctypes mirrors."""
    import ctypes
    import subprocess
    from lidar_transfer_amd import _lib
    src = tmp_path / "abi.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "lidarhip.h"\nint main(void) {\n'
                   '  printf("%zu %zu %zu %zu %u %u %zu %zu %d\\n", sizeof(lt_stats), sizeof(lt_cloud), sizeof(lt_proj_images),\n'
                   '         offsetof(lt_proj_images, bnds), (unsigned)LT_TSDF_MERGE, (unsigned)LT_TSDF_HOST_MODE,\n'
                   '         sizeof(lt_mm_geometry), offsetof(lt_mm_geometry, status), LT_ABI_VERSION);\n  return 0;\n}\n')
    exe = tmp_path / "abi"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert got[0] == ctypes.sizeof(_lib.Stats) and got[1] == ctypes.sizeof(_lib.Cloud)
    assert got[2] == ctypes.sizeof(_lib.ProjImages) and got[3] == _lib.ProjImages.bnds.offset
    assert got[4] == _lib.LT_TSDF_MERGE and got[5] == _lib.LT_TSDF_HOST_MODE
    assert got[6] == ctypes.sizeof(_lib.MMGeometry) and got[7] == _lib.MMGeometry.status.offset
    # the layout version a caller compiled against THIS header carries is the one the library reports and the binding checks
    assert got[8] == _lib.LT_ABI_VERSION == _lib.load().lt_abi_version()


def test_c_trace_argument_checks_raise_before_any_device_work():
    from lidar_transfer_amd.raytracer import C_Trace
    f32, i32 = np.float32, np.int32
    ok = dict(rays=np.zeros(12, f32), origin=np.zeros(3, f32), verts=np.zeros(9, f32), faces=np.zeros(3, i32),
              colors=np.zeros(9, i32), rem=np.zeros(3, f32), ep=np.zeros(12, f32), ec=np.zeros(12, i32),
              rg=np.zeros(4, f32), rm=np.zeros(4, f32))

    def call(**kw):
        a = dict(ok)
        a.update(kw)
        C_Trace(a["rays"], a["origin"], a["verts"], a["faces"], a["colors"], a["rem"], a["ep"], a["ec"], a["rg"],
                a["rm"], 2, 2)

    with pytest.raises(ValueError):
        call(rays=np.zeros(12, np.float64))
    with pytest.raises(ValueError):
        call(faces=np.zeros(3, np.int64))
    with pytest.raises(ValueError):
        call(verts=np.zeros((3, 3), f32))
    with pytest.raises(ValueError):
        call(rg=np.zeros(8, f32)[::2])
    with pytest.raises(TypeError):
        call(rem=[0.0, 0.0, 0.0])


# ---- projection oracle (oracle/projection.py) pinned against the reference's own arrays ----------------------------
def _f6_case(g, tag, beams):
    return (g[f"{tag}_points"], g[f"{tag}_rem"], g[f"{tag}_label"], list(g[f"{tag}_beam_angles"]) if beams else None,
            int(g["H"]), int(g["W"]), float(g["fov_up"]), float(g["fov_down"]))


@pytest.mark.parametrize("tag,beams", [("f32", False), ("f64", False), ("f64_beams", True)])
@pytest.mark.parametrize("method", ["old", "new"])
def test_projection_restatement_vs_reference_python(tag, beams, method):
    """oracle/projection.py reproduces `do_range_projection` / `do_range_projection_new` + the label projection of
    the reference (golden F6: float32, float64 and beam-snapped clouds with depth-0 points and exact depth ties)."""
    from oracle import projection as op
    g = np.load(os.path.join(GOLD, "f6_range_projection.npz"))
    pts, rem, lab, beam_angles, H, W, fu, fd = _f6_case(g, tag, beams)
    o = op.range_projection(pts, rem, H, W, fu, fd, beam_angles=beam_angles, remove=True, method=method)
    k = f"{tag}_{method}"
    assert np.array_equal(pts[o["kept"]], g[f"{k}_points_kept"])
    assert np.array_equal(o["unproj_range"], g[f"{k}_unproj_range"])
    assert _bits_equal(o["range"], g[f"{k}_proj_range"])
    if method == "old":
        # the winning index may differ only between points of exactly equal depth in one cell: the reference
        # orders them with an unstable argsort (laserscan.py:272), the restatement takes the lower index
        diff = o["index"] != g[f"{k}_proj_idx"]
        assert diff.sum() <= 12
        d = o["unproj_range"]
        assert np.array_equal(d[o["index"][diff]], d[g[f"{k}_proj_idx"][diff]])
        assert np.array_equal(o["remission"][~diff], g[f"{k}_proj_remissions"][~diff])
        assert np.array_equal(o["xyz"], g[f"{k}_proj_xyz"]) and np.array_equal(o["mask"][~diff], g[f"{k}_proj_mask"][~diff])
    else:
        assert np.array_equal(o["remission"], g[f"{k}_proj_remissions"])
        assert np.array_equal(o["index"], g[f"{k}_index"])
        own = o["index"] >= 0
        assert np.array_equal(o["px"][o["index"][own]], g[f"{k}_proj_x"][own])
        assert np.array_equal(o["py"][o["index"][own]], g[f"{k}_proj_y"][own])
        lab_img = op.label_projection(o["index"], lab[o["kept"]])
        assert np.array_equal(lab_img[own], g[f"{k}_label_image"][..., 0][own].astype(np.int32))


@pytest.mark.parametrize("method", ["old", "new"])
def test_projection_restatement_at_baseline_scale_f64(method):
    """... and at BASELINE scale (120 000 points -> 64 x 2048, golden F9) for float64 clouds, by SHA-256."""
    from oracle import projection as op
    from lidar_transfer_amd.synth import synth_cloud
    g = np.load(os.path.join(GOLD, "f9_range_projection_full.npz"))
    H, W, fu, fd = int(g["H"]), int(g["W"]), float(g["fov_up"]), float(g["fov_down"])
    pts, rem, lab = synth_cloud(int(g["seed"]), int(g["n_points"]), dtype=np.float64, fov_up=fu, fov_down=fd)
    if method == "new":
        pts[1000:1100] = pts[5000:5100]
    pts[7] = 0
    k = f"f64_{method}"
    assert hashlib.sha256(pts.tobytes()).digest() == bytes(g[f"{k}_points_sha256"])
    o = op.range_projection(pts, rem, H, W, fu, fd, remove=True, method=method)
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest()  # noqa: E731
    assert sha(o["range"]) == bytes(g[f"{k}_proj_range_sha256"])
    assert sha(o["remission"]) == bytes(g[f"{k}_proj_remissions_sha256"])
    assert sha(o["unproj_range"]) == bytes(g[f"{k}_unproj_range_sha256"])
    assert sha(pts[o["kept"]]) == bytes(g[f"{k}_points_kept_sha256"])
    assert sha(o["index"]) == bytes(g[f"{k}_{'proj_idx' if method == 'old' else 'index'}_sha256"])
    if method == "old":
        assert sha(o["xyz"]) == bytes(g[f"{k}_proj_xyz_sha256"]) and sha(o["mask"]) == bytes(g[f"{k}_proj_mask_sha256"])


# ---- the product has no CPU path --------------------------------------------------------------------------------------
def test_product_never_imports_the_oracle_and_fails_loudly_without_its_library(monkeypatch, tmp_path):
    """`oracle/` is test infrastructure: nothing under lidar_transfer_amd/ may import it, and without liblidarhip.so
    (and without a compiler to build it) every entry point must raise instead of computing something else."""
    import re
    pkg = os.path.join(ROOT, "lidar_transfer_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(".py"):
                src = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f"{fn} imports the oracle"
    from lidar_transfer_amd import _lib, build as _build
    from lidar_transfer_amd.raytracer import C_Trace
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_build, "LIB_PATH", str(tmp_path / "liblidarhip.so"))
    monkeypatch.setattr(_build, "needs_build", lambda: True)

    def no_compiler(*a, **k):
        raise RuntimeError("hipcc not found")
    monkeypatch.setattr(_build, "build_lib", no_compiler)
    with pytest.raises(RuntimeError, match="liblidarhip.so is missing"):
        _lib.load()
    z3, z = np.zeros(3, np.float32), np.zeros(1, np.float32)
    with pytest.raises(RuntimeError, match="liblidarhip.so is missing"):
        C_Trace(np.array([1, 0, 0], np.float32), z3, np.zeros(9, np.float32), np.array([0, 1, 2], np.int32),
                np.zeros(9, np.int32), np.zeros(3, np.float32), z3.copy(), np.zeros(3, np.int32), z.copy(), z.copy(),
                1, 1)


def test_compare_restatement_vs_reference_python():
    """oracle/compare.py == the reference's compare() + iouEval on the F7 golden (made by the reference's own Python)."""
    from oracle.compare import compare
    g = np.load(os.path.join(GOLD, "f7_post.npz"))
    r = compare(g["cmp_source_label"], g["cmp_source_color"], g["cmp_target_label"], g["cmp_source_range"],
                g["cmp_target_range"], g["cmp_source_rem"], g["cmp_target_rem"], nclasses=20)
    assert np.array_equal(r["range_diff"].view(np.int32), g["cmp_range_diff"].view(np.int32))
    assert np.array_equal(r["rem_diff"].view(np.int32), g["cmp_rem_diff"].view(np.int32))
    assert r["m_iou"] == float(g["cmp_m_iou"]) and r["m_acc"] == float(g["cmp_m_acc"])
    assert np.float32(r["MSE"]) == g["cmp_mse"]


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_compare_restatement_negative_labels_vs_reference_python(case):
    """Negative labels: the reference's in-place renumbering (laserscan.py:1216-1222) MERGES classes when a rank meets a
    value still to come; the restatement replays that, pinned by golden F7b (made by the reference's own compare())."""
    from oracle.compare import compare
    g = np.load(os.path.join(GOLD, "f7b_compare_negative.npz"))
    r = compare(g[f"{case}_source_label"], g[f"{case}_source_color"], g[f"{case}_target_label"],
                g[f"{case}_source_range"], g[f"{case}_target_range"], g[f"{case}_source_rem"], g[f"{case}_target_rem"],
                nclasses=20)
    assert np.array_equal(r["range_diff"].view(np.int32), g[f"{case}_range_diff"].view(np.int32))
    assert np.array_equal(r["rem_diff"].view(np.int32), g[f"{case}_rem_diff"].view(np.int32))
    assert r["m_iou"] == float(g[f"{case}_m_iou"]) and r["m_acc"] == float(g[f"{case}_m_acc"])
    assert np.float32(r["MSE"]) == g[f"{case}_mse"]


def test_projection_restatement_vs_the_live_reference_on_random_clouds():
    """Where the reference checkout is present (the build container; never the GPU box): oracle/projection.py against the
    reference's OWN `do_range_projection_new` + `do_label_projection_new` run right here, on 40 random float64 clouds --
    random image shapes, fields of view, `remove`, beam tables, depth-0 points, exact duplicates (depth ties), points outside
    the field of view.  Range, remission, index, proj_x / proj_y and label images must be the reference's arrays."""
    ref = os.environ.get("LT_REFERENCE", "/root/reference")
    if not os.path.isdir(os.path.join(ref, "auxiliary")):
        pytest.skip("reference checkout absent")
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden
    from oracle import projection as op
    from lidar_transfer_amd.synth import synth_cloud
    ls, _ = make_golden.import_reference()
    color_dict = {0: [0, 0, 0], 10: [245, 150, 100], 40: [255, 0, 255], 48: [75, 0, 75], 50: [0, 200, 255],
                  70: [0, 175, 0], 80: [150, 240, 255]}
    rng = np.random.default_rng(2025)
    filled = 0
    for k in range(40):
        H, W = int(rng.choice([5, 16, 32, 64])), int(rng.choice([7, 64, 257, 512, 1024]))
        fu, fd = float(rng.choice([2.0, 3.0, 10.0, 15.0])), -float(rng.choice([16.6, 25.0, 30.0]))
        n = int(rng.integers(500, 20000))
        pts, rem, lab = synth_cloud(300 + k, n, dtype=np.float64, fov_up=fu + 3.0, fov_down=fd - 3.0)
        pts[rng.integers(0, n, 3)] = 0.0
        dup = rng.integers(0, n, 40)
        pts[dup[:20]] = pts[dup[20:]]
        remove = bool(rng.random() < 0.7)
        beams = sorted((np.linspace(fd, fu, H) / 180.0 * np.pi).tolist()) if rng.random() < 0.3 else None
        lab = lab.astype(np.uint32)
        s = ls.SemLaserScan(H, W, 300, color_dict, None, beams)
        s.points, s.remissions, s.label = pts.copy(), rem.copy(), lab.copy()
        s.colorize()
        s.do_range_projection_new(fu, fd, remove=remove)
        s.do_label_projection_new()
        o = op.range_projection(pts, rem, H, W, fu, fd, beam_angles=beams, remove=remove, method="new")
        assert _bits_equal(o["range"], np.asarray(s.range_image)), k
        assert np.array_equal(o["index"], np.asarray(s.index)), k
        assert np.array_equal(o["remission"], np.asarray(s.proj_remissions)), k
        own = o["index"] >= 0
        assert np.array_equal(o["px"][o["index"][own]], np.asarray(s.proj_x)[own]), k
        assert np.array_equal(o["py"][o["index"][own]], np.asarray(s.proj_y)[own]), k
        lab_img = op.label_projection(o["index"], lab[o["kept"]])
        assert np.array_equal(lab_img[own], np.asarray(s.label_image)[..., 0][own].astype(np.int32)), k
        filled += int(own.sum())
    assert filled > 50000


def test_create_rays_vs_the_live_reference_on_random_sensors():
    """... and the host mirror of `create_rays` (laserscan.py:1092-1119) against the reference's own, for 30 random sensor
    models, where the checkout is present."""
    ref = os.environ.get("LT_REFERENCE", "/root/reference")
    if not os.path.isdir(os.path.join(ref, "auxiliary")):
        pytest.skip("reference checkout absent")
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden
    ls, _ = make_golden.import_reference()
    rng = np.random.default_rng(7)
    for _k in range(30):
        fu, fd = float(rng.uniform(0.5, 45.0)), -float(rng.uniform(5.0, 50.0))
        H, W = int(rng.integers(1, 130)), int(rng.integers(1, 2100))
        want = ls.MultiSemLaserScan.create_rays(None, fu, fd, H, W)
        got = create_rays(fu, fd, H, W)
        assert got.dtype == want.dtype and got.shape == want.shape
        assert np.array_equal(got.view(np.int32), np.asarray(want).view(np.int32)), (fu, fd, H, W)
