"""Seeded synthetic inputs for the virtual-LiDAR path.

The reference's only sample data (``minimal.zip``) and SemanticKITTI are not
available offline, so every workload in BASELINE.json is reproduced with
shape-faithful synthetic inputs:

* :func:`synth_scene` -- a labelled triangle mesh shaped like the output of the
  reference's ``TSDFVolume.get_mesh`` (auxiliary/fusion_lidar.py:403-424):
  ``verts [V,3] f32`` in metres inside ``voxel_bounds``, ``faces [F,3] i32``,
  ``colors [V,3] i32`` with the semantic label in channel 2, ``rem [V] f32``.
  Ground height-field + boxes (buildings / cars) + poles, all vertices on a
  lattice in at least two coordinates, which yields the flat axis-aligned
  triangles (zero-extent boxes) typical of marching-cubes meshes.
* :func:`synth_cloud` -- a labelled point cloud like one SemanticKITTI scan.
* :data:`WORKLOADS` -- the configurations C1..C4 of SURVEY.md section 8d.

Pure numpy host code; deterministic for a given seed on any machine (only
float64 ``sin``/``cos`` of lattice coordinates feed the geometry, rounded to
float32 afterwards).
"""
from __future__ import annotations

import numpy as np

LABEL_GROUND = 40
LABEL_BUILDING = 50
LABEL_POLE = 80
LABEL_CAR = 10

#: name -> (H, W, fov_up, fov_down, target_tris)   
WORKLOADS = {
    "C1": dict(H=64, W=1024, fov_up=3.0, fov_down=-25.0, tris=200_000),
    "C2": dict(H=64, W=2048, fov_up=3.0, fov_down=-25.0, tris=1_000_000),
    "C3": dict(H=32, W=1024, fov_up=10.0, fov_down=-30.0, tris=1_000_000),
    "C4": dict(H=128, W=2048, fov_up=15.0, fov_down=-25.0, tris=2_500_000),
}


def _grid_patch(origin, du, dv, nu, nv):
    """(nu+1)x(nv+1) lattice patch: verts = origin + i*du + j*dv, 2 tris per cell."""
    i, j = np.meshgrid(np.arange(nu + 1), np.arange(nv + 1), indexing="ij")
    verts = (origin[None, None, :] + i[..., None] * du[None, None, :]
             + j[..., None] * dv[None, None, :]).reshape(-1, 3)
    idx = (i * (nv + 1) + j)[:-1, :-1].reshape(-1)
    a, b, c, d = idx, idx + (nv + 1), idx + (nv + 1) + 1, idx + 1
    faces = np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, d], 1)], 0)
    return verts, faces


def _overlaps(cx, cy, hx, hy, placed, margin=0.75):
    for px, py, phx, phy, _, _ in placed:
        if abs(cx - px) < hx + phx + margin and abs(cy - py) < hy + phy + margin:
            return True
    return False


def synth_scene(seed: int, target_tris: int, bounds=(-50, 50, -50, 50, -5, 5), n_boxes: int = 60,
                n_poles: int = 40, allow_overlap: bool = False):
    """Return ``(verts f32 [V,3], faces i32 [F,3], colors i32 [V,3], rem f32 [V])``.

    ``allow_overlap=True`` lets boxes intersect, which creates coincident
    double surfaces and therefore rays with two hits at exactly equal ``t`` --
    a stress case for the tie rule (the reference keeps the first triangle its
    traversal happens to visit, BVH.cpp:59).  Marching-cubes meshes are
    manifolds without coincident sheets, so the default keeps boxes apart.
    """
    rng = np.random.default_rng(seed)
    xmin, xmax, ymin, ymax, zmin, zmax = [float(b) for b in bounds]
    # --- objects -------------------------------------------------------------
    boxes = []
    while len(boxes) < n_boxes:
        is_car = len(boxes) % 4 == 3
        if is_car:
            hx, hy, h = rng.uniform(1.6, 2.4), rng.uniform(0.7, 1.0), rng.uniform(1.3, 1.8)
        else:
            hx, hy, h = rng.uniform(1, 8), rng.uniform(1, 8), rng.uniform(1, 6)
        cx, cy = rng.uniform(xmin + hx, xmax - hx), rng.uniform(ymin + hy, ymax - hy)
        # keep a 3 m clearing around the sensor
        dx = max(abs(cx) - hx, 0.0)
        dy = max(abs(cy) - hy, 0.0)
        if np.hypot(dx, dy) < 3.0:
            continue
        if not allow_overlap and _overlaps(cx, cy, hx, hy, boxes):
            continue
        boxes.append((cx, cy, hx, hy, h, LABEL_CAR if is_car else LABEL_BUILDING))
    poles = []
    while len(poles) < n_poles:
        cx, cy = rng.uniform(xmin + 1, xmax - 1), rng.uniform(ymin + 1, ymax - 1)
        if np.hypot(cx, cy) < 3.0:
            continue
        if not allow_overlap and _overlaps(cx, cy, 0.1, 0.1, boxes + poles):
            continue
        poles.append((cx, cy, 0.1, 0.1, 4.0, LABEL_POLE))
    objs = boxes + poles
    z_base = -2.0

    def top(h, lab):
        return min(z_base + h + (0.0 if lab == LABEL_POLE else 2.0), zmax)

    area = (xmax - xmin) * (ymax - ymin)
    for cx, cy, hx, hy, h, lab in objs:
        area += (4 * hx + 4 * hy) * (top(h, lab) - z_base) + 4 * hx * hy
    s = float(np.sqrt(2.0 * area / max(target_tris, 8)))
    # --- ground height field -------------------------------------------------
    nx, ny = max(int(round((xmax - xmin) / s)), 1), max(int(round((ymax - ymin) / s)), 1)
    gv, gf = _grid_patch(np.array([xmin, ymin, 0.0]), np.array([(xmax - xmin) / nx, 0, 0.0]),
                         np.array([0, (ymax - ymin) / ny, 0.0]), nx, ny)
    gv[:, 2] = (-1.73 + 0.15 * np.sin(0.3 * gv[:, 0]) * np.cos(0.2 * gv[:, 1])
                + rng.normal(0.0, 0.02, gv.shape[0]))
    vs, fs, ls = [gv], [gf], [np.full(gv.shape[0], LABEL_GROUND, np.int32)]
    off = gv.shape[0]
    # --- boxes / poles: 4 walls + roof on the same lattice pitch -------------
    for cx, cy, hx, hy, h, lab in objs:
        x0, x1 = np.round((cx - hx) / s) * s, np.round((cx + hx) / s) * s
        y0, y1 = np.round((cy - hy) / s) * s, np.round((cy + hy) / s) * s
        if x1 <= x0:
            x1 = x0 + s
        if y1 <= y0:
            y1 = y0 + s
        z1 = top(h, lab)
        nxx, nyy = max(int(round((x1 - x0) / s)), 1), max(int(round((y1 - y0) / s)), 1)
        nzz = max(int(round((z1 - z_base) / s)), 1)
        ex, ey, ez = (x1 - x0) / nxx, (y1 - y0) / nyy, (z1 - z_base) / nzz
        patches = [
            (np.array([x0, y0, z_base]), np.array([0, ey, 0.0]), np.array([0, 0, ez]), nyy, nzz),  # x = x0
            (np.array([x1, y0, z_base]), np.array([0, ey, 0.0]), np.array([0, 0, ez]), nyy, nzz),  # x = x1
            (np.array([x0, y0, z_base]), np.array([ex, 0, 0.0]), np.array([0, 0, ez]), nxx, nzz),  # y = y0
            (np.array([x0, y1, z_base]), np.array([ex, 0, 0.0]), np.array([0, 0, ez]), nxx, nzz),  # y = y1
            (np.array([x0, y0, z1]), np.array([ex, 0, 0.0]), np.array([0, ey, 0.0]), nxx, nyy),    # roof
        ]
        for o, du, dv, nu, nv in patches:
            pv, pf = _grid_patch(o, du, dv, nu, nv)
            vs.append(pv)
            fs.append(pf + off)
            ls.append(np.full(pv.shape[0], lab, np.int32))
            off += pv.shape[0]
    verts = np.ascontiguousarray(np.concatenate(vs, 0).astype(np.float32))
    faces = np.ascontiguousarray(np.concatenate(fs, 0).astype(np.int32))
    labels = np.concatenate(ls, 0)
    colors = np.zeros((verts.shape[0], 3), np.int32)
    colors[:, 0] = rng.integers(0, 256, verts.shape[0])
    colors[:, 1] = (np.arange(verts.shape[0]) * 7) % 256
    colors[:, 2] = labels
    rem = rng.uniform(0.0, 1.0, verts.shape[0]).astype(np.float32)
    return verts, faces, colors, rem


def synth_cloud(seed: int, n_points: int = 120_000, dtype=np.float32, rmin: float = 2.0, rmax: float = 60.0,
                fov_up: float = 3.0, fov_down: float = -25.0):
    """Point cloud shaped like one HDL-64 scan: ``(points [N,3], rem f32 [N], label u32 [N])``.

    Ranges U(rmin, rmax), yaw U(-pi, pi), pitch slightly wider than the FOV so
    that the reference's ``remove`` branch (laserscan.py:245-254) has work.
    """
    rng = np.random.default_rng(seed)
    r = rng.uniform(rmin, rmax, n_points)
    yaw = rng.uniform(-np.pi, np.pi, n_points)
    pitch = np.deg2rad(rng.uniform(fov_down - 1.0, fov_up + 1.0, n_points))
    pts = np.stack([r * np.cos(pitch) * np.cos(yaw), r * np.cos(pitch) * np.sin(yaw), r * np.sin(pitch)], 1)
    rem = rng.uniform(0, 1, n_points).astype(np.float32)
    label = rng.choice(np.array([10, 40, 48, 50, 70, 80], np.uint32), n_points)
    return np.ascontiguousarray(pts.astype(dtype)), rem, label


def soup(verts, faces, colors, rem):
    """Triangle-soup form of a mesh with the face id smuggled through colour channel 0.

    The reference raytracer has no hit-triangle output; giving every face its
    own three vertices and writing the face index into ``colors[3f, 0]`` makes
    ``endcolors[:, 0]`` the hit face.  Exact for
    F < 2**24 (int -> float -> int round trip, RayTracer.cpp:36, :80).
    """
    f = faces.reshape(-1)
    sv = np.ascontiguousarray(verts[f])
    sf = np.arange(f.shape[0], dtype=np.int32).reshape(-1, 3)
    sc = np.ascontiguousarray(colors[f]).copy()
    sc[0::3, 0] = np.arange(faces.shape[0], dtype=np.int32)
    sr = np.ascontiguousarray(rem[f])
    return sv, sf, sc, sr
