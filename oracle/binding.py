"""ctypes bindings for the CPU oracle (TEST INFRASTRUCTURE).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; the product package ``lidar_transfer_amd`` never does.

* :func:`oracle_trace`  -- our C restatement (oracle/lt_oracle.c)
* :func:`ref_trace`     -- the REAL reference ``ctrace`` (RayTracer.cpp:116-124) compiled
  from /root/reference into ``oracle/_ref/*.so`` by ``oracle/Makefile``
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

MODE_REF_BVH, MODE_BRUTE, MODE_LBVH = 0, 1, 2
NORM_SSE, NORM_EXACT, NORM_SSE_TABLE, NORM_AMD_TABLE = 0, 1, 2, 3


class Stats(C.Structure):
    _fields_ = [("t_setup_ms", C.c_double), ("t_build_ms", C.c_double), ("t_trace_ms", C.c_double),
                ("nodes_popped", C.c_longlong), ("tris_tested", C.c_longlong), ("box_tests", C.c_longlong),
                ("n_nodes", C.c_int), ("n_leaves", C.c_int), ("max_stack", C.c_int), ("n_hits", C.c_int)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def build(quiet: bool = True) -> None:
    """(Re)build liblt_oracle.so and, when /root/reference is present, oracle/_ref."""
    subprocess.run(["make", "-C", HERE, "all"], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


_oracle = None
_refs = {}
_dense = None


def dense_lib():
    """oracle/liblt_tsdf_dense.so: the one-thread-per-voxel HIP restatement of the reference's pycuda ``integrate`` kernel
    (fusion_lidar.py:66-229; oracle/lt_tsdf_dense.hip) -- the A/B partner of the product's TSDF kernels on the GPU."""
    global _dense
    if _dense is None:
        path = os.path.join(HERE, "liblt_tsdf_dense.so")
        if not os.path.exists(path):
            subprocess.run(["make", "-C", HERE, "dense"], check=True, stdout=subprocess.DEVNULL)
        _dense = C.CDLL(path)
    return _dense


_ref_tsdf = {}


def ref_tsdf_available() -> bool:
    return all(os.path.exists(os.path.join(HERE, "_ref", f)) for f in ("libref_tsdf_integrate.so",
                                                                       "libref_tsdf_integrate_plain.so"))


def ref_tsdf_lib(merge: bool = True):
    """oracle/_ref/libref_tsdf_integrate.so: the reference's OWN ``integrate`` kernel -- the CUDA C text held in
    /root/reference/auxiliary/fusion_lidar.py:66-229, read in place and compiled by hipcc for gfx950
    (oracle/build_ref_tsdf.py) -- behind a restatement of the pycuda launch (oracle/ref_tsdf_launch.inc,
    fusion_lidar.py:232-250, :267-287).  ``merge=False``: ..._plain.so, the same text with its hard-wired
    ``bool merge = true;`` flipped (the plain running average branch).  ``ref_tsdf_integrate(tsdf, weight, color, rem,
    dims, origin, voxel_size, trunc_margin, fov_up_deg, fov_down_deg, color_im, depth_im, rem_im, im_h, im_w, obs_weight,
    stream)`` on DEVICE pointers; ``ref_tsdf_geometry(dims, out[5])`` -> threads, grid x/y/z, loops."""
    key = bool(merge)
    if key not in _ref_tsdf:
        name = "libref_tsdf_integrate.so" if key else "libref_tsdf_integrate_plain.so"
        lib = C.CDLL(os.path.join(HERE, "_ref", name))
        vp, ip, fp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float)
        lib.ref_tsdf_integrate.argtypes = [vp, vp, vp, vp, ip, fp, C.c_float, C.c_float, C.c_float, C.c_float, vp, vp, vp,
                                           C.c_int, C.c_int, C.c_float, vp]
        lib.ref_tsdf_integrate.restype = C.c_int
        lib.ref_tsdf_geometry.argtypes = [ip, ip]
        lib.ref_tsdf_geometry.restype = C.c_int
        _ref_tsdf[key] = lib
    return _ref_tsdf[key]


def _lib():
    global _oracle
    if _oracle is None:
        path = os.path.join(HERE, "liblt_oracle.so")
        if not os.path.exists(path):
            build()
        lib = C.CDLL(path)
        fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)
        lib.lto_trace.argtypes = [fp, fp, fp, ip, ip, fp, C.c_int, C.c_int, C.c_int, C.c_int, fp, ip, fp, fp, ip,
                                  C.c_int, C.c_int, C.c_int, C.c_float, C.POINTER(Stats)]
        lib.lto_trace.restype = C.c_int
        lib.lto_normalize_rays.argtypes = [fp, C.c_int, C.c_int, fp]
        lib.lto_normalize_rays.restype = None
        lib.lto_num_threads.restype = C.c_int
        lib.lto_tsdf_integrate.argtypes = [fp, fp, fp, fp, C.c_int, C.c_int, C.c_int, fp, C.c_float, C.c_int, C.c_int,
                                           C.c_float, C.c_float, C.c_float, C.c_float, fp, fp, fp, C.c_int]
        lib.lto_tsdf_integrate.restype = None
        _oracle = lib
    return _oracle


def ref_available(kind: str = "strict") -> bool:
    return os.path.exists(os.path.join(HERE, "_ref", f"libref_{kind}.so"))


def _ref(kind: str):
    if kind not in _refs:
        lib = C.CDLL(os.path.join(HERE, "_ref", f"libref_{kind}.so"))
        fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)
        lib.ctrace.argtypes = [fp, fp, fp, ip, ip, fp, C.c_int, C.c_int, C.c_int, C.c_int, fp, ip, fp, fp]
        lib.ctrace.restype = None
        _refs[kind] = lib
    return _refs[kind]


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32).reshape(-1)
    return a, a.ctypes.data_as(C.POINTER(C.c_int))


def _outputs(n_rays):
    return dict(endpoints=np.zeros(3 * n_rays, np.float32), endcolors=np.zeros(3 * n_rays, np.int32),
                range=np.zeros(n_rays, np.float32), endrem=np.zeros(n_rays, np.float32))


def oracle_trace(rays, origin, verts, faces, colors, rem, H, mode=MODE_REF_BVH, norm=NORM_EXACT, nthreads=0,
                 pad=0.0):
    """Run the restatement; returns dict(endpoints [R,3], endcolors [R,3], range [R], endrem [R], tri [R], stats)."""
    lib = _lib()
    rays_a, rays_p = _f(rays)
    org_a, org_p = _f(origin)
    v_a, v_p = _f(verts)
    f_a, f_p = _i(faces)
    c_a, c_p = _i(colors)
    r_a, r_p = _f(rem)
    n_rays = rays_a.size // 3
    out = _outputs(n_rays)
    tri = np.full(n_rays, -1, np.int32)
    st = Stats()
    rc = lib.lto_trace(rays_p, org_p, v_p, f_p, c_p, r_p, n_rays, v_a.size // 3, f_a.size // 3, int(H),
                       out["endpoints"].ctypes.data_as(C.POINTER(C.c_float)),
                       out["endcolors"].ctypes.data_as(C.POINTER(C.c_int)),
                       out["range"].ctypes.data_as(C.POINTER(C.c_float)),
                       out["endrem"].ctypes.data_as(C.POINTER(C.c_float)),
                       tri.ctypes.data_as(C.POINTER(C.c_int)), int(mode), int(norm), int(nthreads), float(pad),
                       C.byref(st))
    if rc != 0:
        raise RuntimeError(f"lto_trace failed rc={rc}")
    out["endpoints"] = out["endpoints"].reshape(-1, 3)
    out["endcolors"] = out["endcolors"].reshape(-1, 3)
    out["tri"] = tri
    out["stats"] = st.asdict()
    return out


def ref_trace(rays, origin, verts, faces, colors, rem, H, kind="strict"):
    """Run the real reference ``ctrace``; same dict minus ``tri``/``stats`` (stdout chatter is the reference's)."""
    lib = _ref(kind)
    rays_a, rays_p = _f(rays)
    org_a, org_p = _f(origin)
    v_a, v_p = _f(verts)
    f_a, f_p = _i(faces)
    c_a, c_p = _i(colors)
    r_a, r_p = _f(rem)
    n_rays = rays_a.size // 3
    out = _outputs(n_rays)
    lib.ctrace(rays_p, org_p, v_p, f_p, c_p, r_p, n_rays, v_a.size // 3, f_a.size // 3, int(H),
               out["endpoints"].ctypes.data_as(C.POINTER(C.c_float)),
               out["endcolors"].ctypes.data_as(C.POINTER(C.c_int)),
               out["range"].ctypes.data_as(C.POINTER(C.c_float)),
               out["endrem"].ctypes.data_as(C.POINTER(C.c_float)))
    out["endpoints"] = out["endpoints"].reshape(-1, 3)
    out["endcolors"] = out["endcolors"].reshape(-1, 3)
    return out


def normalize_rays(rays, norm=NORM_EXACT):
    lib = _lib()
    rays_a, rays_p = _f(rays)
    out = np.zeros_like(rays_a)
    lib.lto_normalize_rays(rays_p, rays_a.size // 3, int(norm), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out.reshape(-1, 3)


def num_threads() -> int:
    return int(_lib().lto_num_threads())


def tsdf_integrate(vols, dims, origin, voxel_size, fov_up, fov_down, color_im, depth_im, rem_im, obs_weight=1.0,
                   merge=True):
    """In-place update of ``vols = (tsdf, weight, color, rem)`` (float32 [dx,dy,dz] C-order) by the C
    restatement of the reference's CUDA kernel (fusion_lidar.py:66-229)."""
    lib = _lib()
    fp = C.POINTER(C.c_float)
    for v in vols:
        assert v.dtype == np.float32 and v.flags["C_CONTIGUOUS"]
    H, W = depth_im.shape
    ims = [np.ascontiguousarray(x, dtype=np.float32) for x in (color_im, depth_im, rem_im)]
    org = np.ascontiguousarray(origin, dtype=np.float32)
    lib.lto_tsdf_integrate(*[v.ctypes.data_as(fp) for v in vols], int(dims[0]), int(dims[1]), int(dims[2]),
                           org.ctypes.data_as(fp), float(voxel_size), int(H), int(W), float(voxel_size * 5), float(obs_weight), float(fov_up), float(fov_down),
                           *[x.ctypes.data_as(fp) for x in ims], int(bool(merge)))


def marching_cubes(tsdf, color_vol, rem_vol, voxel_size, origin):
    """``get_mesh`` (fusion_lidar.py:403-424) = scikit-image 0.18's ``marching_cubes_lewiner`` + the attribute look-ups,
    restated in oracle/lt_mc_oracle.c: the SAME ARRAYS as the reference returns -- vertices and faces, values and order
    (golden F10 of the real scikit-image; tools/mc_lewiner_fuzz.py).  Returns
    ``(verts [V,3] f32 world, faces [F,3] i32, colors [V,3] i32 (r, g, b after the uint8 wrap), rem [V] f32)``."""
    lib = _lib()
    fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)
    lib.lto_marching_cubes.argtypes = [fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_float, fp, fp, ip, ip, fp, C.c_int,
                                       C.c_int, ip, ip]
    lib.lto_marching_cubes.restype = C.c_int
    vols = [np.ascontiguousarray(v, dtype=np.float32) for v in (tsdf, color_vol, rem_vol)]
    nx, ny, nz = vols[0].shape
    org = np.ascontiguousarray(origin, dtype=np.float32)
    nv, nf = C.c_int(0), C.c_int(0)
    args = [v.ctypes.data_as(fp) for v in vols] + [nx, ny, nz, float(voxel_size), org.ctypes.data_as(fp)]
    rc = lib.lto_marching_cubes(*args, None, None, None, None, 0, 0, C.byref(nv), C.byref(nf))
    if rc != 0:
        raise RuntimeError(f"lto_marching_cubes (count) failed rc={rc}")
    verts = np.zeros((max(nv.value, 1), 3), np.float32)
    faces = np.zeros((max(nf.value, 1), 3), np.int32)
    colors = np.zeros((max(nv.value, 1), 3), np.int32)
    rem = np.zeros(max(nv.value, 1), np.float32)
    rc = lib.lto_marching_cubes(*args, verts.ctypes.data_as(fp), faces.ctypes.data_as(ip), colors.ctypes.data_as(ip),
                                rem.ctypes.data_as(fp), verts.shape[0], faces.shape[0], C.byref(nv), C.byref(nf))
    if rc != 0:
        raise RuntimeError(f"lto_marching_cubes failed rc={rc}")
    return verts[:nv.value], faces[:nf.value], colors[:nv.value], rem[:nv.value]
