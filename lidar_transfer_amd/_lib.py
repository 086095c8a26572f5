"""ctypes binding of liblidarhip.so -- the C ABI declared in include/lidarhip.h.

There is deliberately NO fallback: if the HIP library is missing or a call
fails, a ``RuntimeError`` is raised.  Nothing in this package computes the hot
path on the CPU.
"""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build

LT_OK = 0
LT_TRACE_WRITE_MISSES = 1
LT_TRACE_COUNT = 2
LT_TRACE_NORM_EXACT = 4
LT_TRACE_LABEL_IMAGE = 8
LT_TRACE_NORM_AMD = 16
LT_PROJ_REMOVE = 1
LT_PROJ_NEW = 2
LT_TSDF_MERGE = 1
LT_TSDF_HOST_MODE = 2

#: every symbol include/lidarhip.h declares (checked by tests/test_abi.py)
SYMBOLS = ["lt_ctrace", "lt_ctrace_ex", "lt_scene_create", "lt_scene_set_mesh_dev", "lt_scene_set_mesh_host",
           "lt_scene_build", "lt_scene_trace_dev", "lt_scene_status", "lt_scene_destroy", "lt_last_error",
           "lt_version", "lt_create_rays_dev", "lt_range_projection_dev", "lt_range_projection", "lt_rayset_create_dev",
           "lt_rayset_destroy", "lt_scene_render_dev", "lt_scene_render_batch_dev", "lt_scene_set_probe", "lt_reverse_projection_dev",
           "lt_pack_scan_dev", "lt_compare_dev", "lt_tsdf_create", "lt_tsdf_reset", "lt_tsdf_integrate_dev",
           "lt_tsdf_volumes", "lt_tsdf_volume_stride", "lt_tsdf_touch", "lt_tsdf_destroy", "lt_mesh_create", "lt_mesh_destroy", "lt_tsdf_extract_mesh_dev",
           "lt_marching_cubes_dev", "lt_mesh_get", "lt_scene_set_mesh", "lt_fusion_scan_dev", "lt_hostpipe_create", "lt_hostpipe_submit", "lt_hostpipe_wait",
           "lt_hostpipe_flush", "lt_hostpipe_destroy", "lt_host_alloc", "lt_host_free", "lt_projector_create",
           "lt_projector_destroy", "lt_range_projection_batch_dev", "lt_mesh_renumber_dev",
           "lt_tsdf_integrate_multi_dev", "lt_deform_scan_dev", "lt_mm_state_create", "lt_mm_state_destroy", "lt_mm_state_reset",
           "lt_mm_geometry_dev", "lt_mm_geometry_get", "lt_mergemesh_scan_dev", "lt_mergemesh_rerun_dev", "lt_abi_version"]
LT_ABI_VERSION = 7   # include/lidarhip.h: layout version of the structs mirrored below


class Stats(C.Structure):
    """Mirror of ``lt_stats`` (include/lidarhip.h)."""
    _fields_ = [("ms_bounds", C.c_float), ("ms_morton", C.c_float), ("ms_sort", C.c_float),
                ("ms_gather", C.c_float), ("ms_segtree", C.c_float), ("ms_hierarchy", C.c_float),
                ("ms_build", C.c_float), ("ms_trace", C.c_float), ("n_faces", C.c_int), ("n_nodes", C.c_int),
                ("n_rays", C.c_int), ("n_hits", C.c_int), ("nodes_visited", C.c_ulonglong),
                ("tris_tested", C.c_ulonglong), ("stack_overflows", C.c_ulonglong), ("entries_culled", C.c_ulonglong)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Cloud(C.Structure):
    """Mirror of ``lt_cloud`` (include/lidarhip.h): DEVICE pointers of one point cloud of a batch."""
    _fields_ = [("points", C.c_void_p), ("rem", C.c_void_p), ("label", C.c_void_p), ("n", C.c_int)]


class ProjImages(C.Structure):
    """Mirror of ``lt_proj_images``: the [H*W] DEVICE images of one cloud; NULL = not wanted."""
    _fields_ = [(k, C.c_void_p) for k in ("idx", "range", "xyz", "rem", "label", "color", "mask", "label_folded", "proj_x",
                                           "proj_y", "proj_xf", "proj_yf", "n_kept", "bnds")]


class MMGeometry(C.Structure):
    """Mirror of ``lt_mm_geometry``: the volume geometry of one ``mergemesh`` output scan."""
    _fields_ = [("bnds_given", C.c_double * 6), ("bnds_after", C.c_double * 6), ("dim", C.c_int * 3), ("status", C.c_int),
                ("ticket", C.c_int), ("reserved", C.c_int)]


_lib = None


def lib_path() -> str:
    return _build.LIB_PATH


def load():
    """Load (building first if the sources are newer) and type the library."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB_PATH
    if _build.needs_build():
        try:
            _build.build_lib()
        except RuntimeError as e:  # no hipcc: a prebuilt library is acceptable, nothing else is
            if not os.path.exists(path):
                raise RuntimeError(f"liblidarhip.so is missing and cannot be built: {e}") from e
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64.so.7 /
    # libhsa-runtime64.  If the system copy were loaded first (as a dependency of liblidarhip.so)
    # and torch's second, torch would find "No HIP GPUs".  Importing torch first makes the dynamic
    # linker resolve our DT_NEEDED libamdhip64.so.7 to the copy that is already mapped.
    # LIDARHIP_NO_TORCH=1: a numpy-only process (C_Trace / HostScanPipeline callers) that will never import torch may
    # skip this and run on the system ROCm runtime instead of the one bundled with the torch wheel.
    if os.environ.get("LIDARHIP_NO_TORCH", "") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    try:
        lib = C.CDLL(path)
    except OSError as e:
        raise RuntimeError(f"cannot load {path}: {e}") from e
    fp, ip, vp = C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_void_p
    sp = C.POINTER(Stats)
    lib.lt_ctrace.argtypes = [fp, fp, fp, ip, ip, fp, C.c_int, C.c_int, C.c_int, C.c_int, fp, ip, fp, fp]
    lib.lt_ctrace_ex.argtypes = lib.lt_ctrace.argtypes + [ip, sp]
    lib.lt_scene_create.argtypes = [C.POINTER(vp), C.c_int]
    lib.lt_scene_set_mesh_dev.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int]
    lib.lt_scene_set_mesh_host.argtypes = [vp, fp, ip, ip, fp, C.c_int, C.c_int, vp]
    lib.lt_scene_build.argtypes = [vp, vp, sp]
    lib.lt_scene_trace_dev.argtypes = [vp, vp, fp, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_uint, vp, sp]
    lib.lt_scene_status.argtypes = [vp]
    lib.lt_scene_destroy.argtypes = [vp]
    for name in ("lt_ctrace", "lt_ctrace_ex", "lt_scene_create", "lt_scene_set_mesh_dev", "lt_scene_set_mesh_host",
                 "lt_scene_build", "lt_scene_trace_dev", "lt_scene_status", "lt_scene_destroy"):
        getattr(lib, name).restype = C.c_int
    lib.lt_last_error.restype = C.c_char_p
    lib.lt_version.restype = C.c_char_p
    lib.lt_rayset_create_dev.argtypes = [C.POINTER(vp), vp, C.c_int, C.c_int, C.c_uint, vp]
    lib.lt_rayset_create_dev.restype = C.c_int
    lib.lt_rayset_destroy.argtypes = [vp]
    lib.lt_rayset_destroy.restype = C.c_int
    lib.lt_scene_render_dev.argtypes = [vp, vp, fp, vp, vp, vp, vp, vp, C.c_uint, vp, sp]
    lib.lt_scene_render_dev.restype = C.c_int
    pvp = C.POINTER(vp)
    lib.lt_scene_render_batch_dev.argtypes = [C.c_int, pvp, pvp, fp, pvp, pvp, pvp, pvp, pvp, C.c_uint, vp]
    lib.lt_scene_render_batch_dev.restype = C.c_int
    lib.lt_reverse_projection_dev.argtypes = [vp, vp, vp, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, vp, vp]
    lib.lt_reverse_projection_dev.restype = C.c_int
    lib.lt_pack_scan_dev.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, C.POINTER(C.c_int), vp]
    lib.lt_pack_scan_dev.restype = C.c_int
    lib.lt_compare_dev.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp]
    lib.lt_compare_dev.restype = C.c_int
    lib.lt_tsdf_create.argtypes = [C.POINTER(vp), C.POINTER(C.c_double), C.c_double, C.c_double, C.c_double, C.c_int]
    lib.lt_tsdf_reset.argtypes = [vp, vp]
    lib.lt_tsdf_integrate_dev.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_float, C.c_uint, vp]
    lib.lt_tsdf_integrate_multi_dev.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.c_int, C.c_int,
                                                C.c_float, C.c_uint, vp]
    lib.lt_tsdf_integrate_multi_dev.restype = C.c_int
    lib.lt_tsdf_volumes.argtypes = [vp, ip, fp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    lib.lt_tsdf_destroy.argtypes = [vp]
    lib.lt_tsdf_touch.argtypes = [vp]
    for name in ("lt_tsdf_create", "lt_tsdf_reset", "lt_tsdf_integrate_dev", "lt_tsdf_volumes", "lt_tsdf_destroy",
                 "lt_tsdf_touch"):
        getattr(lib, name).restype = C.c_int
    lib.lt_mesh_create.argtypes = [C.POINTER(vp), C.c_int]
    lib.lt_mesh_destroy.argtypes = [vp]
    lib.lt_tsdf_extract_mesh_dev.argtypes = [vp, vp, vp, fp]
    lib.lt_marching_cubes_dev.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, fp, vp, vp, fp]
    lib.lt_mesh_renumber_dev.argtypes = [vp, vp]
    lib.lt_mesh_renumber_dev.restype = C.c_int
    lib.lt_mesh_get.argtypes = [vp, ip, ip, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    lib.lt_scene_set_mesh.argtypes = [vp, vp]
    lib.lt_fusion_scan_dev.argtypes = [vp, vp, vp, vp, C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.c_int, C.c_int,
                                       C.c_float, C.c_uint, fp, vp, vp, vp, vp, vp, C.c_uint, vp, C.c_int]
    for name in ("lt_mesh_create", "lt_mesh_destroy", "lt_tsdf_extract_mesh_dev", "lt_marching_cubes_dev",
                 "lt_mesh_get", "lt_scene_set_mesh", "lt_fusion_scan_dev"):
        getattr(lib, name).restype = C.c_int
    lib.lt_hostpipe_create.argtypes = [C.POINTER(vp), fp, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_int]
    lib.lt_hostpipe_submit.argtypes = [vp, fp, fp, ip, vp, C.c_int, fp, C.c_int, C.c_int, fp, ip, fp, fp, ip,
                                       C.POINTER(C.c_int)]
    lib.lt_hostpipe_wait.argtypes = [vp, C.c_int]
    lib.lt_hostpipe_flush.argtypes = [vp]
    lib.lt_hostpipe_destroy.argtypes = [vp]
    lib.lt_host_alloc.argtypes = [C.POINTER(vp), C.c_size_t]
    lib.lt_host_free.argtypes = [vp]
    for name in ("lt_hostpipe_create", "lt_hostpipe_submit", "lt_hostpipe_wait", "lt_hostpipe_flush",
                 "lt_hostpipe_destroy", "lt_host_alloc", "lt_host_free"):
        getattr(lib, name).restype = C.c_int
    lib.lt_scene_set_probe.argtypes = [vp, vp, vp]
    lib.lt_scene_set_probe.restype = C.c_int
    lib.lt_create_rays_dev.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int, vp, vp]
    lib.lt_create_rays_dev.restype = C.c_int
    proj = [vp, C.c_int, vp, vp, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, vp, C.c_int, C.c_uint, vp, C.c_int,
            vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_float, C.c_float, C.c_float,
            C.POINTER(C.c_int)]
    lib.lt_range_projection.argtypes = proj
    lib.lt_range_projection.restype = C.c_int
    lib.lt_range_projection_dev.argtypes = proj + [vp]
    lib.lt_range_projection_dev.restype = C.c_int
    lib.lt_projector_create.argtypes = [C.POINTER(vp), C.c_int]
    lib.lt_projector_destroy.argtypes = [vp]
    lib.lt_range_projection_batch_dev.argtypes = [vp, C.c_int, C.POINTER(Cloud), C.c_int, C.c_double, C.c_double, C.c_int,
                                                  C.c_int, vp, C.c_int, C.c_uint, vp, C.c_int, C.POINTER(ProjImages),
                                                  C.c_float, C.c_float, C.c_float, vp]
    lib.lt_deform_scan_dev.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.POINTER(Cloud), C.c_int, C.c_double, C.c_double, C.c_int,
                                       C.c_int, vp, C.c_int, C.c_float, C.c_uint, fp, vp, vp, vp, vp, vp, C.c_uint, vp, C.c_int]
    lib.lt_deform_scan_dev.restype = C.c_int
    for name in ("lt_projector_create", "lt_projector_destroy", "lt_range_projection_batch_dev"):
        getattr(lib, name).restype = C.c_int
    lib.lt_mm_state_create.argtypes = [C.POINTER(vp), C.POINTER(C.c_double), C.c_int, C.c_double, C.c_int]
    lib.lt_mm_state_destroy.argtypes = [vp]
    lib.lt_mm_state_reset.argtypes = [vp, C.POINTER(C.c_double), vp]
    lib.lt_mm_geometry_dev.argtypes = [vp, vp, C.c_int, C.POINTER(C.c_int), vp]
    lib.lt_mergemesh_scan_dev.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, C.POINTER(Cloud), C.c_int, C.c_double, C.c_double, C.c_int,
                                          C.c_int, vp, C.c_int, C.c_float, C.c_uint, fp, vp, vp, vp, vp, vp, C.c_uint, vp,
                                          C.POINTER(MMGeometry), C.POINTER(C.c_int)]
    lib.lt_mergemesh_rerun_dev.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_float, C.c_uint, fp, vp, vp, vp, vp, vp, C.c_uint, vp]
    lib.lt_mm_geometry_get.argtypes = [vp, C.c_int, C.POINTER(MMGeometry)]
    for name in ("lt_mm_state_create", "lt_mm_state_destroy", "lt_mm_state_reset", "lt_mm_geometry_dev", "lt_mm_geometry_get",
                 "lt_mergemesh_scan_dev", "lt_mergemesh_rerun_dev", "lt_abi_version"):
        getattr(lib, name).restype = C.c_int
    lib.lt_abi_version.argtypes = []
    lib.lt_tsdf_volume_stride.argtypes = []
    if lib.lt_abi_version() != LT_ABI_VERSION:  # a stale prebuilt library: its structs are laid out differently
        raise RuntimeError(f"{path}: ABI version {lib.lt_abi_version()} != {LT_ABI_VERSION} of this binding (rebuild the library)")
    _lib = lib
    return lib


def check(rc: int, what: str = "liblidarhip") -> None:
    if rc != LT_OK:
        msg = load().lt_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed (status {rc}): {msg}")
