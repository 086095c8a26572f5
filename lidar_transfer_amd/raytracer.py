"""Host-side mirror of the reference's native raytracer interface.

* :func:`C_Trace` -- same positional signature, dtype/contiguity checks and in-place output
  semantics as ``auxiliary/raytracer/RayTracerCython.pyx:15-33`` of the reference; numpy in,
  numpy out, one call = upload + BVH build + ray cast + download on the current HIP device.
* :class:`Scene` -- the split, device-resident API (``lt_scene_*`` in include/lidarhip.h) for
  callers that keep meshes, rays and images in HBM as ``torch`` tensors (the multi-GPU driver,
  ``bench.py``).  torch is only used for device memory and streams.

Everything routes through the C ABI of ``liblidarhip.so``; there is no CPU implementation here.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def _chk1d(name, a, dtype):
    """Reproduce the failure modes of Cython's ``float[::1]`` / ``int[::1]`` typed memoryviews."""
    if not isinstance(a, np.ndarray):
        raise TypeError(f"{name}: expected a numpy array, got {type(a).__name__}")
    if a.ndim != 1:
        raise ValueError(f"{name}: Buffer has wrong number of dimensions (expected 1, got {a.ndim})")
    if a.dtype != dtype:
        raise ValueError(f"{name}: Buffer dtype mismatch, expected '{np.dtype(dtype).name}' but got '{a.dtype.name}'")
    if not a.flags["C_CONTIGUOUS"]:
        raise ValueError(f"{name}: ndarray is not C-contiguous")
    return a


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def C_Trace(rays, origin, verts, faces, colors, rem, ray_endpoints, ray_colors, range_image, rem_image, H, W,
            tri_image=None, stats=None):
    """Drop-in for ``RayTracerCython.C_Trace`` (RayTracerCython.pyx:15-33).

    All arrays are flat, C-contiguous ``float32`` / ``int32``.  Outputs are modified in place and
    only for rays that hit (the caller pre-zeroes them, fusion_lidar.py:440-447).  ``W`` is unused,
    exactly as in the reference (width = n_rays // H, RayTracer.cpp:56).

    Extensions (keyword-only use): ``tri_image`` (flat int32) receives the hit face index;
    ``stats`` (a dict) is filled with per-phase timings and traversal counters.
    """
    rays = _chk1d("rays", rays, np.float32)
    origin = _chk1d("origin", origin, np.float32)
    verts = _chk1d("verts", verts, np.float32)
    faces = _chk1d("faces", faces, np.int32)
    colors = _chk1d("colors", colors, np.int32)
    rem = _chk1d("rem", rem, np.float32)
    ray_endpoints = _chk1d("ray_endpoints", ray_endpoints, np.float32)
    ray_colors = _chk1d("ray_colors", ray_colors, np.int32)
    range_image = _chk1d("range_image", range_image, np.float32)
    rem_image = _chk1d("rem_image", rem_image, np.float32)
    n_rays = len(rays) // 3
    n_verts = len(verts) // 3
    n_faces = len(faces) // 3
    lib = _lib.load()
    st = _lib.Stats()
    tri_p = None
    if tri_image is not None:
        tri_image = _chk1d("tri_image", tri_image, np.int32)
        tri_p = _ip(tri_image)
    rc = lib.lt_ctrace_ex(_fp(rays), _fp(origin), _fp(verts), _ip(faces), _ip(colors), _fp(rem), n_rays, n_verts,
                          n_faces, int(H), _fp(ray_endpoints), _ip(ray_colors), _fp(range_image), _fp(rem_image),
                          tri_p, C.byref(st) if stats is not None else None)
    _lib.check(rc, "lt_ctrace")
    if stats is not None:
        stats.update(st.asdict())


def _norm_flag(mode):
    """``exact_normalize`` argument -> trace flag: False / "intel" (default: the replayed RSQRTSS seed of an Intel
    host, where the golden vectors were made), True / "exact" (correctly rounded), "amd" (the seed of an AMD host,
    measured on the MI355X box's EPYC)."""
    if mode in (False, None, "intel"):
        return 0
    if mode in (True, "exact"):
        return _lib.LT_TRACE_NORM_EXACT
    if mode == "amd":
        return _lib.LT_TRACE_NORM_AMD
    raise ValueError("exact_normalize: False | True | 'intel' | 'exact' | 'amd'")


class Scene:
    """Device-resident mesh + BVH (``lt_scene`` in include/lidarhip.h).

    ``verts [V,3] f32``, ``faces [F,3] i32``, ``colors [V,3] i32``, ``rem [V] f32`` are contiguous
    ``torch`` tensors on the scene's GPU; they are borrowed, so keep them alive until the traces of
    this mesh have finished.
    """

    def __init__(self, device=None):
        import torch
        self._torch = torch
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", device if isinstance(device, int) else torch.device(device).index or 0)
        self._lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self._lib.lt_scene_create(C.byref(h), self.device.index), "lt_scene_create")
        self._h = h
        self._mesh = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.lt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _t(self, x, dtype, name):
        torch = self._torch
        if not isinstance(x, torch.Tensor):
            raise TypeError(f"{name}: expected a torch tensor")
        if x.device != self.device:
            raise ValueError(f"{name}: tensor on {x.device}, scene on {self.device}")
        if x.dtype != dtype or not x.is_contiguous():
            raise ValueError(f"{name}: must be contiguous {dtype}")
        return x

    def set_mesh(self, verts, faces, colors, rem):
        torch = self._torch
        verts = self._t(verts, torch.float32, "verts")
        faces = self._t(faces, torch.int32, "faces")
        colors = self._t(colors, torch.int32, "colors")
        rem = self._t(rem, torch.float32, "rem")
        self._mesh = (verts, faces, colors, rem)
        _lib.check(self._lib.lt_scene_set_mesh_dev(self._h, verts.data_ptr(), faces.data_ptr(), colors.data_ptr(),
                                                   rem.data_ptr(), verts.numel() // 3, faces.numel() // 3),
                   "lt_scene_set_mesh_dev")

    def set_device_mesh(self, mesh):
        """Attach a :class:`~lidar_transfer_amd.fusion.DeviceMesh` (``lt_scene_set_mesh``): the arrays marching cubes
        wrote are borrowed until the next extraction into ``mesh``."""
        self._mesh = mesh
        _lib.check(self._lib.lt_scene_set_mesh(self._h, mesh._h), "lt_scene_set_mesh")

    def _stream(self, stream):
        torch = self._torch
        if stream is None:
            stream = torch.cuda.current_stream(self.device)
        return C.c_void_p(stream.cuda_stream)

    def build(self, stream=None, stats=False):
        st = _lib.Stats()
        _lib.check(self._lib.lt_scene_build(self._h, self._stream(stream), C.byref(st) if stats else None),
                   "lt_scene_build")
        return st.asdict() if stats else None

    def trace(self, rays, origin, H, out=None, stream=None, write_misses=True, count=False, stats=False,
              exact_normalize=False, label_image=False):
        """Cast ``rays [R,3] f32`` (device) from ``origin`` (3 floats, host); returns dict of device tensors.

        ``out`` may carry preallocated ``endpoints [R,3] f32, endcolors [R,3] i32, range [R] f32,
        endrem [R] f32, tri [R] i32`` tensors to reuse.
        """
        torch = self._torch
        rays = self._t(rays, torch.float32, "rays")
        n_rays = rays.numel() // 3
        if out is None:
            out = self.alloc_outputs(n_rays, label_image=label_image)
        org = (C.c_float * 3)(*[float(v) for v in origin])
        flags = ((_lib.LT_TRACE_WRITE_MISSES if write_misses else 0) | (_lib.LT_TRACE_COUNT if count else 0)
                 | _norm_flag(exact_normalize)
                 | (_lib.LT_TRACE_LABEL_IMAGE if label_image else 0))
        st = _lib.Stats()

        def p(k):
            t = out.get(k)
            return t.data_ptr() if t is not None else None

        _lib.check(self._lib.lt_scene_trace_dev(self._h, rays.data_ptr(), org, n_rays, int(H), p("endpoints"),
                                                p("endcolors"), p("range"), p("endrem"), p("tri"), flags,
                                                self._stream(stream), C.byref(st) if (stats or count) else None),
                   "lt_scene_trace_dev")
        if stats or count:
            out = dict(out)
            out["stats"] = st.asdict()
        return out

    def render(self, rayset, origin, out=None, stream=None, write_misses=True, count=False, stats=False,
               label_image=False):
        """Closest hits of a :class:`RaySet` against the CURRENT mesh with the single-origin scatter
        strategy (``lt_scene_render_dev``): no BVH build; bit-identical to :meth:`build` + :meth:`trace`."""
        if out is None:
            out = self.alloc_outputs(rayset.n_rays, label_image=label_image)
        org = (C.c_float * 3)(*[float(v) for v in origin])
        flags = (_lib.LT_TRACE_WRITE_MISSES if write_misses else 0) | (_lib.LT_TRACE_COUNT if count else 0) | \
            (_lib.LT_TRACE_LABEL_IMAGE if label_image else 0)
        st = _lib.Stats()

        def p(k):
            t = out.get(k)
            return t.data_ptr() if t is not None else None

        _lib.check(self._lib.lt_scene_render_dev(self._h, rayset._h, org, p("endpoints"), p("endcolors"), p("range"),
                                                 p("endrem"), p("tri"), flags, self._stream(stream),
                                                 C.byref(st) if (stats or count) else None), "lt_scene_render_dev")
        if stats or count:
            out = dict(out)
            out["stats"] = st.asdict()
        return out

    @staticmethod
    def render_batch(scenes, raysets, origins, outs=None, stream=None, write_misses=True, label_image=False):
        """``lt_scene_render_batch_dev``: the scans ``(scenes[i], raysets[i], origins[i])`` -- at most 8, every
        scene distinct and with its own current mesh, the raysets normally one shared ray set -- rendered with three kernel launches for all of
        them.  Returns the list of output dicts (``outs[i]`` or freshly allocated)."""
        n = len(scenes)
        if not (n == len(raysets) == len(origins)) or (outs is not None and len(outs) != n):
            raise ValueError("render_batch: scenes, raysets, origins (and outs) must have one entry per scan")
        if n == 0:
            return []
        if outs is None:
            outs = [scenes[i].alloc_outputs(raysets[i].n_rays, label_image=label_image) for i in range(n)]
        vp = C.c_void_p
        arr = lambda vals: (vp * n)(*vals)  # noqa: E731
        org = (C.c_float * (3 * n))(*[float(v) for o in origins for v in o])

        def col(k):
            return arr([(o[k].data_ptr() if o.get(k) is not None else None) for o in outs])

        flags = (_lib.LT_TRACE_WRITE_MISSES if write_misses else 0) | (_lib.LT_TRACE_LABEL_IMAGE if label_image else 0)
        _lib.check(scenes[0]._lib.lt_scene_render_batch_dev(n, arr([s._h for s in scenes]), arr([r._h for r in raysets]),
                                                            org, col("endpoints"), col("endcolors"), col("range"),
                                                            col("endrem"), col("tri"), flags, scenes[0]._stream(stream)),
                   "lt_scene_render_batch_dev")
        return outs

    def set_probe(self, ev_start, ev_stop):
        """Record two ``torch.cuda.Event(enable_timing=True)`` around the dominant kernel of the next cast."""
        for ev in (ev_start, ev_stop):  # materialise the underlying hipEvent_t
            if not ev.cuda_event:
                ev.record()
        _lib.check(self._lib.lt_scene_set_probe(self._h, C.c_void_p(ev_start.cuda_event),
                                                C.c_void_p(ev_stop.cuda_event)), "lt_scene_set_probe")

    def alloc_outputs(self, n_rays, label_image=False):
        """Output images; with ``label_image`` the colour output is the [n_rays] semantic-label image
        (``LT_TRACE_LABEL_IMAGE``: channel 2 only, ``deform``'s ``label_image = ray_colors[:, :, 2]``)."""
        torch = self._torch
        d = self.device
        return dict(endpoints=torch.empty((n_rays, 3), dtype=torch.float32, device=d),
                    endcolors=torch.empty((n_rays,) if label_image else (n_rays, 3), dtype=torch.int32, device=d),
                    range=torch.empty((n_rays,), dtype=torch.float32, device=d),
                    endrem=torch.empty((n_rays,), dtype=torch.float32, device=d),
                    tri=torch.empty((n_rays,), dtype=torch.int32, device=d))

    def status(self):
        _lib.check(self._lib.lt_scene_status(self._h), "lt_scene_status")


class RaySet:
    """A ray batch prepared for :meth:`Scene.render` (``lt_rayset`` in include/lidarhip.h): directions
    normalised like the reference (Vector3.h:73-89) and binned by azimuth x elevation.  One per sensor
    model; reuse it for every scan."""

    def __init__(self, rays, H, exact_normalize=False, stream=None):
        import torch
        if not isinstance(rays, torch.Tensor) or rays.dtype != torch.float32 or not rays.is_contiguous() \
                or not rays.is_cuda:
            raise ValueError("rays: contiguous float32 CUDA tensor [R, 3] expected")
        self._lib = _lib.load()
        self.n_rays = (rays.numel() // 3 // int(H)) * int(H)
        self.H = int(H)
        st = torch.cuda.current_stream(rays.device) if stream is None else stream
        h = C.c_void_p()
        with torch.cuda.device(rays.device):
            _lib.check(self._lib.lt_rayset_create_dev(C.byref(h), rays.data_ptr(), rays.numel() // 3, int(H),
                                                      _norm_flag(exact_normalize),
                                                      C.c_void_p(st.cuda_stream)), "lt_rayset_create_dev")
            st.synchronize()
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.lt_rayset_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
