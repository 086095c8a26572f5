"""``throw_rays_at_mesh`` -- Python side of the drop-in boundary
(reference: ``TSDFVolume.throw_rays_at_mesh``, auxiliary/fusion_lidar.py:426-455).

The mesh may come from anywhere -- the reference's own ``TSDFVolume.get_mesh`` (host arrays, uploaded per call) or
this module's device-resident :class:`TSDFVolume` (fusion, marching cubes and ray cast all in HBM).  Either call
:func:`throw_rays_at_mesh` with any object that has the reference's ``get_mesh(color_lut)``, or
:func:`install` it on the reference's class::

    import auxiliary.fusion_lidar as fl
    from lidar_transfer_amd import fusion
    fusion.install(fl)          # fl.TSDFVolume.throw_rays_at_mesh now runs on the MI355X
"""
from __future__ import annotations

import numpy as np

from .raytracer import C_Trace


def throw_rays_at_mesh(volume, rays, origin, H, W, color_lut):
    """Same arguments, same 7-tuple, same shapes and dtypes as fusion_lidar.py:426-455:

    ``(ray_endpoints [R,3] f32, ray_colors [R,3] i32, verts, colors, faces, range_image [H,W] f32,
    rem_image [H,W] f32)`` -- with the hit triangle image available afterwards as
    ``volume.last_tri_image`` ([H,W] i32, -1 = miss; an extension).
    """
    verts, faces, norms, colors, rem = volume.get_mesh(color_lut)
    # Arrays must be contiguous and 1D (fusion_lidar.py:433-438)
    verts_c = np.ascontiguousarray(np.asarray(verts, dtype=np.float32).reshape(-1))
    faces_c = np.ascontiguousarray(np.asarray(faces, dtype=np.int32).reshape(-1))
    colors_c = np.ascontiguousarray(np.asarray(colors).reshape(-1).astype(np.int32))
    rem_c = np.ascontiguousarray(np.asarray(rem, dtype=np.float32).reshape(-1))
    rays = np.ascontiguousarray(np.asarray(rays, dtype=np.float32).reshape(-1))
    origin = np.ascontiguousarray(np.asarray(origin, dtype=np.float32).reshape(-1))
    ray_endpoints = np.zeros(H * W * 3, dtype=np.float32)
    ray_colors = np.zeros(H * W * 3, dtype=np.int32)
    range_image = np.zeros(H * W, dtype=np.float32)
    rem_image = np.zeros(H * W, dtype=np.float32)
    tri_image = np.full(H * W, -1, dtype=np.int32)
    C_Trace(rays, origin, verts_c, faces_c, colors_c, rem_c, ray_endpoints, ray_colors, range_image, rem_image, H, W,
            tri_image=tri_image)
    try:
        volume.last_tri_image = tri_image.reshape(-1, W)
    except AttributeError:
        pass
    return ray_endpoints.reshape(-1, 3), ray_colors.reshape(-1, 3), verts, colors, faces, \
        range_image.reshape(-1, W), rem_image.reshape(-1, W)


def unpack_deform(ray_colors, color_lut, H, W):
    """Result unpacking of ``MultiSemLaserScan.deform`` (auxiliary/laserscan.py:910-914, :1001-1005):
    ``label_image = ray_colors[..., 2]``, ``proj_color = color_lut[label_image]``."""
    proj_color = np.asarray(ray_colors).reshape(H, W, 3)
    label_image = np.copy(proj_color[:, :, 2])
    return label_image, np.asarray(color_lut)[label_image]


def install(fusion_module):
    """Replace ``TSDFVolume.throw_rays_at_mesh`` of the reference's ``auxiliary.fusion_lidar`` module."""
    fusion_module.TSDFVolume.throw_rays_at_mesh = throw_rays_at_mesh
    return fusion_module


class MeshVolume:
    """Minimal stand-in for a ``TSDFVolume`` that already holds a mesh (tests, batch drivers)."""

    def __init__(self, verts, faces, colors, rem, norms=None):
        self._mesh = (verts, faces, norms, colors, rem)

    def get_mesh(self, color_lut=None):
        return self._mesh

    throw_rays_at_mesh = throw_rays_at_mesh


class TSDFVolume:
    """Device-resident mirror of the reference's ``TSDFVolume`` fusion part
    (auxiliary/fusion_lidar.py:21-63 constructor, :252-287 ``integrate``, :395-400 ``get_volume``).

    Same constructor and ``integrate`` signature; the four fields live in HBM, one record per voxel (``lt_tsdf``) and every
    ``integrate`` is one launch of the HIP kernel -- no per-launch image round trip, no grid loop.
    ``merge=True`` (default) is the class-aware branch the reference's CUDA kernel runs; ``mode="numpy"`` selects the
    arithmetic of the reference's OTHER fusion mode instead -- its numpy branch (``FUSION_GPU_MODE == 0``, what it runs where
    pycuda is absent, fusion_lidar.py:290-388: float64 voxel projection, plain running average, no remissions) -- on the
    device (``LT_TSDF_HOST_MODE``; goldens F8 / F13 / F14 are made by that branch).  ``get_mesh`` / ``extract_mesh`` run
    marching cubes on the device (``lt_mc.hip``): the mesh is born in HBM and
    ``throw_rays_at_mesh`` renders it there.
    """

    def __init__(self, vol_bnds, voxel_size, fov_up, fov_down, device=None, merge=True, mode="cuda"):
        import ctypes as C

        import torch

        from . import _lib
        if mode not in ("cuda", "numpy"):
            raise ValueError("TSDFVolume: mode is 'cuda' (the reference's kernel) or 'numpy' (its FUSION_GPU_MODE == 0 branch)")
        self._lib = _lib.load()
        self._C, self._torch, self._libmod = C, torch, _lib
        self.fov_up, self.fov_down = fov_up, fov_down
        self.merge = merge
        self.mode = mode
        self._flags = _lib.LT_TSDF_HOST_MODE if mode == "numpy" else (_lib.LT_TSDF_MERGE if merge else 0)
        self._vol_bnds = np.array(vol_bnds, dtype=np.float64).reshape(3, 2)
        self._voxel_size = float(voxel_size)
        self._trunc_margin = self._voxel_size * 5
        self._vol_dim = np.ceil((self._vol_bnds[:, 1] - self._vol_bnds[:, 0]) / self._voxel_size).astype(int)
        bnds_in = self._vol_bnds.copy()  # the library derives the dimensions from the bounds as given (:33-34)
        self._vol_bnds[:, 1] = self._vol_bnds[:, 0] + self._vol_dim * self._voxel_size
        self._vol_origin = self._vol_bnds[:, 0].copy().astype(np.float32)
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        bnds = (C.c_double * 6)(*bnds_in.reshape(-1))
        h = C.c_void_p()
        _lib.check(self._lib.lt_tsdf_create(C.byref(h), bnds, self._voxel_size, float(fov_up), float(fov_down),
                                            self.device.index), "lt_tsdf_create")
        self._h = h

    def close(self):
        for name in ("_rs", "_scene", "_mesh"):
            obj = getattr(self, name, None)
            if obj is not None:
                (obj[2] if name == "_rs" else obj).close()
                setattr(self, name, None)
        if getattr(self, "_h", None):
            self._lib.lt_tsdf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def integrate(self, color_im, depth_im, rem_im, cam_pose=None, obs_weight=1.):
        """``color_im [H,W,3]`` (label image, three channels), ``depth_im``, ``rem_im`` ``[H,W]``; numpy or CUDA
        tensors.  ``cam_pose`` is accepted and ignored, as by the reference kernel (fusion_lidar.py:253-256)."""
        torch, C = self._torch, self._C

        def dev(a):
            if isinstance(a, torch.Tensor):
                return a.to(self.device, torch.float32).contiguous()
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)

        c = dev(color_im)
        # Fold RGB color image into a single channel image (fusion_lidar.py:261-264), float32 like the reference
        folded = torch.floor(c[:, :, 0] * 256 * 256 + c[:, :, 1] * 256 + c[:, :, 2]).contiguous()
        d, r = dev(depth_im), dev(rem_im)
        st = torch.cuda.current_stream(self.device)
        self._libmod.check(self._lib.lt_tsdf_integrate_dev(self._h, folded.data_ptr(), d.data_ptr(), r.data_ptr(),
                                                           d.shape[0], d.shape[1], float(obs_weight),
                                                           self._flags,
                                                           C.c_void_p(st.cuda_stream)), "lt_tsdf_integrate_dev")
        st.synchronize()  # the temporaries above must outlive the kernel

    def integrate_multi(self, observations, obs_weight=1.):
        """``integrate`` for a list of ``(color_im, depth_im, rem_im)`` observations, in order -- the loop of
        ``deform``'s mesh adaption (laserscan.py:889-897) as ONE native call (``lt_tsdf_integrate_multi_dev``): same
        volume, bit for bit, as one ``integrate`` per observation; on a fresh volume the class-aware update of up to
        eight observations runs as one pass over the union of their candidate voxels."""
        torch, C = self._torch, self._C

        def dev(a):
            if isinstance(a, torch.Tensor):
                return a.to(self.device, torch.float32).contiguous()
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)

        n = len(observations)
        if n == 0:
            return
        keep = []
        vp = C.c_void_p
        cp, dp, rp = (vp * n)(), (vp * n)(), (vp * n)()
        h = w = 0
        for k, (color_im, depth_im, rem_im) in enumerate(observations):
            c = dev(color_im)
            if c.dim() == 3:
                c = torch.floor(c[:, :, 0] * 256 * 256 + c[:, :, 1] * 256 + c[:, :, 2]).contiguous()
            d, r = dev(depth_im), dev(rem_im)
            if k and tuple(d.shape) != (h, w):
                raise ValueError("observations of one call must have one image shape")
            h, w = int(d.shape[0]), int(d.shape[1])
            keep += [c, d, r]
            cp[k], dp[k], rp[k] = c.data_ptr(), d.data_ptr(), r.data_ptr()
        st = torch.cuda.current_stream(self.device)
        self._libmod.check(self._lib.lt_tsdf_integrate_multi_dev(self._h, n, cp, dp, rp, h, w, float(obs_weight),
                                                                 self._flags,
                                                                 C.c_void_p(st.cuda_stream)), "lt_tsdf_integrate_multi_dev")
        st.synchronize()  # the temporaries above must outlive the kernels

    def reset(self):
        """Back to the initial volume (what ``TSDFVolume(...)`` of the reference starts from); only the voxel columns
        written since the last reset are re-initialised."""
        st = self._torch.cuda.current_stream(self.device)
        self._libmod.check(self._lib.lt_tsdf_reset(self._h, self._C.c_void_p(st.cuda_stream)), "lt_tsdf_reset")

    def touch(self):
        """Call after writing into the tensors of :meth:`get_volume_tensors` directly."""
        self._libmod.check(self._lib.lt_tsdf_touch(self._h), "lt_tsdf_touch")

    def get_volume(self):
        """``(tsdf, color, rem)`` as float32 numpy arrays ``[dx,dy,dz]`` (fusion_lidar.py:395-400)."""
        t, w, c, r = self.get_volume_tensors()
        return t.cpu().numpy(), c.cpu().numpy(), r.cpu().numpy()

    def extract_mesh(self, mesh=None, timed=False):
        """Marching cubes on the device over the current volume (``lt_tsdf_extract_mesh_dev``): returns a
        :class:`DeviceMesh` whose arrays stay in HBM -- ``Scene.set_device_mesh(mesh)`` renders it without any PCIe
        traffic.  Pass the previous ``mesh`` to reuse its buffers."""
        C = self._C
        if mesh is None:
            mesh = getattr(self, "_mesh", None) or DeviceMesh(self.device.index)
            self._mesh = mesh
        ms = (C.c_float * 2)()
        st = self._torch.cuda.current_stream(self.device)
        self._libmod.check(self._lib.lt_tsdf_extract_mesh_dev(self._h, mesh._h, C.c_void_p(st.cuda_stream),
                                                              ms if timed else None), "lt_tsdf_extract_mesh_dev")
        mesh.last_ms = (ms[0], ms[1]) if timed else None
        return mesh

    def get_mesh(self, color_lut=None):
        """``(verts, faces, norms, colors, rem)`` as numpy arrays like fusion_lidar.py:403-424: verts ``[V,3]`` f32 in
        world coordinates, faces ``[F,3]`` i32, colors ``[V,3]`` uint8 (r, g, b), rem ``[V]`` f32 -- the arrays scikit-image
        0.18 + fusion_lidar.py:409-423 return, element for element (golden F10).  ``norms`` is
        ``None``: scikit-image's vertex normals are returned by the reference but read by nothing on the path
        (only by the PLY writer that is commented out, fusion_lidar.py:430-431)."""
        v, f, c, r = self.extract_mesh().renumber().tensors()   # scikit-image's vertex numbers: the reference's arrays
        return v.cpu().numpy(), f.cpu().numpy(), None, c.cpu().numpy().astype(np.uint8), r.cpu().numpy()

    def throw_rays_at_mesh_device(self, rayset, origin, out=None, scene=None, label_image=False, renumber=False):
        """Fusion -> range image without leaving HBM: marching cubes on the device, then the single-origin render of
        ``rayset`` (a :class:`~lidar_transfer_amd.raytracer.RaySet`).  Returns the dict of device tensors of
        ``Scene.render`` plus ``mesh`` (the :class:`DeviceMesh`).  ``renumber``: number the vertices as scikit-image does
        before the render (the images do not depend on it; the mesh arrays then equal the reference's)."""
        from .raytracer import Scene
        mesh = self.extract_mesh()
        if renumber:
            mesh.renumber()
        if scene is None:
            scene = getattr(self, "_scene", None) or Scene(self.device.index)
            self._scene = scene
        scene.set_device_mesh(mesh)
        o = dict(scene.render(rayset, origin, out=out, label_image=label_image))
        o["mesh"] = mesh
        return o

    def throw_rays_at_mesh(self, rays, origin, H, W, color_lut=None):
        """Same arguments and 7-tuple as fusion_lidar.py:426-455 -- ``(ray_endpoints [R,3] f32, ray_colors [R,3] i32,
        verts, colors, faces, range_image [H,W] f32, rem_image [H,W] f32)`` -- computed without the mesh ever
        visiting the host on its way from the volume to the ray cast; the tuple's mesh members (which ``deform`` only
        keeps for the visualiser) and the images are downloaded at the end."""
        from .raytracer import RaySet
        torch = self._torch
        rays_t = torch.from_numpy(np.ascontiguousarray(np.asarray(rays, np.float32).reshape(-1, 3))).to(self.device)
        key = (rays_t.shape[0], int(H))
        cached = getattr(self, "_rs", None)
        if cached is None or cached[0] != key or not torch.equal(cached[1], rays_t):
            if cached is not None:
                cached[2].close()
            cached = (key, rays_t, RaySet(rays_t, int(H)))
            self._rs = cached
        o = self.throw_rays_at_mesh_device(cached[2], [float(x) for x in np.asarray(origin).reshape(-1)[:3]], renumber=True)
        torch.cuda.synchronize(self.device)
        v, f, c, r = o["mesh"].tensors()
        return (o["endpoints"].cpu().numpy().reshape(-1, 3), o["endcolors"].cpu().numpy().reshape(-1, 3),
                v.cpu().numpy(), c.cpu().numpy().astype(np.uint8), f.cpu().numpy(),
                o["range"].cpu().numpy().reshape(-1, W), o["endrem"].cpu().numpy().reshape(-1, W))

    def get_volume_tensors(self):
        """Zero-copy ``torch`` views ``[dx, dy, dz]`` of the four device volumes (tsdf, weight, color, rem).  The library keeps
        ONE record of the four fields per voxel (``lt_tsdf_volume_stride()`` floats apart): the views are strided, reads and
        writes (``copy_``, indexing; then :meth:`touch`) go straight to the volume; ``.contiguous()`` for a packed copy."""
        torch, C = self._torch, self._C
        dims = (C.c_int * 3)()
        org = (C.c_float * 3)()
        ptrs = [C.c_void_p() for _ in range(4)]
        self._libmod.check(self._lib.lt_tsdf_volumes(self._h, dims, org, *[C.byref(p) for p in ptrs]),
                           "lt_tsdf_volumes")
        n = int(dims[0]) * int(dims[1]) * int(dims[2])
        vs = int(self._lib.lt_tsdf_volume_stride())
        if [p.value for p in ptrs] != [ptrs[0].value + 4 * k for k in range(4)] or vs != 4:
            raise RuntimeError("lt_tsdf_volumes: not the interleaved layout this binding was written for")
        rec = _wrap_device_pointer(torch, ptrs[0].value, n * vs, self.device).view(int(dims[0]), int(dims[1]), int(dims[2]), vs)
        return [rec[..., k] for k in range(4)]


class DeviceMesh:
    """An indexed triangle mesh in HBM (``lt_mesh`` in include/lidarhip.h): what marching cubes writes and the ray cast
    reads.  ``verts [V,3] f32`` (world), ``faces [F,3] i32``, ``colors [V,3] i32`` (r, g, b -- label in channel 2),
    ``rem [V] f32`` are zero-copy ``torch`` views, valid until the next extraction into this object."""

    def __init__(self, device=None):
        import ctypes as C

        import torch

        from . import _lib
        self._C, self._torch, self._libmod = C, torch, _lib
        self._lib = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        h = C.c_void_p()
        _lib.check(self._lib.lt_mesh_create(C.byref(h), self.device.index), "lt_mesh_create")
        self._h = h
        self.last_ms = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.lt_mesh_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def extract(self, tsdf, color_vol, rem_vol, voxel_size, origin, timed=False):
        """Marching cubes (level 0) over caller-owned device fields ``[nx, ny, nz]`` f32 (``lt_marching_cubes_dev``)."""
        torch, C = self._torch, self._C
        for t in (tsdf, color_vol, rem_vol):
            if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
                    and t.shape == tsdf.shape and t.dim() == 3):
                raise ValueError("extract: three contiguous float32 CUDA tensors [nx, ny, nz] expected")
        org = (C.c_float * 3)(*[float(x) for x in origin])
        ms = (C.c_float * 2)()
        st = torch.cuda.current_stream(self.device)
        nx, ny, nz = tsdf.shape
        self._libmod.check(self._lib.lt_marching_cubes_dev(tsdf.data_ptr(), color_vol.data_ptr(), rem_vol.data_ptr(), nx,
                                                           ny, nz, float(voxel_size), org, self._h,
                                                           C.c_void_p(st.cuda_stream), ms if timed else None),
                           "lt_marching_cubes_dev")
        self.last_ms = (ms[0], ms[1]) if timed else None
        return self

    def _get(self):
        C = self._C
        nv, nf = C.c_int(0), C.c_int(0)
        ptrs = [C.c_void_p() for _ in range(4)]
        self._libmod.check(self._lib.lt_mesh_get(self._h, C.byref(nv), C.byref(nf), *[C.byref(p) for p in ptrs]),
                           "lt_mesh_get")
        return nv.value, nf.value, [p.value for p in ptrs]

    @property
    def n_verts(self):
        return self._get()[0]

    @property
    def n_faces(self):
        return self._get()[1]

    def renumber(self):
        """Number the vertices as scikit-image does -- by first use in the face stream -- (``lt_mesh_renumber_dev``): the four
        arrays then equal the reference's ``get_mesh`` arrays element for element.  Before any ``Scene.set_device_mesh`` of
        this extraction: the vertex arrays move."""
        C = self._C
        st = self._torch.cuda.current_stream(self.device)
        self._libmod.check(self._lib.lt_mesh_renumber_dev(self._h, C.c_void_p(st.cuda_stream)), "lt_mesh_renumber_dev")
        return self

    def tensors(self):
        """``(verts, faces, colors, rem)`` as zero-copy device tensors."""
        torch = self._torch
        nv, nf, (pv, pf, pc, pr) = self._get()
        def wrap(ptr, n, typestr, dt, *shape):
            if n == 0:
                return torch.empty(shape, dtype=dt, device=self.device)
            return _wrap_device_pointer(torch, ptr, n, self.device, typestr).view(*shape)

        return (wrap(pv, 3 * nv, "<f4", torch.float32, nv, 3), wrap(pf, 3 * nf, "<i4", torch.int32, nf, 3),
                wrap(pc, 3 * nv, "<i4", torch.int32, nv, 3), wrap(pr, nv, "<f4", torch.float32, nv))


def _wrap_device_pointer(torch, ptr, n, device, typestr="<f4"):
    """torch tensor over a raw device pointer (no copy, no ownership) via the CUDA array interface."""
    class _Holder:
        pass
    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (int(ptr), False), "version": 3}
    return torch.as_tensor(h, device=device)
