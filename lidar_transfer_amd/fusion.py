"""``throw_rays_at_mesh`` -- Python side of the drop-in boundary
(reference: ``TSDFVolume.throw_rays_at_mesh``, auxiliary/fusion_lidar.py:426-455).

The TSDF fusion and marching cubes that PRODUCE the mesh are consumed unchanged (out of scope,
SURVEY.md section 8f); this module only replaces what happens to the mesh afterwards.  Either call
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

    Same constructor and ``integrate`` signature; the four volumes live in HBM (``lt_tsdf``) and every
    ``integrate`` is one launch of the HIP kernel -- no per-launch image round trip, no grid loop.
    ``merge=True`` (default) is the class-aware branch the reference runs.  Marching cubes (``get_mesh``) is
    consumed unchanged from the reference / skimage and is out of scope here.
    """

    def __init__(self, vol_bnds, voxel_size, fov_up, fov_down, device=None, merge=True):
        import ctypes as C

        import torch

        from . import _lib
        self._lib = _lib.load()
        self._C, self._torch, self._libmod = C, torch, _lib
        self.fov_up, self.fov_down = fov_up, fov_down
        self.merge = merge
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
                                                           self._libmod.LT_TSDF_MERGE if self.merge else 0,
                                                           C.c_void_p(st.cuda_stream)), "lt_tsdf_integrate_dev")
        st.synchronize()  # the temporaries above must outlive the kernel

    def get_volume(self):
        """``(tsdf, color, rem)`` as float32 numpy arrays ``[dx,dy,dz]`` (fusion_lidar.py:395-400)."""
        t, w, c, r = self.get_volume_tensors()
        return t.cpu().numpy(), c.cpu().numpy(), r.cpu().numpy()

    def get_volume_tensors(self):
        """Zero-copy ``torch`` views of the four device volumes (tsdf, weight, color, rem)."""
        torch, C = self._torch, self._C
        dims = (C.c_int * 3)()
        org = (C.c_float * 3)()
        ptrs = [C.c_void_p() for _ in range(4)]
        self._libmod.check(self._lib.lt_tsdf_volumes(self._h, dims, org, *[C.byref(p) for p in ptrs]),
                           "lt_tsdf_volumes")
        n = int(dims[0]) * int(dims[1]) * int(dims[2])
        out = []
        for p in ptrs:
            out.append(_wrap_device_pointer(torch, p.value, n, self.device).view(int(dims[0]), int(dims[1]), int(dims[2])))
        return out


def _wrap_device_pointer(torch, ptr, n, device):
    """torch tensor over a raw device pointer (no copy, no ownership) via the CUDA array interface."""
    class _Holder:
        pass
    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (int(ptr), False), "version": 3}
    return torch.as_tensor(h, device=device)
