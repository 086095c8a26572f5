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
