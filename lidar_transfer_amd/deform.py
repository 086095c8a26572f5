"""``MultiSemLaserScan.deform`` + ``write`` from point clouds without leaving HBM.

The loop body of the reference per output scan (lidar_deform.py:393-462) for its two adaptions, composed from the
library's device entry points on ONE HIP stream:

``mesh``  (auxiliary/laserscan.py:863-918, :1121-1178)
    per source scan ``do_range_projection_new(fov, remove=True)`` + ``do_label_projection_new`` (:874-881)
    -> fresh ``TSDFVolume``, ``integrate(proj_label3, proj_range, proj_remissions)`` per scan (:886-897)
    -> ``throw_rays_at_mesh`` = marching cubes + ray cast of the target sensor (:899-907), unpack (:909-914)
    -> ``write``: filter + pack ``velodyne/N.bin`` / ``labels/N.label`` (:1133-1178)
    =  ``lt_range_projection_batch_dev`` -> ``lt_fusion_scan_dev`` -> ``lt_pack_scan_dev``

``mergemesh``  (:921-1012, the adaption config/lidar_transfer.yaml selects)
    the scans merged into ONE cloud (:939-949), ``do_range_projection_new`` with the TARGET field of view onto the SOURCE
    H x W (:929-931, :952-954), ``vol_bnds`` clipped IN PLACE by the rounded bounds of the kept points (:957-962), a fresh
    ``TSDFVolume`` of the TARGET field of view (:968), one ``integrate``, rays of the target sensor, ray cast, ``write``
    =  ``lt_range_projection_batch_dev`` (+ the bounds, one 48-byte read-back) -> ``lt_fusion_scan_dev`` -> ``lt_pack_scan_dev``

``cp``    (:827-861, :1121-1178)
    the merged cloud through ``do_range_projection_new(target fov, remove=True)`` + ``do_label_projection_new`` +
    ``do_reverse_projection_new`` (:841-845), then ``write`` with ``index > 0`` (:1133-1144)
    =  ``lt_range_projection_batch_dev`` -> ``lt_reverse_projection_dev`` -> ``lt_pack_scan_dev``

The clouds are CUDA tensors in the frame the reference projects them in (pose handling and file I/O are out of scope,
DESIGN.md section 1); the host is touched twice per output scan: the mesh sizes inside marching cubes and the number
of packed points.  There is no CPU path: everything ends in ``liblidarhip.so``.
"""
from __future__ import annotations

import ctypes as C

from . import _lib
from .raytracer import RaySet, Scene


class DeviceDeform:
    """One chain of device objects (projector, TSDF volume, mesh, scene, ray set of the target sensor) that turns the
    source scans of one output scan into the target sensor's scan.

        dd = DeviceDeform(source=(H, W, fov_up, fov_down), target=(t_H, t_W, t_fov_up, t_fov_down),
                          vol_bnds=[[-50, 50], [-50, 50], [-5, 5]], voxel_size=0.05)
        out = dd.mesh([(points, rem, label), ...])     # CUDA tensors, one triple per source scan
        out["bin"], out["label_file"]                  # [N,4] f32 / [N] i32: the bytes of velodyne/N.bin, labels/N.label
        out = dd.cp([(points, rem, label), ...])       # the scans of the merged cloud (concatenated here)
    """

    def __init__(self, source, target, vol_bnds=None, voxel_size=0.1, beam_angles=None, t_beam_angles=None,
                 preserve_float=False, device=None, merge=True, fusion="cuda", mesh_volume=True):
        """``fusion``: ``"cuda"`` -- the arithmetic of the reference's CUDA kernel (class-aware with ``merge``), or ``"numpy"`` --
        that of its numpy branch (``FUSION_GPU_MODE == 0``, fusion_lidar.py:290-388; what goldens F13 / F14 are made by).
        ``vol_bnds``: [3,2]; for :meth:`mergemesh` it is STATE, clipped in place call after call exactly as the reference
        clips the one ``voxel_bounds`` array it hands every ``MultiSemLaserScan`` (lidar_deform.py:321-401,
        laserscan.py:960-962, fusion_lidar.py:36) -- pass a numpy array to see it.  ``mesh_volume=False``: do not allocate the
        fixed volume of :meth:`mesh` (a caller that only runs ``mergemesh``)."""
        import numpy as np
        import torch

        from .fusion import DeviceMesh, TSDFVolume
        from .laserscan import Projector, create_rays_device
        self._torch = torch
        self._lib = _lib.load()
        idx = torch.cuda.current_device() if device is None else int(device)
        self.device = torch.device("cuda", idx)
        self.H, self.W, self.fov_up, self.fov_down = int(source[0]), int(source[1]), float(source[2]), float(source[3])
        self.t_H, self.t_W, self.t_fov_up, self.t_fov_down = int(target[0]), int(target[1]), float(target[2]), float(target[3])
        self.beam_angles = sorted(beam_angles) if beam_angles else None      # laserscan.py:732-735
        # the reference reads the TARGET's beam angles from the source config (laserscan.py:744)
        self.t_beam_angles = sorted(t_beam_angles) if t_beam_angles else self.beam_angles
        self.preserve_float = bool(preserve_float)
        self.projector = Projector(idx)
        self._merge = _lib.LT_TSDF_MERGE if merge else 0
        self.vol = self.mesh_obj = self.scene = self.rayset = None
        self._merge_flag, self._fusion, self._voxel_size, self._idx = bool(merge), fusion, voxel_size, idx
        self._mm_vols = {}   # mergemesh: geometry -> TSDFVolume (the clipped bounds settle after a few output scans)
        self.vol_bnds = None
        if vol_bnds is not None:
            self.vol_bnds = vol_bnds if isinstance(vol_bnds, np.ndarray) else np.array(vol_bnds)
            if self.vol_bnds.shape != (3, 2):
                raise ValueError("DeviceDeform: vol_bnds is [3, 2] (rows x, y, z; columns min, max)")
            # the volume is the SOURCE sensor's (laserscan.py:886-887), the rays the TARGET's (:899-900)
            if mesh_volume:
                self.vol = TSDFVolume(self.vol_bnds, voxel_size, self.fov_up, self.fov_down, device=idx, merge=merge, mode=fusion)
                self._merge = self.vol._flags
            self.mesh_obj = DeviceMesh(idx)
            self.scene = Scene(idx)
            rays = create_rays_device(self.t_fov_up, self.t_fov_down, self.t_H, self.t_W, device=idx)
            self.rayset = RaySet(rays, self.t_H)
            self._rays = rays
        self.n_rays = self.t_H * self.t_W

    def _stream(self):
        """The caller's current stream -- ordered behind the stream of the previous call when that was another one: the
        projector re-arms its z-min keys, and the volume / mesh / scene are reused, in stream order."""
        torch = self._torch
        st = torch.cuda.current_stream(self.device)
        last = getattr(self, "_last_stream", None)
        if last is not None and last.cuda_stream != st.cuda_stream:
            ev = torch.cuda.Event()
            ev.record(last)
            st.wait_event(ev)
        self._last_stream = st
        return st

    # ---- write(): filter + pack (laserscan.py:1133-1178) -------------------------------------------------------------
    def _pack(self, points, is_f64, rem, label, index, n, st):
        torch = self._torch
        out_bin = torch.empty((n, 4), dtype=torch.float32, device=self.device)
        out_lab = torch.empty((n,), dtype=torch.int32, device=self.device)
        kept = C.c_int(0)
        _lib.check(self._lib.lt_pack_scan_dev(points.data_ptr(), int(is_f64), rem.data_ptr(), label.data_ptr(),
                                              index.data_ptr() if index is not None else None, n, out_bin.data_ptr(),
                                              out_lab.data_ptr(), C.byref(kept), C.c_void_p(st.cuda_stream)),
                   "lt_pack_scan_dev")
        return out_bin[:kept.value], out_lab[:kept.value]

    # ---- deform('mesh') + write ---------------------------------------------------------------------------------------
    def mesh(self, clouds, origin=(0.0, 0.0, 0.0), pack=True, timing=None):
        """``clouds``: one (points [n,3] f32|f64, remissions [n] f32, label [n] i32) triple of CUDA tensors per source scan,
        already in the primary scan's frame (laserscan.py:876-879).  Returns the target scan: ``range`` / ``rem`` [t_H,t_W]
        f32, ``label`` [t_H,t_W] i32 (``label_image``, :912), ``endpoints`` [t_H*t_W,3] f32 (``back_points``), ``tri``, the
        source images per scan under ``source``, and -- with ``pack`` -- ``bin`` [N,4] f32 + ``label_file`` [N] i32, the
        bytes ``write`` puts into ``velodyne/N.bin`` and ``labels/N.label``.  ``timing``: a list that receives CUDA events
        (start, projected, rendered, packed) when given."""
        if self.vol is None:
            raise RuntimeError("DeviceDeform.mesh: constructed without vol_bnds")
        torch, lib = self._torch, self._lib
        st = self._stream()
        n = len(clouds)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if timing is not None else None
        if ev:
            ev[0].record(st)
        src = self.projector.project(clouds, self.fov_up, self.fov_down, self.H, self.W, new=True, remove=True,
                                     beam_angles=self.beam_angles, outputs=("range", "rem", "label_folded"), stream=st)
        if ev:
            ev[1].record(st)
        vp = C.c_void_p
        cp, dp, rp = (vp * max(n, 1))(), (vp * max(n, 1))(), (vp * max(n, 1))()
        for k, o in enumerate(src):
            cp[k], dp[k], rp[k] = o["label_folded"].data_ptr(), o["range"].data_ptr(), o["rem"].data_ptr()
        out = self.scene.alloc_outputs(self.n_rays, label_image=True)
        org = (C.c_float * 3)(*[float(x) for x in origin])
        flags = _lib.LT_TRACE_WRITE_MISSES | _lib.LT_TRACE_LABEL_IMAGE
        with torch.cuda.device(self.device):
            _lib.check(lib.lt_fusion_scan_dev(self.vol._h, self.mesh_obj._h, self.scene._h, self.rayset._h, n, cp, dp, rp,
                                              self.H, self.W, 1.0, self._merge, org, out["endpoints"].data_ptr(),
                                              out["endcolors"].data_ptr(), out["range"].data_ptr(),
                                              out["endrem"].data_ptr(), out["tri"].data_ptr(), flags, vp(st.cuda_stream), 0),
                       "lt_fusion_scan_dev")
            if ev:
                ev[2].record(st)
            res = dict(range=out["range"].view(self.t_H, self.t_W), rem=out["endrem"].view(self.t_H, self.t_W),
                       label=out["endcolors"].view(self.t_H, self.t_W), endpoints=out["endpoints"], tri=out["tri"],
                       source=src, n_verts=self.mesh_obj.n_verts, n_faces=self.mesh_obj.n_faces)
            if pack:  # adaption != 'cp': no index filter (laserscan.py:1145-1148)
                res["bin"], res["label_file"] = self._pack(out["endpoints"], False, out["endrem"], out["endcolors"], None,
                                                           self.n_rays, st)
            if ev:
                ev[3].record(st)
                timing.append(ev)
        return res

    # ---- deform('mergemesh') + write ------------------------------------------------------------------------------------
    def _mergemesh_volume(self, merged_bnds):
        """laserscan.py:957-969 + fusion_lidar.py:33-37 on ``self.vol_bnds`` -- the same numpy statements on the same array,
        in place -- and the device volume of the resulting geometry (kept per geometry: the bounds only ever shrink)."""
        import numpy as np

        from .fusion import TSDFVolume
        vb = self.vol_bnds
        mb = np.rint(merged_bnds).astype(int)                           # :957
        # (no short cut for "nothing clipped": with float bounds fusion_lidar.py:34-36 re-derives the upper bounds from
        # ceil((max - min) / voxel) every scan, and that can GROW the volume by a voxel per call -- the statements are run as they are)
        vb[:, 0] = np.maximum(vb[:, 0], mb[:, 0])                       # :961
        vb[:, 1] = np.minimum(vb[:, 1], mb[:, 1])                       # :962
        as_given = np.array(vb, dtype=np.float64)
        dim = np.ceil((vb[:, 1] - vb[:, 0]) / self._voxel_size).copy(order='C').astype(int)   # fusion_lidar.py:34-35
        if (dim <= 0).any():
            raise RuntimeError(f"DeviceDeform.mergemesh: the clipped volume is empty (bounds {vb.tolist()})")
        vb[:, 1] = vb[:, 0] + dim * self._voxel_size                    # fusion_lidar.py:36 (an int array truncates, as there)
        key = (tuple(float(x) for x in as_given.reshape(-1)),)
        vol = self._mm_vols.pop(key, None)
        if vol is None:
            while len(self._mm_vols) >= 3:
                self._mm_vols.pop(next(iter(self._mm_vols))).close()
            vol = TSDFVolume(as_given, self._voxel_size, self.t_fov_up, self.t_fov_down, device=self._idx,
                             merge=self._merge_flag, mode=self._fusion)   # (3) the TARGET field of view (:968-969)
            assert tuple(int(x) for x in vol._vol_dim) == tuple(int(x) for x in dim)
        self._mm_vols[key] = vol                                        # most recently used last
        return vol

    def mergemesh(self, clouds, origin=(0.0, 0.0, 0.0), pack=True):
        """``clouds``: the (points, remissions, label) CUDA triples of the source scans, already in the primary scan's frame
        (``apply_inv_pose``, laserscan.py:949: pose handling is out of scope).  Returns what :meth:`mesh` returns -- the
        target scan's ``range`` / ``rem`` / ``label`` images, ``endpoints``, ``tri``, the merged cloud's source image under
        ``source``, ``bin`` / ``label_file`` with ``pack`` -- plus ``vol_dim`` / ``vol_origin`` of this scan's volume.  One
        read-back of 48 bytes (the kept points' bounds decide the volume's geometry) besides those of :meth:`mesh`."""
        if self.vol_bnds is None:
            raise RuntimeError("DeviceDeform.mergemesh: constructed without vol_bnds")
        torch, lib = self._torch, self._lib
        st = self._stream()
        pts = torch.cat([c[0] for c in clouds]) if len(clouds) != 1 else clouds[0][0]
        rem = torch.cat([c[1] for c in clouds]) if len(clouds) != 1 else clouds[0][1]
        lab = torch.cat([c[2] for c in clouds]) if len(clouds) != 1 else clouds[0][2]
        # (1) + (2): the SOURCE image size and beam angles, the TARGET field of view (laserscan.py:929-931, :952-954)
        src = self.projector.project([(pts, rem, lab)], self.t_fov_up, self.t_fov_down, self.H, self.W, new=True, remove=True,
                                     beam_angles=self.beam_angles, outputs=("range", "rem", "label_folded", "bnds"),
                                     stream=st)[0]
        # (everything the host can prepare goes before the read-back: the GPU idles from the read-back to the next launch)
        vp = C.c_void_p
        cp, dp, rp = (vp * 1)(src["label_folded"].data_ptr()), (vp * 1)(src["range"].data_ptr()), (vp * 1)(src["rem"].data_ptr())
        out = self.scene.alloc_outputs(self.n_rays, label_image=True)
        org = (C.c_float * 3)(*[float(x) for x in origin])
        flags = _lib.LT_TRACE_WRITE_MISSES | _lib.LT_TRACE_LABEL_IMAGE
        bnds = src["bnds"].cpu().numpy()      # (synchronises the stream)
        if not (bnds[:, 0] <= bnds[:, 1]).all():
            raise ValueError("DeviceDeform.mergemesh: no point survives the projection (numpy: zero-size array to amin)")
        vol = self._mergemesh_volume(bnds)
        with torch.cuda.device(self.device):
            _lib.check(lib.lt_fusion_scan_dev(vol._h, self.mesh_obj._h, self.scene._h, self.rayset._h, 1, cp, dp, rp,
                                              self.H, self.W, 1.0, vol._flags, org, out["endpoints"].data_ptr(),
                                              out["endcolors"].data_ptr(), out["range"].data_ptr(),
                                              out["endrem"].data_ptr(), out["tri"].data_ptr(), flags, vp(st.cuda_stream), 0),
                       "lt_fusion_scan_dev")
            res = dict(range=out["range"].view(self.t_H, self.t_W), rem=out["endrem"].view(self.t_H, self.t_W),
                       label=out["endcolors"].view(self.t_H, self.t_W), endpoints=out["endpoints"], tri=out["tri"],
                       source=src, n_verts=self.mesh_obj.n_verts, n_faces=self.mesh_obj.n_faces,
                       vol_dim=tuple(int(x) for x in vol._vol_dim), vol_origin=vol._vol_origin.copy(), volume=vol)
            if pack:
                res["bin"], res["label_file"] = self._pack(out["endpoints"], False, out["endrem"], out["endcolors"], None,
                                                           self.n_rays, st)
        return res

    # ---- deform('cp') + write -----------------------------------------------------------------------------------------
    def cp(self, clouds, pack=True):
        """Closest point: the source scans merged into one cloud (laserscan.py:834-839), projected into the TARGET image
        (:841-843), re-projected to points (:844-845) and written (:1133-1160).  Returns ``range``, ``rem``, ``label``,
        ``index`` images, ``back_points`` [t_H*t_W,3] f64 and -- with ``pack`` -- ``bin`` / ``label_file``.
        The reference's ``cp`` path always holds float64 points (``apply_pose``, laserscan.py:98-104) and goldens F12 / F12b pin
        that; float32 clouds with ``preserve_float`` are re-projected in float64 here as well (numpy would stay in float32)."""
        torch, lib = self._torch, self._lib
        st = self._stream()
        pts = torch.cat([c[0] for c in clouds]) if len(clouds) != 1 else clouds[0][0]
        rem = torch.cat([c[1] for c in clouds]) if len(clouds) != 1 else clouds[0][1]
        lab = torch.cat([c[2] for c in clouds]) if len(clouds) != 1 else clouds[0][2]
        pf = self.preserve_float
        outs = ("idx", "range", "rem", "label") + (("proj_xf", "proj_yf") if pf else ("proj_x", "proj_y"))
        o = self.projector.project([(pts, rem, lab)], self.t_fov_up, self.t_fov_down, self.t_H, self.t_W, new=True,
                                   remove=True, beam_angles=self.t_beam_angles, outputs=outs, stream=st)[0]
        px, py = (o["proj_xf"], o["proj_yf"]) if pf else (o["proj_x"], o["proj_y"])
        if pf and px.dtype != torch.float64:
            px, py = px.double(), py.double()
        back = torch.empty((self.n_rays, 3), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(lib.lt_reverse_projection_dev(o["range"].data_ptr(), px.data_ptr(), py.data_ptr(), int(pf),
                                                     self.t_fov_up, self.t_fov_down, self.t_H, self.t_W, back.data_ptr(),
                                                     C.c_void_p(st.cuda_stream)), "lt_reverse_projection_dev")
            res = dict(range=o["range"], rem=o["rem"], label=o["label"], index=o["idx"], back_points=back)
            if pack:
                res["bin"], res["label_file"] = self._pack(back, True, o["rem"].view(-1), o["label"].view(-1),
                                                           o["idx"].view(-1), self.n_rays, st)
        return res

    @staticmethod
    def write(out, out_dir, idx):
        """``MultiSemLaserScan.write(out_dir, idx)`` (laserscan.py:1162-1178) for a result of :meth:`mesh` / :meth:`cp` that
        was packed: ``velodyne/NNNNNN.bin`` ([N,4] float32 x, y, z, remission) and ``labels/NNNNNN.label`` ([N] uint32) --
        the bytes the reference's per-point ``struct.pack`` loops write.  Returns the number of points."""
        import os
        if "bin" not in out or "label_file" not in out:
            raise ValueError("DeviceDeform.write: the result was produced with pack=False")
        os.makedirs(os.path.join(out_dir, "velodyne"), exist_ok=True)
        os.makedirs(os.path.join(out_dir, "labels"), exist_ok=True)
        b = out["bin"].cpu().numpy()
        b.tofile(os.path.join(out_dir, "velodyne", str(idx).zfill(6) + ".bin"))
        out["label_file"].cpu().numpy().view("uint32").tofile(os.path.join(out_dir, "labels", str(idx).zfill(6) + ".label"))
        return int(b.shape[0])

    def close(self):
        for v in getattr(self, "_mm_vols", {}).values():
            v.close()
        self._mm_vols = {}
        for name in ("rayset", "scene", "mesh_obj", "vol", "projector"):
            obj = getattr(self, name, None)
            if obj is not None:
                obj.close()
                setattr(self, name, None)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
