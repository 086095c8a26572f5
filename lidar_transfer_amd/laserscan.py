"""Host-side mirror of the scan-model functions that sit on the hot path
(reference: auxiliary/laserscan.py).  Only what the path needs is here."""
from __future__ import annotations

import numpy as np


def create_rays(fov_up, fov_down, H, W):
    """Unit ray direction per (beam, azimuth) cell, ``float32 [H*W, 3]``, row-major ``h*W + w``.

    Restatement of ``MultiSemLaserScan.create_rays`` (auxiliary/laserscan.py:1092-1119), quirks
    included: ``linspace(0, 360, W)`` contains both end points, so column 0 and column W-1 are
    the same direction; ``beam_angles`` are ignored; float64 trigonometry, cast to float32 last.
    """
    yaw = np.linspace(0, 360, W) + 180
    yaw[yaw > 360] -= 360
    yaw = yaw / 180. * np.pi
    pitch = np.pi / 2 - np.linspace(fov_up, fov_down, H) / 180. * np.pi
    sp, cp = np.sin(pitch), np.cos(pitch)
    beams = np.empty((H, W, 3), dtype=np.float64)
    beams[:, :, 0] = sp[:, None] * np.cos(-yaw)[None, :]
    beams[:, :, 1] = sp[:, None] * np.sin(-yaw)[None, :]
    beams[:, :, 2] = cp[:, None] * np.ones(W)[None, :]
    return np.ascontiguousarray(beams.reshape(H * W, 3).astype(np.float32))


def create_rays_device(fov_up, fov_down, H, W, device=None, stream=None):
    """:func:`create_rays` computed by the HIP kernel into a ``torch`` tensor ``[H*W, 3] f32`` on the GPU."""
    import ctypes as C

    import torch

    from . import _lib
    lib = _lib.load()
    dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
    out = torch.empty((H * W, 3), dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream(dev) if stream is None else stream
    with torch.cuda.device(dev):
        _lib.check(lib.lt_create_rays_dev(float(fov_up), float(fov_down), int(H), int(W), out.data_ptr(),
                                          C.c_void_p(st.cuda_stream)), "lt_create_rays_dev")
    return out


class Projector:
    """Batched, synchronisation-free spherical projection on the GPU (``lt_range_projection_batch_dev``): the
    ``number_of_scans`` clouds of one output scan -- ``do_range_projection_new`` + ``do_label_projection_new`` per source scan
    in ``MultiSemLaserScan.deform`` (auxiliary/laserscan.py:874-881) -- in one launch sequence on the caller's stream, images
    out as CUDA tensors, nothing read back.  One projector per caller thread / stream (it owns the z-min workspace).

        pj = Projector()
        imgs = pj.project([(points, rem, label), ...], fov_up, fov_down, H, W, new=True, remove=True,
                          outputs=("range", "rem", "label_folded"))      # list of dicts of [H, W] CUDA tensors

    ``points`` [n,3] float32 or float64 (one dtype per call: the arithmetic follows it, as numpy's does), ``rem`` [n]
    float32 or None, ``label`` [n] int32 / uint32 or None.  Outputs: ``idx`` (numbering of the KEPT points, -1 empty),
    ``range``, ``xyz``, ``rem``, ``label``, ``color`` (needs ``color_lut``), ``mask``, ``label_folded`` (what
    ``TSDFVolume.integrate`` folds from ``proj_label3``), ``proj_x`` / ``proj_y`` / ``proj_xf`` / ``proj_yf``, ``n_kept``
    (a 1-element int32 tensor), ``bnds`` (a [3,2] float64 tensor: ``get_bnds()`` of the kept points, laserscan.py:678-681).
    Empty cells: 0 / -1 / 0 for range / rem / xyz with ``new`` (laserscan.py:362-368), -1
    everywhere for the old variant (:37-53)."""

    _IMG = {"idx": ("int32", 1), "range": ("float32", 1), "xyz": ("float32", 3), "rem": ("float32", 1), "label": ("int32", 1),
            "color": ("float32", 3), "mask": ("float32", 1), "label_folded": ("float32", 1), "proj_x": ("int32", 1),
            "proj_y": ("int32", 1), "proj_xf": (None, 1), "proj_yf": (None, 1)}

    def __init__(self, device=None):
        import ctypes as C

        import torch

        from . import _lib
        self._lib = _lib.load()
        self._torch = torch
        idx = torch.cuda.current_device() if device is None else (device.index if hasattr(device, "index") else int(device))
        self.device = torch.device("cuda", idx)
        h = C.c_void_p()
        _lib.check(self._lib.lt_projector_create(C.byref(h), idx), "lt_projector_create")
        self._h = h

    def project(self, clouds, fov_up, fov_down, H, W, new=True, remove=False, beam_angles=None, color_lut=None,
                outputs=("range", "rem", "label"), out=None, stream=None):
        import ctypes as C

        from . import _lib
        torch = self._torch
        n = len(clouds)
        if n == 0:
            return []
        dt = clouds[0][0].dtype
        if dt not in (torch.float32, torch.float64):
            raise TypeError("points: float32 or float64")
        keep, cl, im, res = [], (_lib.Cloud * n)(), (_lib.ProjImages * n)(), []
        for k, (pts, rem, lab) in enumerate(clouds):
            if pts.dtype != dt:
                raise TypeError("all clouds of one call share one dtype")
            for name, t in (("points", pts), ("remissions", rem), ("label", lab)):
                if t is not None and not (isinstance(t, torch.Tensor) and t.is_cuda and t.device == self.device):
                    raise ValueError(f"Projector.project: {name} of cloud {k} must be a CUDA tensor on {self.device}")
            pts = pts.contiguous()
            rem = rem.to(torch.float32).contiguous() if rem is not None else None
            lab = lab.to(torch.int32).contiguous() if lab is not None else None   # (uint32 labels: same bits)
            keep += [pts, rem, lab]
            cl[k].points = pts.data_ptr()
            cl[k].rem = rem.data_ptr() if rem is not None else None
            cl[k].label = lab.data_ptr() if lab is not None else None
            cl[k].n = int(pts.shape[0])
            o = dict(out[k]) if out is not None else {}
            for name in outputs:
                if name in o:
                    continue
                if name == "n_kept":
                    o[name] = torch.empty(1, dtype=torch.int32, device=self.device)
                    continue
                if name == "bnds":
                    o[name] = torch.empty((3, 2), dtype=torch.float64, device=self.device)
                    continue
                tdt, ch = self._IMG[name]
                tdt = dt if tdt is None else getattr(torch, tdt)
                o[name] = torch.empty((H, W, ch) if ch > 1 else (H, W), dtype=tdt, device=self.device)
            for name, t in o.items():
                setattr(im[k], name, t.data_ptr())
            res.append(o)
        lut = color_lut.to(torch.float32).contiguous() if color_lut is not None else None
        beams = None
        if beam_angles is not None and len(beam_angles):
            import numpy as np
            beams = np.ascontiguousarray(beam_angles, dtype=np.float64)
        st = torch.cuda.current_stream(self.device) if stream is None else stream
        flags = (_lib.LT_PROJ_NEW if new else 0) | (_lib.LT_PROJ_REMOVE if remove else 0)
        init = (0.0, -1.0, 0.0) if new else (-1.0, -1.0, -1.0)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.lt_range_projection_batch_dev(
                self._h, n, cl, int(dt == torch.float64), float(fov_up), float(fov_down), int(H), int(W),
                beams.ctypes.data_as(C.c_void_p) if beams is not None else None, 0 if beams is None else len(beams), flags,
                lut.data_ptr() if lut is not None else None, 0 if lut is None else int(lut.shape[0]), im, *init,
                C.c_void_p(st.cuda_stream)), "lt_range_projection_batch_dev")
        self._keep = (keep, lut)  # inputs stay referenced until the next call (the kernels are queued, not finished)
        return res

    def close(self):
        if getattr(self, "_h", None):
            self._lib.lt_projector_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LaserScan:
    """Spherical projection part of the reference's ``LaserScan`` (auxiliary/laserscan.py:14-292),
    computed by ``liblidarhip.so`` (``lt_range_projection``, HIP atomic z-min).

    Same attribute names, dtypes and "no data" values as the reference's ``reset()``
    (laserscan.py:28-53).  File I/O, pose handling and visualisation are out of scope: set
    ``points`` ([N,3] float32 or float64), ``remissions`` ([N] float32) directly.
    """

    def __init__(self, H, W, transformation=None, beam_angles=None):
        self.proj_H = H
        self.proj_W = W
        self.beam_angles = beam_angles
        self.reset()

    def reset(self):
        H, W = self.proj_H, self.proj_W
        self.points = np.zeros((0, 3), dtype=np.float32)
        self.remissions = np.zeros((0,), dtype=np.float32)
        self.proj_range = np.full((H, W), -1, dtype=np.float32)
        self.proj_xyz = np.full((H, W, 3), -1, dtype=np.float32)
        self.proj_remissions = np.full((H, W), -1, dtype=np.float32)
        self.proj_idx = np.full((H, W), -1, dtype=np.int32)
        self.proj_x = np.zeros((0, 1), dtype=np.float32)
        self.proj_y = np.zeros((0, 1), dtype=np.float32)
        self.unproj_range = np.zeros((0, 1), dtype=np.float32)
        self.proj_mask = np.zeros((H, W), dtype=np.int32)
        self.label = np.zeros((0,), dtype=np.uint32)
        self.label_color = np.zeros((0, 3), dtype=np.float32)

    def size(self):
        return self.points.shape[0]

    def __len__(self):
        return self.size()

    # ---- shared driver ---------------------------------------------------------------------------------
    def _project(self, fov_up, fov_down, flags, range_init, rem_init, xyz_init, color_lut=None):
        import ctypes as C

        from . import _lib
        lib = _lib.load()
        H, W = self.proj_H, self.proj_W
        pts = np.ascontiguousarray(self.points)
        if pts.dtype not in (np.float32, np.float64):
            pts = pts.astype(np.float64)
        is64 = pts.dtype == np.float64
        n = pts.shape[0]
        rem = np.ascontiguousarray(self.remissions, dtype=np.float32) if len(self.remissions) == n else None
        lab = np.ascontiguousarray(self.label, dtype=np.uint32) if len(getattr(self, "label", ())) == n else None
        lut = np.ascontiguousarray(color_lut, dtype=np.float32) if color_lut is not None else None
        beams = np.ascontiguousarray(self.beam_angles, dtype=np.float64) if self.beam_angles else None
        ft = pts.dtype
        o = dict(points=np.empty((n, 3), ft), rem=np.empty(n, np.float32), label=np.empty(n, np.uint32),
                 depth=np.empty(n, ft), px=np.empty(n, np.int32), py=np.empty(n, np.int32), xf=np.empty(n, ft),
                 yf=np.empty(n, ft), idx=np.empty((H, W), np.int32), range=np.empty((H, W), np.float32),
                 xyz=np.empty((H, W, 3), np.float32), remi=np.empty((H, W), np.float32),
                 labi=np.empty((H, W), np.int32), col=np.empty((H, W, 3), np.float32),
                 mask=np.empty((H, W), np.float32))
        vp = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None  # noqa: E731
        kept = C.c_int(0)
        rc = lib.lt_range_projection(vp(pts), int(is64), vp(rem), vp(lab), n, float(fov_up), float(fov_down), H, W,
                                     vp(beams), 0 if beams is None else len(beams), flags, vp(lut),
                                     0 if lut is None else lut.shape[0], vp(o["points"]),
                                     vp(o["rem"]) if rem is not None else None,
                                     vp(o["label"]) if lab is not None else None, vp(o["depth"]), vp(o["px"]),
                                     vp(o["py"]), vp(o["xf"]), vp(o["yf"]), vp(o["idx"]), vp(o["range"]), vp(o["xyz"]),
                                     vp(o["remi"]), vp(o["labi"]), vp(o["col"]), vp(o["mask"]), range_init, rem_init,
                                     xyz_init, C.byref(kept))
        _lib.check(rc, "lt_range_projection")
        k = kept.value
        # what remove_points (laserscan.py:139-148) leaves behind
        self.points = o["points"][:k]
        if rem is not None:
            self.remissions = o["rem"][:k]
        if lab is not None:
            self.label = o["label"][:k]
            if hasattr(self, "color_lut") and len(self.label_color) == n:
                self.label_color = self.color_lut[self.label].reshape((-1, 3))  # colorize() of the kept points
        for key in ("depth", "px", "py", "xf", "yf"):
            o[key] = o[key][:k]
        return o

    def do_range_projection(self, fov_up, fov_down, remove=False):
        """Mirror of laserscan.py:202-292 (closest point per cell; lowest index among equal depths)."""
        from . import _lib
        o = self._project(fov_up, fov_down, _lib.LT_PROJ_REMOVE if remove else 0, -1.0, -1.0, -1.0)
        self.unproj_range = o["depth"].copy()
        self.depth = o["depth"]
        self.proj_range = o["range"]
        self.proj_xyz = o["xyz"]
        self.proj_remissions = o["remi"]
        self.proj_idx = o["idx"]
        self.proj_x = o["px"]
        self.proj_y = o["py"]
        self.proj_mask = o["mask"]
        self._last = o

    def do_range_projection_new(self, fov_up, fov_down, remove=False, method="depth"):
        """Mirror of laserscan.py:294-391, ``method="depth"`` (the only branch any caller reaches)."""
        if method != "depth":
            raise ValueError("only method='depth' is on the reference's call paths (laserscan.py:841, :952)")
        from . import _lib
        lut = getattr(self, "color_lut", None)
        o = self._project(fov_up, fov_down, _lib.LT_PROJ_NEW | (_lib.LT_PROJ_REMOVE if remove else 0), 0.0, -1.0, 0.0,
                          color_lut=lut)
        self.index = o["idx"]
        self.range_image = o["range"]
        self.proj_range = self.range_image
        self.proj_remissions = o["remi"]
        self.label_image = o["labi"].astype(np.float64)[:, :, None]
        self.label_color_image = o["col"].astype(np.float64)
        n = o["px"].shape[0]
        if n:  # numpy fancy indexing with -1 wraps to the last point (laserscan.py:384-388)
            self.proj_y = o["py"][self.index]
            self.proj_x = o["px"][self.index]
            self.proj_y_float = o["yf"][self.index]
            self.proj_x_float = o["xf"][self.index]
        self.unproj_range = o["depth"].copy()
        self._last = o


class SemLaserScan(LaserScan):
    """Label part of the reference's ``SemLaserScan`` (auxiliary/laserscan.py:537-676)."""

    def __init__(self, H, W, nclasses, color_dict=None, transformation=None, beam_angles=None):
        super().__init__(H, W, transformation, beam_angles)
        self.nclasses = nclasses
        self.color_dict = color_dict or {}
        max_key = max([k + 1 for k in self.color_dict] + [0])
        self.color_lut = np.zeros((max_key + 100, 3), dtype=np.float32)
        for key, value in self.color_dict.items():
            self.color_lut[key] = np.array(value, np.float32) / 255.0
        self.proj_label = np.zeros((H, W), dtype=np.int32)
        self.proj_color = np.zeros((H, W, 3), dtype=np.float64)

    def colorize(self):
        self.label_color = self.color_lut[self.label].reshape((-1, 3))

    def do_label_projection(self):
        """laserscan.py:645-649 -- the label/colour of the winning point was gathered by the resolve kernel."""
        o = self._last
        mask = self.proj_idx >= 0
        self.proj_label = np.where(mask, o["labi"], 0).astype(np.int32)
        self.proj_color = np.where(mask[:, :, None], self._color_of(o, mask), 0.0)

    def do_label_projection_new(self):
        """laserscan.py:672-676."""
        o = self._last
        mask = self.index >= 0
        self.proj_label = np.where(mask, o["labi"], 0).astype(np.int32)
        self.proj_color = np.where(mask[:, :, None], self._color_of(o, mask), 0.0)

    def _color_of(self, o, mask):
        """``color_lut[label[...]]`` of the occupied cells; like the reference's fancy indexing (laserscan.py:649, :676)
        a label beyond the look-up table raises IndexError instead of being mapped to some other class."""
        lab = np.asarray(o["labi"])
        n = self.color_lut.shape[0]
        bad = mask & ((lab < 0) | (lab >= n))
        if bad.any():
            raise IndexError(f"index {int(lab[bad].reshape(-1)[0])} is out of bounds for axis 0 with size {n}")
        return self.color_lut[np.where(mask, lab, 0)].astype(np.float64)
