"""Scan pipeline: the reference's batch loop body (``lidar_deform.py:393-462`` -- per output scan: a mesh, one
``throw_rays_at_mesh`` call, the unpacked images) kept fed on one MI355X.

The reference renders one scan per loop iteration and waits for it.  Here the scans of a sequence are *submitted*:
:class:`ScanPipeline` owns a pool of scene handles and HIP streams, groups consecutive scans into batches of up to
8 and hands each batch to ``lt_scene_render_batch_dev`` (three kernel launches for the whole batch, DESIGN.md
section 5b), with several batches in flight so that the small clean-up kernels of one batch run under the big
kernel of the next.  All scans share ONE read-only ray set (one target sensor model).  This is the loop of
``bench.py`` as a library facility; torch is only used for device memory and streams.

    pipe = ScanPipeline(rays, H)                      # rays: [H*W, 3] f32 CUDA tensor (create_rays)
    for k, (verts, faces, colors, rem) in enumerate(meshes):     # device tensors, layouts of throw_rays_at_mesh
        pipe.submit(verts, faces, colors, rem, origin, range_out=ranges[k], label_out=labels[k])
    pipe.flush()                                      # everything submitted so far is in `ranges` / `labels`
"""
from __future__ import annotations

import ctypes as C

from . import _lib
from .raytracer import RaySet, Scene


class ScanPipeline:
    """Render scans with a new mesh each through batched calls, ``in_flight`` batches deep.

    A submitted scan's mesh tensors and output tensors must stay untouched until its batch has COMPLETED on the
    device: :meth:`flush`, or ``submit`` returning for a scan that reuses the same pool slot (``submit`` waits on
    the host for the batch that last used the slot -- an event recorded on the batch's stream after its render --
    before it lets go of that batch's tensors; the pipeline itself keeps them referenced until then, so the caller
    may drop its own references right after ``submit``).  Outputs that are not passed go to per-slot scratch
    tensors and are overwritten by the scan that takes the slot next.
    """

    def __init__(self, rays, H, device=None, batch=8, in_flight=2, label_image=True, write_misses=True):
        import torch
        self._torch = torch
        self.device = rays.device if device is None else torch.device(device)
        if not 1 <= batch <= 8:
            raise ValueError("batch: 1 .. 8 scans per lt_scene_render_batch_dev call")
        if in_flight < 1:
            raise ValueError("in_flight: at least one batch")
        self.batch, self.in_flight = int(batch), int(in_flight)
        self.label_image = bool(label_image)
        self._lib = _lib.load()
        self.rayset = RaySet(rays, H)  # returns a finished, read-only ray set
        self.n_rays = self.rayset.n_rays
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        n = self.batch * self.in_flight
        self._scenes = [Scene(idx) for _ in range(n)]
        self._streams = [torch.cuda.Stream(self.device) for _ in range(self.in_flight)]
        self._scratch = [self._scenes[0].alloc_outputs(self.n_rays, label_image=self.label_image) for _ in range(n)]
        self._flags = (_lib.LT_TRACE_WRITE_MISSES if write_misses else 0) | \
                      (_lib.LT_TRACE_LABEL_IMAGE if self.label_image else 0)
        self._group = 0     # batch slot the pending scans belong to
        self._pending = []  # [(origin, outputs dict)] of the batch being collected
        self._keep = [[] for _ in range(self.in_flight)]  # tensors referenced by the batch in flight per group
        # completion of the batch last launched per group: the renders run on side streams, while the tensors were
        # allocated on the caller's stream -- torch's caching allocator would hand their memory back to the caller
        # the moment the last reference goes, whether or not the side stream is done with it
        self._done = [None] * self.in_flight
        self.n_submitted = 0

    def submit(self, verts, faces, colors, rem, origin, range_out=None, label_out=None, endpoints_out=None,
               rem_out=None, tri_out=None):
        """Queue one scan: its mesh (device tensors in the reference's layouts: verts [V,3] f32, faces [F,3] i32,
        colors [V,3] i32, rem [V] f32), the ray origin (3 floats, host) and where its images go.  ``label_out`` is
        the [n_rays] int32 semantic-label image with ``label_image=True`` (``deform``'s unpack, laserscan.py:912),
        else the [n_rays, 3] colour image."""
        g, j = self._group, len(self._pending)
        slot = g * self.batch + j
        sc = self._scenes[slot]
        if j == 0:
            # the slots of this group were last used `in_flight` batches ago, on this group's stream: device-side
            # the new batch is ordered behind it (same stream), but its tensors may only be released -- and its
            # scene handles given a new mesh -- once that batch has really finished
            if self._done[g] is not None:
                self._done[g].synchronize()
                self._done[g] = None
            self._keep[g] = []
        sc.set_mesh(verts, faces, colors, rem)
        o = dict(self._scratch[slot])
        for key, t in (("range", range_out), ("endcolors", label_out), ("endpoints", endpoints_out),
                       ("endrem", rem_out), ("tri", tri_out)):
            if t is not None:
                if not t.is_cuda or not t.is_contiguous() or t.numel() != o[key].numel() or t.dtype != o[key].dtype:
                    raise ValueError(f"{key}: contiguous {o[key].dtype} CUDA tensor with {o[key].numel()} elements expected")
                o[key] = t
        self._pending.append((tuple(float(x) for x in origin), o))
        self._keep[g].append((verts, faces, colors, rem, o))
        self.n_submitted += 1
        if len(self._pending) == self.batch:
            self._launch()

    def _launch(self):
        n = len(self._pending)
        if n == 0:
            return
        g = self._group
        vp = C.c_void_p
        scenes = self._scenes[g * self.batch:g * self.batch + n]
        arr = lambda vals: (vp * n)(*vals)  # noqa: E731
        org = (C.c_float * (3 * n))(*[x for o, _ in self._pending for x in o])
        col = lambda key: arr([o[key].data_ptr() for _, o in self._pending])  # noqa: E731
        with self._torch.cuda.device(self.device):
            # the meshes were produced on the caller's stream: this batch's stream starts after it
            ev = self._torch.cuda.Event()
            ev.record(self._torch.cuda.current_stream(self.device))
            self._streams[g].wait_event(ev)
            _lib.check(self._lib.lt_scene_render_batch_dev(n, arr([s._h for s in scenes]),
                                                           arr([self.rayset._h] * n), org, col("endpoints"),
                                                           col("endcolors"), col("range"), col("endrem"), col("tri"),
                                                           self._flags, vp(self._streams[g].cuda_stream)),
                       "lt_scene_render_batch_dev")
            done = self._torch.cuda.Event()
            done.record(self._streams[g])
            self._done[g] = done
        self._pending = []
        self._group = (g + 1) % self.in_flight

    def flush(self):
        """Submit the partial batch and wait until every submitted scan's images are complete."""
        self._launch()
        for st in self._streams:
            st.synchronize()
        self._keep = [[] for _ in range(self.in_flight)]
        self._done = [None] * self.in_flight

    def status(self):
        """Raise if a mesh submitted since the last call referenced vertices outside ``[0, n_verts)``."""
        self.flush()
        for sc in self._scenes:
            sc.status()

    def close(self):
        if getattr(self, "_scenes", None):
            self.flush()
            for sc in self._scenes:
                sc.close()
            self._scenes = []
            self.rayset.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostScanPipeline:
    """The reference's loop over output scans with HOST meshes (``throw_rays_at_mesh`` -> ``C_Trace`` per scan,
    fusion_lidar.py:434-451), pipelined: ``lt_hostpipe`` keeps ``depth`` scans in flight -- scan i + 1 uploads while
    scan i renders and scan i - 1 downloads -- so that a sequence runs at the rate of the PCIe link instead of the sum
    of its phases.  numpy in, numpy out; nothing here touches torch.

        pipe = HostScanPipeline(rays, H)                       # rays: [H*W, 3] float32 (create_rays)
        for verts, faces, colors, rem in meshes:               # get_mesh's arrays; colors uint8 [V,3] or int32
            t = pipe.submit(verts, faces, colors, rem, origin)
            ...
            images = pipe.wait(t)                              # dict: endpoints, endcolors, range, endrem, tri
        pipe.close()

    Every cell of the images is written (misses: 0, ``tri`` -1), like the reference's pre-zeroed arrays hold after
    ``ctrace``.  ``label_image=True`` makes ``endcolors`` the [H*W] semantic-label image (``deform``'s unpack,
    laserscan.py:912)."""

    def __init__(self, rays, H, depth=4, label_image=False, device=None, normalize="intel"):
        import numpy as np
        self._np = np
        self._lib = _lib.load()
        rays = np.ascontiguousarray(np.asarray(rays, np.float32).reshape(-1, 3))
        self.n_rays = (rays.shape[0] // int(H)) * int(H)
        self.label_image = bool(label_image)
        flags = (_lib.LT_TRACE_LABEL_IMAGE if label_image else 0) | \
            {"intel": 0, "exact": _lib.LT_TRACE_NORM_EXACT, "amd": _lib.LT_TRACE_NORM_AMD}[normalize]
        h = C.c_void_p()
        _lib.check(self._lib.lt_hostpipe_create(C.byref(h), rays.ctypes.data_as(C.POINTER(C.c_float)), rays.shape[0],
                                                int(H), int(depth), flags, -1 if device is None else int(device)),
                   "lt_hostpipe_create")
        self._h = h
        self.depth = int(depth)
        self._live = {}  # ticket -> [inputs kept alive (dropped when the slot is reused), outputs (kept until wait())]

    def alloc_outputs(self):
        np = self._np
        R = self.n_rays
        return dict(endpoints=np.empty((R, 3), np.float32),
                    endcolors=np.empty((R,) if self.label_image else (R, 3), np.int32),
                    range=np.empty(R, np.float32), endrem=np.empty(R, np.float32), tri=np.empty(R, np.int32))

    def submit(self, verts, faces, colors, rem, origin, out=None):
        """Queue one scan; returns its ticket.  The arrays are referenced (not copied) until :meth:`wait`."""
        np = self._np
        verts = np.ascontiguousarray(verts, np.float32)
        faces = np.ascontiguousarray(faces, np.int32)
        rem = np.ascontiguousarray(rem, np.float32)
        colors = np.asarray(colors)
        u8 = colors.dtype == np.uint8
        colors = np.ascontiguousarray(colors, np.uint8 if u8 else np.int32)
        org = np.ascontiguousarray(np.asarray(origin, np.float32).reshape(-1)[:3])
        if out is None:
            out = self.alloc_outputs()
        fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)

        def p(key, typ):
            a = out.get(key)
            return a.ctypes.data_as(typ) if a is not None else None

        t = C.c_int(-1)
        rc = self._lib.lt_hostpipe_submit(self._h, org.ctypes.data_as(fp), verts.ctypes.data_as(fp),
                                          faces.ctypes.data_as(ip), colors.ctypes.data_as(C.c_void_p), int(u8),
                                          rem.ctypes.data_as(fp), verts.size // 3, faces.size // 3, p("endpoints", fp),
                                          p("endcolors", ip), p("range", fp), p("endrem", fp), p("tri", ip), C.byref(t))
        if t.value >= 0:
            self._live[t.value] = [(verts, faces, colors, rem, org), out]
            # the scan whose slot this submit reused was completed by it (its images are already in ITS output
            # arrays): the library no longer reads its inputs -- drop those, but keep the outputs until the caller
            # collects them with wait() (a caller may submit more than `depth` scans before waiting)
            old = self._live.get(t.value - self.depth)
            if old is not None:
                old[0] = None
        _lib.check(rc, "lt_hostpipe_submit")
        return t.value

    def wait(self, ticket):
        """Images of the scan with this ticket (complete when the call returns).  A ticket is handed out once: a second
        wait() for it -- or a ticket this pipe never issued -- raises ``KeyError``."""
        ticket = int(ticket)
        if ticket not in self._live:
            raise KeyError(f"HostScanPipeline.wait: ticket {ticket} is unknown or was already collected")
        _lib.check(self._lib.lt_hostpipe_wait(self._h, ticket), "lt_hostpipe_wait")
        return self._live.pop(ticket)[1]

    def flush(self):
        """Complete every submitted scan; their images stay collectable with :meth:`wait`."""
        _lib.check(self._lib.lt_hostpipe_flush(self._h), "lt_hostpipe_flush")
        for item in self._live.values():
            item[0] = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.lt_hostpipe_destroy(self._h)
            self._h = None
            self._live = {}

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
