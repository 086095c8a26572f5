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
        self._live = {}  # ticket -> [inputs kept alive (dropped when the slot is reused), outputs (kept until wait()), own]

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
        own = out is None  # outputs allocated here must stay collectable; the caller's own arrays need no keeping
        if own:
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
            self._live[t.value] = [(verts, faces, colors, rem, org), out, own]
            # the scan whose slot this submit reused was completed by it (its images are already in ITS output
            # arrays): the library no longer reads its inputs -- drop those, but keep the outputs until the caller
            # collects them with wait() (a caller may submit more than `depth` scans before waiting).  Entries whose
            # outputs are the CALLER's own arrays are dropped by flush(): a caller who passes `out` and never calls
            # wait() does not grow this table beyond the scans between two flushes
            old = self._live.get(t.value - self.depth)
            if old is not None:
                old[0] = None
        _lib.check(rc, "lt_hostpipe_submit")
        return t.value

    def wait(self, ticket):
        """Images of the scan with this ticket (complete when the call returns).  A ticket is handed out once: a second
        wait() for it, a ticket this pipe never issued, or a ticket submitted with the caller's own ``out`` arrays and
        completed by a flush() since (those are forgotten there) raises ``KeyError``."""
        ticket = int(ticket)
        if ticket not in self._live:
            raise KeyError(f"HostScanPipeline.wait: ticket {ticket} is unknown or was already collected")
        _lib.check(self._lib.lt_hostpipe_wait(self._h, ticket), "lt_hostpipe_wait")
        return self._live.pop(ticket)[1]

    def flush(self):
        """Complete every submitted scan.  Images the pipe allocated stay collectable with :meth:`wait`; scans submitted
        with the caller's own ``out`` arrays are complete in those arrays and forgotten here."""
        _lib.check(self._lib.lt_hostpipe_flush(self._h), "lt_hostpipe_flush")
        for t in [t for t, item in self._live.items() if not item[2]]:
            del self._live[t]
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


def _fusion_chain_worker(pipe_ref, q):
    """Thread body of one chain of :class:`FusionScanPipeline`: takes jobs from the chain's queue until it receives
    ``None``.  Holds the pipeline only while a job runs (weak reference otherwise)."""
    first = True
    while True:
        job = q.get()
        if job is None:
            return
        pipe = pipe_ref()
        if pipe is None:
            return
        if first:
            pipe._torch.cuda.set_device(pipe.device)
            first = False
        ch = next(c for c in pipe._chains_all if c["q"] is q)
        pipe._run_job(ch, job)
        del pipe, ch


class FusionScanPipeline:
    """The reference's ``mesh`` adaption per output scan -- fuse ``number_of_scans`` range images into a fresh TSDF volume
    (laserscan.py:874-903), ``get_mesh`` (fusion_lidar.py:403-424), ``throw_rays_at_mesh`` (:426-455) -- entirely in HBM
    and with ``chains`` output scans IN FLIGHT: output scans are independent (own volume, mesh and image,
    lidar_deform.py:393-462), the chain's kernels are mostly sparse sweeps that leave the chip half empty (DESIGN.md
    section 7c), and 12.8 GB per default volume is nothing in 288 GB.  Every chain owns a :class:`TSDFVolume`, a
    :class:`DeviceMesh`, a :class:`Scene`, a HIP stream and a host thread; scans are dealt to the chains in turn.

        pipe = FusionScanPipeline(vol_bnds, voxel_size, fov_up, fov_down, rays, H)   # source fov; target rays [H*W,3] CUDA
        t = pipe.submit([(color_im, depth_im, rem_im), ...], origin)   # CUDA tensors; color_im [h,w,3] or folded [h,w]
        ...
        out = pipe.wait(t)     # dict of CUDA tensors: endpoints, endcolors, range, endrem, tri; 'n_verts', 'n_faces'
        pipe.close()

    The observation tensors must be COMPLETE when a chain reads them: by default ``submit`` waits (on the host) for the
    caller's current stream; a caller whose images are finished anyway passes ``inputs_ready=True``.  The tensors are kept
    referenced until the scan is done; the images are complete when ``wait`` returns.  A scan is ONE native call
    (``lt_fusion_scan_dev``) on its chain's thread, so the interpreter lock is free while the GPU works."""

    def __init__(self, vol_bnds, voxel_size, fov_up, fov_down, rays, H, chains=3, device=None, merge=True,
                 label_image=False, source_hw=None, beam_angles=None, fixed_volume=True):
        import queue
        import threading
        import weakref

        import torch

        from .fusion import DeviceMesh, TSDFVolume
        if chains < 1:
            raise ValueError("chains: at least one")
        self._torch = torch
        self._lib = _lib.load()
        self.device = rays.device if device is None else torch.device("cuda", device)
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        self.rayset = RaySet(rays, H)  # one read-only ray set for all chains
        self.n_rays = self.rayset.n_rays
        self.label_image = bool(label_image)
        self._flags = _lib.LT_TRACE_WRITE_MISSES | (_lib.LT_TRACE_LABEL_IMAGE if label_image else 0)
        self._merge = _lib.LT_TSDF_MERGE if merge else 0
        # submit_clouds(): the source sensor's image shape (H, W) the clouds are projected into (laserscan.py:874-881); the
        # field of view is the volume's.  Every chain then owns a Projector (its z-min workspace) as well.
        self._src_hw = tuple(int(x) for x in source_hw) if source_hw is not None else None
        self._src_fov = (float(fov_up), float(fov_down))
        self._beam_angles = sorted(beam_angles) if beam_angles else None
        # submit_mergemesh(): `vol_bnds` is the sequence's STATE (kept by reference: a numpy array passed in is kept current,
        # laserscan.py:960-962), every chain gets a DeviceDeform of its own, all share one MergeMeshState and this ray set.
        # fixed_volume=False: no volume of the unclipped bounds per chain (a pipeline that only runs submit_mergemesh)
        self._mm_args = dict(vol_bnds=vol_bnds, voxel_size=voxel_size, merge=merge, H=int(H))
        self._mm_state = None
        self._mm_seq = 0
        self._chains = []
        for _ in range(int(chains)):
            ch = dict(vol=TSDFVolume(vol_bnds, voxel_size, fov_up, fov_down, device=idx, merge=merge) if fixed_volume else None,
                      mesh=DeviceMesh(idx), scene=Scene(idx), stream=torch.cuda.Stream(self.device), q=queue.Queue())
            if self._src_hw is not None:
                from .laserscan import Projector
                ch["projector"] = Projector(idx)
            # the worker holds a WEAK reference to the pipeline: a bound method as thread target would keep an un-closed
            # pipeline (and its chains' volumes: GBs of HBM each) alive for ever -- __del__ could never run
            ch["thread"] = threading.Thread(target=_fusion_chain_worker, args=(weakref.ref(self), ch["q"]), daemon=True)
            self._chains.append(ch)
        self._chains_all = list(self._chains)  # (close() empties _chains first; a worker finishing its job still finds its chain)
        self._lock = threading.Lock()
        self._done = {}    # ticket -> threading.Event
        self._result = {}  # ticket -> outputs dict | exception
        self._next = 0
        for ch in self._chains:
            ch["thread"].start()

    # ---- a chain's thread ---------------------------------------------------------------------------------------------
    def _scan(self, ch, obs, origin, out):
        torch, lib = self._torch, self._lib
        st = ch["stream"]
        keep = []
        n = len(obs)
        vp = C.c_void_p
        cp, dp, rp = (vp * max(n, 1))(), (vp * max(n, 1))(), (vp * max(n, 1))()
        with torch.cuda.stream(st):
            if out is None:
                out = ch["scene"].alloc_outputs(self.n_rays, label_image=self.label_image)
            h = w = 0
            if obs and len(obs[0]) == 5:  # ("mergemesh", points, rem, label, seq) items
                try:
                    return self._scan_mergemesh(ch, obs, origin, out)
                except BaseException:
                    self._mm_state.skip(obs[0][4])  # (a failed scan must not hold up the sequence's later scans)
                    raise
            if ch["vol"] is None:
                raise RuntimeError("FusionScanPipeline: constructed with fixed_volume=False (submit_mergemesh only)")
            if obs and len(obs[0]) == 4:  # ("clouds", points, rem, label) items: ONE native call (lt_deform_scan_dev)
                return self._scan_clouds(ch, obs, origin, out)
            for k, (color_im, depth_im, rem_im) in enumerate(obs):
                # float32 FIRST, then fold RGB into one channel -- the reference's order (fusion_lidar.py:260-264:
                # color_im.astype(np.float32), then floor(b*256*256 + g*256 + r)); folding in the caller's dtype wraps a
                # uint8 image to 0 and overflows float16
                c = color_im.to(torch.float32)
                if c.dim() == 3:
                    c = torch.floor(c[:, :, 0] * 256 * 256 + c[:, :, 1] * 256 + c[:, :, 2])
                c = c.contiguous()
                d = depth_im.to(torch.float32).contiguous()
                r = rem_im.to(torch.float32).contiguous()
                if k and (d.shape[0], d.shape[1]) != (h, w):
                    raise ValueError("observations of one output scan must have one image shape")
                h, w = d.shape[0], d.shape[1]
                keep += [c, d, r]
                cp[k], dp[k], rp[k] = c.data_ptr(), d.data_ptr(), r.data_ptr()
        org = (C.c_float * 3)(*[float(x) for x in origin])

        def p(key):
            a = out.get(key)
            return a.data_ptr() if a is not None else None
        # the whole chain of the scan in ONE native call: the interpreter lock is released for all of it.  The call
        # returns with the render QUEUED (it waits for the stream once, inside marching cubes: the mesh sizes); an event
        # behind the render is what wait() waits for, so the chain's next scan is queued while this one still renders.
        _lib.check(lib.lt_fusion_scan_dev(ch["vol"]._h, ch["mesh"]._h, ch["scene"]._h, self.rayset._h, n, cp, dp, rp, h, w,
                                          1.0, self._merge, org, p("endpoints"), p("endcolors"), p("range"), p("endrem"),
                                          p("tri"), self._flags, vp(st.cuda_stream), 0), "lt_fusion_scan_dev")
        done = torch.cuda.Event()
        done.record(st)
        res = dict(out)
        res["n_verts"], res["n_faces"] = ch["mesh"].n_verts, ch["mesh"].n_faces
        res["_done"] = (done, keep, obs)  # (temporaries and observations stay referenced until the event has passed)
        return res

    def _scan_clouds(self, ch, items, origin, out):
        """projection of the source scans + the fusion chain, one native call on this chain's stream"""
        torch, lib = self._torch, self._lib
        st = ch["stream"]
        n = len(items)
        cl = (_lib.Cloud * n)()
        keep = []
        dt = items[0][1].dtype
        for k, (_, pts, rem, lab) in enumerate(items):
            if pts.dtype != dt or dt not in (torch.float32, torch.float64):
                raise TypeError("clouds: float32 or float64 points, one dtype per output scan")
            pts = pts.contiguous()
            rem = rem.contiguous() if rem.dtype == torch.float32 else rem.to(torch.float32).contiguous()
            lab = lab.contiguous() if lab.dtype == torch.int32 else lab.to(torch.int32).contiguous()
            keep += [pts, rem, lab]
            cl[k].points, cl[k].rem, cl[k].label, cl[k].n = pts.data_ptr(), rem.data_ptr(), lab.data_ptr(), int(pts.shape[0])
        beams = None
        if self._beam_angles:
            import numpy as np
            beams = np.ascontiguousarray(self._beam_angles, dtype=np.float64)
        org = (C.c_float * 3)(*[float(x) for x in origin])

        def p(key):
            a = out.get(key)
            return a.data_ptr() if a is not None else None
        _lib.check(lib.lt_deform_scan_dev(ch["projector"]._h, ch["vol"]._h, ch["mesh"]._h, ch["scene"]._h, self.rayset._h, n, cl,
                                          int(dt == torch.float64), self._src_fov[0], self._src_fov[1], self._src_hw[0],
                                          self._src_hw[1], beams.ctypes.data_as(C.c_void_p) if beams is not None else None,
                                          0 if beams is None else len(beams), 1.0, self._merge, org, p("endpoints"),
                                          p("endcolors"), p("range"), p("endrem"), p("tri"), self._flags,
                                          C.c_void_p(st.cuda_stream), 0), "lt_deform_scan_dev")
        done = torch.cuda.Event()
        done.record(st)
        res = dict(out)
        res["n_verts"], res["n_faces"] = ch["mesh"].n_verts, ch["mesh"].n_faces
        res["_done"] = (done, keep, items)
        return res

    def _scan_mergemesh(self, ch, items, origin, out):
        """deform('mergemesh') of one output scan on this chain (DeviceDeform.mergemesh on the chain's stream): projection
        with the target field of view -> the bounds statements on the shared device state, in sequence order -> fusion
        chain on the predicted geometry -> verified against the record"""
        torch = self._torch
        dd = ch.get("deform")
        if dd is None:
            from .deform import DeviceDeform
            a = self._mm_args
            sensor_s = (self._src_hw[0], self._src_hw[1], self._src_fov[0], self._src_fov[1])
            sensor_t = (a["H"], self.n_rays // a["H"], self._src_fov[0], self._src_fov[1])
            dd = DeviceDeform(sensor_s, sensor_t, None, a["voxel_size"], beam_angles=self._beam_angles, device=self.device.index,
                              merge=a["merge"], mesh_volume=False, rayset=self.rayset, mm_state=self._mm_state)
            ch["deform"] = dd
        clouds = [(pts, rem, lab) for _, pts, rem, lab, _ in items]
        with torch.cuda.stream(ch["stream"]):
            got = dd.mergemesh(clouds, origin, pack=False, out=out, seq=items[0][4])
            done = torch.cuda.Event()
            done.record(ch["stream"])
        res = dict(out)
        for k in ("n_verts", "n_faces", "vol_dim", "vol_origin", "vol_bnds_after"):
            res[k] = got[k]
        res["_done"] = (done, [got.get("source"), got.get("_keep")], items)
        return res

    def _run_job(self, ch, job):
        ticket, obs, origin, out = job
        try:
            res = self._scan(ch, obs, origin, out)
        except BaseException as e:  # noqa: BLE001  (handed to the waiter)
            res = e
        with self._lock:
            self._result[ticket] = res
            ev = self._done[ticket]
        ev.set()

    # ---- caller's side --------------------------------------------------------------------------------------------------
    def submit(self, observations, origin=(0.0, 0.0, 0.0), out=None, inputs_ready=False):
        """Queue one output scan: ``observations`` = the (color_im, depth_im, rem_im) CUDA tensors fused into its volume,
        in order.  Returns the ticket.  ``out``: a dict like ``Scene.alloc_outputs`` returns (missing keys are not written).
        ``inputs_ready``: the caller guarantees that the observation tensors are complete (no wait for its stream)."""
        import threading
        torch = self._torch
        if not self._chains:
            raise RuntimeError("FusionScanPipeline.submit: the pipeline is closed")
        obs = [tuple(o) for o in observations]
        for o in obs:
            if len(o) != 3 or not all(isinstance(a, torch.Tensor) and a.is_cuda for a in o):
                raise ValueError("observations: (color_im, depth_im, rem_im) CUDA tensors")
        return self._submit(obs, origin, out, inputs_ready)

    def _submit(self, obs, origin, out, inputs_ready):
        import threading
        torch = self._torch
        if not inputs_ready:
            torch.cuda.current_stream(self.device).synchronize()  # (see the class docstring)
        with self._lock:
            t = self._next
            self._next += 1
            self._done[t] = threading.Event()
        self._chains[t % len(self._chains)]["q"].put((t, obs, tuple(origin), out))
        return t

    def submit_clouds(self, clouds, origin=(0.0, 0.0, 0.0), out=None, inputs_ready=False):
        """Queue one output scan from the POINT CLOUDS of its source scans: ``clouds`` = ``(points [n,3] f32|f64, remissions
        [n] f32, label [n] i32)`` CUDA tensors per source scan, in the primary scan's frame (laserscan.py:876-879).  The
        chain projects them (``lt_range_projection_batch_dev``, one launch sequence, nothing read back) and runs the fusion
        chain on the images -- ``deform('mesh')`` without ``write``; needs ``source_hw`` at construction."""
        torch = self._torch
        if self._src_hw is None:
            raise RuntimeError("FusionScanPipeline.submit_clouds: construct with source_hw=(H, W)")
        items = []
        for c in clouds:
            if len(c) != 3 or not isinstance(c[0], torch.Tensor) or not c[0].is_cuda:
                raise ValueError("clouds: (points, remissions, label) CUDA tensors")
            items.append(("clouds", c[0], c[1], c[2]))
        return self._submit(items, origin, out, inputs_ready)

    def submit_mergemesh(self, clouds, origin=(0.0, 0.0, 0.0), out=None, inputs_ready=False):
        """Queue one output scan of the reference's DEFAULT adaption (config/lidar_transfer.yaml:3; laserscan.py:921-1012): the
        source scans' clouds merged, projected with the TARGET field of view (this pipeline's ``fov_up`` / ``fov_down``) onto the
        SOURCE image (``source_hw``), ``vol_bnds`` clipped by the kept points' rounded bounds -- the statements run on the
        device in SUBMISSION order, whichever chain a scan lands on --, a volume of that geometry, one integrate, marching
        cubes, ray cast.  ``out`` must hold ``range`` / ``endrem`` / ``endcolors`` (label image) / ``endpoints`` / ``tri`` as
        ``Scene.alloc_outputs(n, label_image=True)`` returns.  The scans of ONE sequence, in order; :meth:`reset_bounds`
        between sequences.  Results additionally carry ``vol_dim`` / ``vol_origin`` / ``vol_bnds_after``."""
        torch = self._torch
        if not self._chains:
            raise RuntimeError("FusionScanPipeline.submit_mergemesh: the pipeline is closed")
        if self._src_hw is None:
            raise RuntimeError("FusionScanPipeline.submit_mergemesh: construct with source_hw=(H, W)")
        if not self.label_image:
            raise RuntimeError("FusionScanPipeline.submit_mergemesh: construct with label_image=True")
        if self._mm_state is None:
            from .deform import MergeMeshState
            self._mm_state = MergeMeshState(self._mm_args["vol_bnds"], self._mm_args["voxel_size"], self.device.index)
        items = []
        for c in clouds:
            if len(c) != 3 or not isinstance(c[0], torch.Tensor) or not c[0].is_cuda:
                raise ValueError("clouds: (points, remissions, label) CUDA tensors")
            items.append(("mergemesh", c[0], c[1], c[2], self._mm_seq))
        self._mm_seq += 1
        return self._submit(items, origin, out, inputs_ready)

    def reset_bounds(self, vol_bnds):
        """A new sequence for :meth:`submit_mergemesh`: every submitted scan is completed, then the bounds state goes back to
        ``vol_bnds`` (the configured bounds; the reference starts a new process per sequence)."""
        self.flush()
        self._mm_seq = 0   # (the new sequence's scans are numbered from 0 again)
        if self._mm_state is not None:
            self._mm_state.reset(vol_bnds)

    def wait(self, ticket):
        """Images of the scan with this ticket (complete when the call returns).  A ticket is handed out once."""
        ticket = int(ticket)
        with self._lock:
            ev = self._done.get(ticket)
        if ev is None:
            raise KeyError(f"FusionScanPipeline.wait: ticket {ticket} is unknown or was already collected")
        ev.wait()
        with self._lock:
            res = self._result.pop(ticket)
            del self._done[ticket]
        if isinstance(res, BaseException):
            raise res
        res.pop("_done")[0].synchronize()
        # the images were allocated on the chain's stream: tell the caching allocator that the caller's stream uses them,
        # so that freeing them early cannot hand their memory back to the chain while the caller still reads it
        cur = self._torch.cuda.current_stream(self.device)
        for v in res.values():
            if isinstance(v, self._torch.Tensor) and v.is_cuda:
                v.record_stream(cur)
        return res

    def flush(self):
        """Complete every submitted scan; their images stay collectable with :meth:`wait`."""
        with self._lock:
            evs = list(self._done.values())
        for ev in evs:
            ev.wait()
        for ch in self._chains:
            ch["stream"].synchronize()

    def close(self):
        chains, self._chains = getattr(self, "_chains", []), []
        for ch in chains:
            ch["q"].put(None)
        import threading
        for ch in chains:
            if ch["thread"] is not threading.current_thread():  # (__del__ may run on a worker that dropped the last reference)
                ch["thread"].join()
            if ch.get("deform") is not None:
                ch["deform"].close()
            ch["mesh"].close()
            ch["scene"].close()
            if ch["vol"] is not None:
                ch["vol"].close()
            if ch.get("projector") is not None:
                ch["projector"].close()
        if getattr(self, "_mm_state", None) is not None and chains:
            self._mm_state.close()
            self._mm_state = None
        if getattr(self, "rayset", None) is not None and chains:
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
