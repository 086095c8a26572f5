"""What the reference does with the rendered images right after the hot path:
back-projection, SemanticKITTI packing / writing, and the source-vs-target comparison -- computed by
``liblidarhip.so`` (lt_post.hip); torch is used for device buffers only."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib


def _dev(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _device():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("lidar_transfer_amd.post needs a GPU (there is no CPU implementation)")
    return torch.device("cuda", torch.cuda.current_device())


def do_reverse_projection_new(range_image, proj_x, proj_y, fov_up, fov_down, preserve_float=False):
    """``LaserScan.do_reverse_projection_new`` (auxiliary/laserscan.py:475-501).

    ``range_image [H,W] f32``; ``proj_x / proj_y [H,W]``: the integer pixel coordinates of the winning point of
    each cell (``scan.proj_x``), or the float ones (``scan.proj_x_float``) with ``preserve_float``.  Returns
    ``back_points [H*W, 3] float64``.
    """
    import torch
    lib = _lib.load()
    dev = _device()
    H, W = range_image.shape
    rng = _dev(np.asarray(range_image, np.float32), dev)
    dt = np.float64 if preserve_float else np.int32
    px, py = _dev(np.asarray(proj_x, dt), dev), _dev(np.asarray(proj_y, dt), dev)
    out = torch.empty((H * W, 3), dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream(dev)
    _lib.check(lib.lt_reverse_projection_dev(rng.data_ptr(), px.data_ptr(), py.data_ptr(), int(preserve_float),
                                             float(fov_up), float(fov_down), H, W, out.data_ptr(),
                                             C.c_void_p(st.cuda_stream)), "lt_reverse_projection_dev")
    return out.cpu().numpy()


def pack_scan(back_points, label_image, remissions, index=None):
    """Filtering + packing of ``MultiSemLaserScan.write`` (auxiliary/laserscan.py:1133-1160): returns
    ``(bin [N,4] float32, label [N] uint32)`` -- byte-for-byte what the reference writes to
    ``velodyne/NNNNNN.bin`` and ``labels/NNNNNN.label``.  ``index`` is the ``merged.index`` image of the
    ``cp`` adaption (cells with ``index <= 0`` are dropped there), ``None`` for the mesh adaptions."""
    import torch
    lib = _lib.load()
    dev = _device()
    pts = np.ascontiguousarray(np.asarray(back_points).reshape(-1, 3))
    if pts.dtype not in (np.float32, np.float64):
        pts = pts.astype(np.float64)
    n = pts.shape[0]
    lab = np.asarray(label_image).reshape(-1)
    lab = np.where(np.isfinite(lab.astype(np.float64)), lab, -1).astype(np.int32) if lab.dtype.kind == "f" \
        else lab.astype(np.int32)
    rem = np.asarray(remissions, np.float32).reshape(-1)
    tp, tr, tl = _dev(pts, dev), _dev(rem, dev), _dev(lab, dev)
    ti = _dev(np.asarray(index, np.int32).reshape(-1), dev) if index is not None else None
    out_bin = torch.empty((n, 4), dtype=torch.float32, device=dev)
    out_lab = torch.empty((n,), dtype=torch.int32, device=dev)
    kept = C.c_int(0)
    st = torch.cuda.current_stream(dev)
    _lib.check(lib.lt_pack_scan_dev(tp.data_ptr(), int(pts.dtype == np.float64), tr.data_ptr(), tl.data_ptr(),
                                    ti.data_ptr() if ti is not None else None, n, out_bin.data_ptr(),
                                    out_lab.data_ptr(), C.byref(kept), C.c_void_p(st.cuda_stream)),
               "lt_pack_scan_dev")
    k = kept.value
    return out_bin[:k].cpu().numpy(), out_lab[:k].cpu().numpy().view(np.uint32)


def write_scan(out_dir, idx, back_points, label_image, remissions, index=None):
    """Write ``velodyne/NNNNNN.bin`` and ``labels/NNNNNN.label`` like laserscan.py:1162-1178."""
    b, l = pack_scan(back_points, label_image, remissions, index)
    os.makedirs(os.path.join(out_dir, "velodyne"), exist_ok=True)
    os.makedirs(os.path.join(out_dir, "labels"), exist_ok=True)
    b.tofile(os.path.join(out_dir, "velodyne", str(idx).zfill(6) + ".bin"))
    l.tofile(os.path.join(out_dir, "labels", str(idx).zfill(6) + ".label"))
    return b.shape[0]


def compare(source_label, source_color, target_label, source_range, target_range, source_rem, target_rem, nclasses):
    """Array part of ``compare()`` (auxiliary/laserscan.py:1181-1301) + ``iouEval`` (np_ioueval.py).

    Images ``[H,W]`` (colour ``[H,W,3]``).  Returns ``dict(range_diff, rem_diff, m_iou, m_acc, MSE, iou,
    source_label, target_label)`` with the reference's masking, class compaction (labels are renumbered by
    rank among the values present, laserscan.py:1216-1222) and ignore-empty-classes rule.
    """
    import torch
    lib = _lib.load()
    dev = _device()
    H, W = np.asarray(source_label).shape
    n = H * W
    src_l = np.asarray(source_label, np.int32).reshape(-1)
    tgt_l = np.asarray(target_label, np.int32).reshape(-1)
    mx = int(max(np.max(src_l), np.max(tgt_l), 0)) + 1
    lo = int(min(np.min(src_l), np.min(tgt_l), 0))
    if lo < 0:
        # The reference renumbers the labels among the values present (np.unique, laserscan.py:1216-1222) BEFORE
        # they index the confusion matrix (in place -- see the replay below: negative labels can merge classes).  The
        # histogram kernel counts non-negative codes: negatives travel as codes above the largest label (mx - 1 - v)
        # -- raw 0 stays 0, which is what the background rule tests -- and are put back in value order below.
        src_l = np.where(src_l < 0, mx - 1 - src_l, src_l).astype(np.int32)
        tgt_l = np.where(tgt_l < 0, mx - 1 - tgt_l, tgt_l).astype(np.int32)
    NL = 1
    while NL < mx - lo:
        NL <<= 1
    sl, tl = _dev(src_l, dev), _dev(tgt_l, dev)
    sc = _dev(np.asarray(source_color, np.float32).reshape(-1, 3), dev)
    sr, tr = _dev(np.asarray(source_range, np.float32).reshape(-1), dev), _dev(np.asarray(target_range, np.float32).reshape(-1), dev)
    sm, tm = _dev(np.asarray(source_rem, np.float32).reshape(-1), dev), _dev(np.asarray(target_rem, np.float32).reshape(-1), dev)
    conf = torch.empty((NL, NL), dtype=torch.int64, device=dev)
    rd = torch.empty(n, dtype=torch.float32, device=dev)
    md = torch.empty(n, dtype=torch.float32, device=dev)
    slm = torch.empty(n, dtype=torch.int32, device=dev)
    tlm = torch.empty(n, dtype=torch.int32, device=dev)
    sq = torch.zeros(1, dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream(dev)
    _lib.check(lib.lt_compare_dev(sl.data_ptr(), sc.data_ptr(), tl.data_ptr(), sr.data_ptr(), tr.data_ptr(),
                                  sm.data_ptr(), tm.data_ptr(), n, NL, conf.data_ptr(), rd.data_ptr(), md.data_ptr(),
                                  slm.data_ptr(), tlm.data_ptr(), sq.data_ptr(), C.c_void_p(st.cuda_stream)),
               "lt_compare_dev")
    conf = conf.cpu().numpy()          # conf[target, source] over raw labels
    # class compaction + iouEval on the (tiny) confusion matrix: host bookkeeping, no image work
    present = np.nonzero(conf.sum(0) + conf.sum(1))[0]
    # codes of negative labels back to values: code c >= mx stands for mx - 1 - c
    value = np.where(present >= mx, mx - 1 - present, present) if lo < 0 else present.copy()
    order = np.argsort(value, kind="stable")
    present, value = present[order], value[order]
    # The reference renumbers IN PLACE on the arrays it scans (laserscan.py:1216-1222: label[label == value] = i for
    # the sorted values present).  For non-negative labels that is the rank (u_i >= i: no rank meets a value still to
    # come); with negative labels a rank can equal a later value and the two classes MERGE ({-1, 0, 3}: -1 -> 0, then
    # every 0 -> 1).  Replayed here on the list of values, not on the images.
    cur = value.copy()
    for i, v in enumerate(value):
        cur[cur == v] = i
    final = cur  # final[j] = class of the pixels whose raw label is value[j]
    if len(final) and int(final.max()) >= nclasses:
        # np.add.at would raise IndexError in the reference (np_ioueval.py:47)
        raise IndexError(f"compare: class index {int(final.max())} after renumbering but nclasses = {nclasses}")
    cm = np.zeros((nclasses, nclasses), np.int64)
    np.add.at(cm, (final[:, None], final[None, :]), conf[np.ix_(present, present)])
    used = np.unique(final)
    ignore = np.setdiff1d(np.arange(nclasses), used)
    include = np.array([c for c in range(nclasses) if c not in set(ignore.tolist())], dtype=np.int64)
    c2 = cm.copy()
    c2[ignore] = 0
    c2[:, ignore] = 0
    tp = np.diag(c2)
    fp = c2.sum(axis=1) - tp
    fn = c2.sum(axis=0) - tp
    union = tp + fp + fn + 1e-15
    iou = tp / union
    m_iou = (tp[include] / union[include]).mean()
    m_acc = tp.sum() / (tp[include].sum() + fp[include].sum() + 1e-15)
    remap = np.full(NL, -1, np.int32)
    remap[present] = final.astype(np.int32)
    return dict(range_diff=rd.cpu().numpy().reshape(H, W), rem_diff=md.cpu().numpy().reshape(H, W), m_iou=m_iou,
                m_acc=m_acc, MSE=float(sq.item()) / n, iou=iou,
                source_label=remap[slm.cpu().numpy()].reshape(H, W), target_label=remap[tlm.cpu().numpy()].reshape(H, W))
