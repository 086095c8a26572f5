"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the array part of ``compare()``
(/root/reference auxiliary/laserscan.py:1181-1301) and of ``iouEval`` (auxiliary/np_ioueval.py:25-70).

Pinned against golden vectors produced by the reference's own Python (tests/golden F7,
tests/test_oracle_cpu.py); used by the GPU tests as the checker for label sets the golden scene does not
contain (negative labels).  Nothing in the product imports this module.
"""
import numpy as np


def compare(source_label, source_color, target_label, source_range, target_range, source_rem, target_rem, nclasses):
    source_color = np.copy(source_color)
    source_label = np.copy(source_label)
    target_label = np.copy(target_label)
    # laserscan.py:1201-1211: no data (= black) in the source is background in both; so is source label 0
    black_values = np.sum(source_color, axis=2) == 0
    source_label[black_values] = 0
    target_label[black_values] = 0
    bg_label = source_label == 0
    target_label[bg_label] = 0
    # laserscan.py:1216-1222: renumber by rank among the values present -- IN PLACE, on the arrays being scanned, as
    # the reference does: with negative labels a rank can exceed a value that is still to come ({-1, 0, 3}: -1 -> 0,
    # then every 0, the former -1 included, -> 1), so classes merge.  Non-negative label sets never alias (u_i >= i).
    unique_values = np.union1d(np.unique(source_label), np.unique(target_label))
    sl, tl = source_label, target_label   # (copies made above)
    for i, value in enumerate(unique_values):
        mask_source = sl == value
        mask_target = tl == value
        sl[mask_source] = i
        tl[mask_target] = i
    unique_values = np.union1d(np.unique(sl), np.unique(tl))
    empty = np.isin(np.arange(nclasses), unique_values, invert=True)
    ignore = np.arange(nclasses)[empty]
    include = np.array([n for n in range(nclasses) if n not in ignore], dtype=np.int64)
    # np_ioueval.py:31-47 (rows = predictions = target image, cols = ground truth = source image)
    conf = np.zeros((nclasses, nclasses), dtype=np.int64)
    np.add.at(conf, (tl.reshape(-1), sl.reshape(-1)), 1)
    conf[ignore] = 0
    conf[:, ignore] = 0
    tp = np.diag(conf)
    fp = conf.sum(axis=1) - tp
    fn = conf.sum(axis=0) - tp
    union = tp + fp + fn + 1e-15
    iou = tp / union
    m_iou = (tp[include] / union[include]).mean()
    m_acc = tp.sum() / (tp[include].sum() + fp[include].sum() + 1e-15)
    # laserscan.py:1239-1281
    sr, tr = np.copy(source_range), np.copy(target_range)
    sr[bg_label] = 0
    tr[bg_label] = 0
    range_diff = (sr - tr) ** 2
    mse = range_diff.sum() / range_diff.size
    sm, tm = np.copy(source_rem), np.copy(target_rem)
    sm[bg_label] = 0
    tm[bg_label] = 0
    rem_diff = (sm - tm) ** 2
    return dict(range_diff=range_diff, rem_diff=rem_diff, m_iou=m_iou, m_acc=m_acc, MSE=mse, iou=iou,
                source_label=sl, target_label=tl)
