"""Validation metrics on the host, as in the reference (utils/metrics.py keeps them in NumPy; val.py:417-421 calls them once per
validation run on the concatenated per-image statistics).  Not on the device hot path: the per-image half of the evaluation --
non_max_suppression, scale_boxes, process_batch -- is (yolov3_amd/general.py, yolov3_amd/val.py); this file turns their output into
P / R / AP exactly like the reference so that a `val.py`-style loop has everything it needs.

`smooth` is upstream ultralytics.utils.metrics.smooth (un-vendored; restated from the published function, it only picks the F1
operating point -- AP does not depend on it).  No plotting."""
from __future__ import annotations

import numpy as np


def fitness(x):
    """reference utils/metrics.py:15-18: weighted sum of [P, R, mAP@0.5, mAP@0.5:0.95] with weights [0, 0, 0.1, 0.9]"""
    w = [0.0, 0.0, 0.1, 0.9]
    return (x[:, :4] * w).sum(1)


def smooth(y, f=0.05):
    """box filter of fraction f (upstream ultralytics.utils.metrics.smooth)"""
    nf = round(len(y) * f * 2) // 2 + 1   # filter length, odd
    p = np.ones(nf // 2)
    yp = np.concatenate((p * y[0], y, p * y[-1]), 0)
    return np.convolve(yp, np.ones(nf) / nf, mode="valid")


def compute_ap(recall, precision):
    """reference utils/metrics.py:89-118: 101-point interpolated AP (COCO) of one precision / recall curve.
    Returns (ap, precision envelope, recall with sentinels)."""
    mrec = np.concatenate(([0.0], recall, [1.0]))
    mpre = np.concatenate(([1.0], precision, [0.0]))
    mpre = np.flip(np.maximum.accumulate(np.flip(mpre)))   # precision envelope
    x = np.linspace(0, 1, 101)
    trapz = np.trapezoid if hasattr(np, "trapezoid") else np.trapz
    return trapz(np.interp(x, mrec, mpre), x), mpre, mrec


def ap_per_class(tp, conf, pred_cls, target_cls, plot=False, save_dir=".", names=(), eps=1e-16, prefix=""):
    """reference utils/metrics.py:22-86 (same positional contract and return arity; `plot` / `save_dir` / `prefix` are accepted and
    ignored -- plotting is out of scope).  tp: (n, n_iou) bool / 0-1 from process_batch, conf / pred_cls: (n,), target_cls: (n_labels,).
    Returns (tp, fp, p, r, f1, ap, unique_classes) with ap of shape (n_classes_with_labels, n_iou)."""
    i = np.argsort(-conf)
    tp, conf, pred_cls = tp[i], conf[i], pred_cls[i]
    unique_classes, nt = np.unique(target_cls, return_counts=True)
    nc = unique_classes.shape[0]
    px = np.linspace(0, 1, 1000)
    ap, p, r = np.zeros((nc, tp.shape[1])), np.zeros((nc, 1000)), np.zeros((nc, 1000))
    for ci, c in enumerate(unique_classes):
        i = pred_cls == c
        n_l, n_p = nt[ci], i.sum()
        if n_p == 0 or n_l == 0:
            continue
        fpc = (1 - tp[i]).cumsum(0)
        tpc = tp[i].cumsum(0)
        recall = tpc / (n_l + eps)
        r[ci] = np.interp(-px, -conf[i], recall[:, 0], left=0)   # negative x: xp must increase
        precision = tpc / (tpc + fpc)
        p[ci] = np.interp(-px, -conf[i], precision[:, 0], left=1)
        for j in range(tp.shape[1]):
            ap[ci, j], _, _ = compute_ap(recall[:, j], precision[:, j])
    f1 = 2 * p * r / (p + r + eps)
    i = smooth(f1.mean(0), 0.1).argmax()   # max-F1 operating point
    p, r, f1 = p[:, i], r[:, i], f1[:, i]
    tp = (r * nt).round()
    fp = (tp / (p + eps) - tp).round()
    return tp, fp, p, r, f1, ap, unique_classes.astype(int)


def mean_results(stats):
    """val.py:416-421: the concatenated per-image statistics [(correct, conf, pred_cls, target_cls), ...] (NumPy arrays) ->
    (mp, mr, map50, map)."""
    stats = [np.concatenate(x, 0) for x in zip(*stats)]
    if len(stats) and stats[0].any():
        _, _, p, r, _, ap, _ = ap_per_class(*stats)
        ap50, ap = ap[:, 0], ap.mean(1)
        return float(p.mean()), float(r.mean()), float(ap50.mean()), float(ap.mean())
    return 0.0, 0.0, 0.0, 0.0
