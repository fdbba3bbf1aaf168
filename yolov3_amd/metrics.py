"""Precision / recall / AP accumulation on the host (NumPy), the last step of a validation run.

The device path ends at `process_batch` (csrc/val_edge.hip): one row of IoU-threshold hits per detection.  This module turns the
concatenated rows of a run into the numbers reference val.py:417-421 prints, with the call contract of reference
utils/metrics.py:22 (`ap_per_class`, positional 7-tuple) and :89 (`compute_ap`) so a `val.py`-style loop can call it unchanged.
It is written from the definitions, not from the reference's code:

  * detections are ranked once by confidence; for a class c the k-th ranked detection of that class has
        recall_k = TP_k / n_labels(c),   precision_k = TP_k / k            (TP_k = hits among the first k)
    per IoU threshold -- all thresholds at once as a (k, T) array;
  * AP is COCO's 101-point interpolation: area under the monotone precision envelope sampled at recall 0, 0.01, ..., 1;
  * the reported P / R / F1 are read at the confidence where the class-mean F1 curve, box-filtered over 10 % of its 1000 samples
    (upstream ultralytics.utils.metrics.smooth, un-vendored), peaks.

Pinned to the unmodified reference by tests/golden/metrics.pt (the golden-vector tests, 1e-12).  No plotting."""
from __future__ import annotations

import numpy as np

_CONF_GRID = np.linspace(0.0, 1.0, 1000)     # confidence axis of the P(conf) / R(conf) curves
_RECALL_GRID = np.linspace(0.0, 1.0, 101)    # COCO recall samples


def fitness(x):
    """model-selection score of reference utils/metrics.py:15: 0.1 * mAP@0.5 + 0.9 * mAP@0.5:0.95 of rows [P, R, mAP50, mAP, ...]"""
    x = np.asarray(x)
    return 0.1 * x[:, 2] + 0.9 * x[:, 3]


def smooth(y, f=0.05):
    """moving average over a window of ~2 f len(y) samples (odd length), edges extended by their end values"""
    win = round(len(y) * f * 2) // 2 + 1
    half = win // 2
    padded = np.concatenate((np.full(half, y[0], dtype=float), y, np.full(half, y[-1], dtype=float)))
    return np.convolve(padded, np.full(win, 1.0 / win), mode="valid")


def compute_ap(recall, precision):
    """101-point interpolated average precision of one curve (recall ascending).  Returns (ap, envelope, recall) with the
    sentinels (recall 0 / precision 1 in front, recall 1 / precision 0 behind) included, like reference utils/metrics.py:89."""
    rec = np.concatenate(([0.0], recall, [1.0]))
    env = np.concatenate(([1.0], precision, [0.0]))
    env = np.maximum.accumulate(env[::-1])[::-1]            # best precision at this recall or beyond
    samples = np.interp(_RECALL_GRID, rec, env)
    area = float(((samples[1:] + samples[:-1]) * np.diff(_RECALL_GRID)).sum() * 0.5)   # trapezoid rule on the 101 samples
    return area, env, rec


def _class_curves(hits, conf, n_labels, eps):
    """hits (k, T) 0/1 in rank order, conf (k,) descending.  -> recall (k, T), precision (k, T), and both sampled on the
    confidence grid at the FIRST threshold (what the reference reports as P / R)."""
    tp_run = np.cumsum(hits, axis=0)
    fp_run = np.cumsum(1 - hits, axis=0)
    recall = tp_run / (n_labels + eps)
    precision = tp_run / (tp_run + fp_run)
    # np.interp wants ascending abscissae: walk the confidence axis downwards
    r_of_conf = np.interp(-_CONF_GRID, -conf, recall[:, 0], left=0)
    p_of_conf = np.interp(-_CONF_GRID, -conf, precision[:, 0], left=1)
    return recall, precision, r_of_conf, p_of_conf


def ap_per_class(tp, conf, pred_cls, target_cls, plot=False, save_dir=".", names=(), eps=1e-16, prefix=""):
    """Per-class statistics of a validation run; call contract of reference utils/metrics.py:22 (`plot`, `save_dir`, `names`,
    `prefix` accepted and ignored).  tp (n, T) from process_batch, conf / pred_cls (n,), target_cls (n_labels,).
    Returns (tp_count, fp_count, p, r, f1, ap (classes, T), classes) over the classes that have labels."""
    order = np.argsort(-conf)
    tp, conf, pred_cls = tp[order], conf[order], pred_cls[order]
    classes, label_counts = np.unique(target_cls, return_counts=True)
    n_cls, n_thr = classes.shape[0], tp.shape[1]
    ap = np.zeros((n_cls, n_thr))
    p_curve = np.zeros((n_cls, _CONF_GRID.size))
    r_curve = np.zeros((n_cls, _CONF_GRID.size))
    for row, (cls, n_lab) in enumerate(zip(classes, label_counts)):
        mine = pred_cls == cls
        if not mine.any() or n_lab == 0:
            continue
        recall, precision, r_curve[row], p_curve[row] = _class_curves(tp[mine], conf[mine], n_lab, eps)
        ap[row] = [compute_ap(recall[:, t], precision[:, t])[0] for t in range(n_thr)]
    f1_curve = 2 * p_curve * r_curve / (p_curve + r_curve + eps)
    best = smooth(f1_curve.mean(0), 0.1).argmax()
    p, r, f1 = p_curve[:, best], r_curve[:, best], f1_curve[:, best]
    tp_count = (r * label_counts).round()
    fp_count = (tp_count / (p + eps) - tp_count).round()
    return tp_count, fp_count, p, r, f1, ap, classes.astype(int)


def mean_results(stats):
    """[(correct, conf, pred_cls, target_cls), ...] per image (NumPy) -> (mean P, mean R, mAP@0.5, mAP@0.5:0.95), the summary line
    of reference val.py:416-421."""
    cols = [np.concatenate(c, 0) for c in zip(*stats)]
    if not cols or not cols[0].any():
        return 0.0, 0.0, 0.0, 0.0
    _, _, p, r, _, ap, _ = ap_per_class(*cols)
    return float(p.mean()), float(r.mean()), float(ap[:, 0].mean()), float(ap.mean(1).mean())
