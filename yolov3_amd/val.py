"""Host-side mirror of the metric-matching step of reference val.py: ``process_batch`` (:147-188) with the reference
signature, and a batched form that consumes the batched NMS output without a per-image Python loop or device->host
copies.  The matching runs in csrc/val_edge.hip; AP accumulation (``ap_per_class``) stays NumPy in the
reference: yolov3_amd/metrics.py is its host mirror."""
from __future__ import annotations

import torch

from . import ops


def process_batch(detections, labels, iouv):
    """Drop-in for reference val.py:147: detections (N, 6) [x1,y1,x2,y2,conf,cls], labels (M, 5) [cls,x1,y1,x2,y2], iouv (T,)
    -> bool (N, T) on ``iouv.device``.  Exact IoU ties between two labels of one detection resolve to the larger label index
    (what numpy's reversed argsort gives the reference for <= 16 candidate pairs; undefined there beyond that)."""
    ops.require_gpu(detections, "process_batch")
    dev = detections.device
    n = detections.shape[0]
    if n == 0:
        return torch.zeros(0, iouv.numel(), dtype=torch.bool, device=iouv.device)
    if n > 4096:
        raise ValueError("process_batch: more than 4096 detections per image is not supported")
    dets = detections if (detections.dtype == torch.float32 and detections.stride(1) == 1) else detections.float().contiguous()
    lab = labels.to(dev, torch.float32).contiguous()
    offs = torch.tensor([0, lab.shape[0]], dtype=torch.int32).to(dev, non_blocking=True)
    thr = iouv.to(dev, torch.float32).contiguous()
    correct = ops.match_detections_raw(dets, n * dets.stride(0), dets.stride(0), None, 1, n, lab, offs, thr)
    return correct[0].bool().to(iouv.device)


def process_batch_batched(rows, counts, labels, label_offsets, iouv):
    """Matching for a whole batch in one launch: rows (bs, max_det, 6) fp32 + counts (device int32, or None) as returned by
    `non_max_suppression_batched` (after `scale_boxes_batched`), labels (nl, 5) fp32 [cls,x1,y1,x2,y2] grouped by image with
    label_offsets (bs+1 int32: image i owns labels[label_offsets[i]:label_offsets[i+1]]).  Returns uint8 (bs, max_det, T) on
    the device, rows beyond counts[i] are 0."""
    ops.require_gpu(rows, "process_batch_batched")
    if rows.dtype != torch.float32 or rows.dim() != 3 or not rows.is_contiguous():
        raise TypeError("process_batch_batched expects the contiguous (bs, max_det, 6) fp32 NMS output")
    dev = rows.device
    lab = labels.to(dev, torch.float32).contiguous()
    offs = label_offsets.to(dev, torch.int32).contiguous()
    if offs.numel() != rows.shape[0] + 1:
        raise ValueError("label_offsets must hold bs + 1 entries")
    thr = iouv.to(dev, torch.float32).contiguous()
    return ops.match_detections_raw(rows, rows.stride(0), rows.stride(1), counts, rows.shape[0], rows.shape[1], lab, offs, thr)


def detect_batches(model, batches, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False, max_det=300):
    """Throughput form of the `model(im)` -> `non_max_suppression(preds)` pair of reference val.py:364-376 / detect.py:196-200
    over a stream of batches: the NMS of batch i (a chain of small launches ending in the one device->host copy of the counts)
    runs on a second HIP stream while the forward of batch i+1 fills the CUs on the current stream.  Yields, in order and one
    batch late, the list of (n, 6) detections of every batch -- the same tensors the sequential pair returns."""
    from .general import non_max_suppression

    cur = torch.cuda.current_stream()
    side = torch.cuda.Stream(device=cur.device)
    pending = None

    def finish(p):
        pred, ev = p
        with torch.cuda.stream(side):
            side.wait_event(ev)
            dets = non_max_suppression(pred, conf_thres, iou_thres, classes, agnostic, multi_label, max_det=max_det)
        for d in dets:
            d.record_stream(cur)  # allocated on the side stream, consumed by the caller on the current one
        cur.wait_stream(side)
        return dets

    for x in batches:
        out = model(x)
        pred = out[0] if isinstance(out, (list, tuple)) else out
        ev = torch.cuda.Event()
        ev.record(cur)
        if pending is not None:
            yield finish(pending)
        pending = (pred, ev)
    if pending is not None:
        yield finish(pending)
