"""Host-side mirror of the hot-path pieces of reference utils/general.py: ``non_max_suppression`` (:630-750)
with the reference signature, plus ``scale_boxes`` / ``clip_boxes`` / ``xywh2xyxy`` helpers used around it.
The NMS itself runs as one batched HIP pipeline (csrc/detect_nms.hip) -- no per-image Python loop, no
torchvision.  CPU tensors are rejected: there is no fallback."""
from __future__ import annotations

import torch

from . import ops


def xywh2xyxy(x):
    """(cx,cy,w,h)->(x1,y1,x2,y2) (upstream ultralytics.utils.ops.xywh2xyxy; reference utils/general.py:705)."""
    y = x.clone()
    half = x[..., 2:4] / 2
    y[..., 0:2] = x[..., 0:2] - half
    y[..., 2:4] = x[..., 0:2] + half
    return y


def clip_boxes(boxes, shape):
    """Clamp xyxy boxes to (h, w) in place (upstream ultralytics.utils.ops.clip_boxes; reference utils/general.py:625)."""
    boxes[..., 0].clamp_(0, shape[1])
    boxes[..., 1].clamp_(0, shape[0])
    boxes[..., 2].clamp_(0, shape[1])
    boxes[..., 3].clamp_(0, shape[0])
    return boxes


def box_iou(box1, box2, eps=1e-7):
    """Pairwise IoU of xyxy boxes (N,4) x (M,4) -> (N,M) (upstream ultralytics.utils.metrics.box_iou; reference val.py:176,
    utils/general.py:737).  Plain tensor arithmetic: the hot use of it -- process_batch -- runs in csrc/val_edge.hip."""
    a1, a2 = box1.float().unsqueeze(1).chunk(2, 2)
    b1, b2 = box2.float().unsqueeze(0).chunk(2, 2)
    inter = (torch.min(a2, b2) - torch.max(a1, b1)).clamp_(0).prod(2)
    return inter / ((a2 - a1).prod(2) + (b2 - b1).prod(2) - inter + eps)


def _gain_pad(img1_shape, img0_shape, ratio_pad):
    """(gain, pad_x, pad_y) exactly as reference utils/general.py:615-620 computes them (Python floats)."""
    if ratio_pad is None:
        gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
        pad = (img1_shape[1] - img0_shape[1] * gain) / 2, (img1_shape[0] - img0_shape[0] * gain) / 2
    else:
        gain = ratio_pad[0][0]
        pad = ratio_pad[1]
    return float(gain), float(pad[0]), float(pad[1])


def scale_boxes(img1_shape, boxes, img0_shape, ratio_pad=None):
    """Undo letterbox scaling/padding and clip, in place (reference utils/general.py:613-626).  fp32 boxes on the MI355X
    (an (n, 4+) tensor or a column view such as ``predn[:, :4]``) go through y3_scale_boxes; anything else (CPU label
    tensors, half boxes) keeps the reference's tensor arithmetic."""
    if isinstance(boxes, torch.Tensor) and boxes.is_cuda and boxes.dtype == torch.float32 and boxes.dim() == 2 and boxes.shape[1] >= 4 and boxes.stride(1) == 1:
        if boxes.shape[0]:
            gain, px, py = _gain_pad(img1_shape, img0_shape, ratio_pad)
            params = torch.tensor([gain, px, py, float(img0_shape[1]), float(img0_shape[0])], dtype=torch.float32, device=boxes.device)
            ops.scale_boxes_raw(boxes, boxes.shape[0] * boxes.stride(0), boxes.stride(0), None, 1, boxes.shape[0], params)
        return boxes
    gain, px, py = _gain_pad(img1_shape, img0_shape, ratio_pad)
    boxes[..., [0, 2]] -= px
    boxes[..., [1, 3]] -= py
    boxes[..., :4] /= gain
    clip_boxes(boxes, img0_shape)
    return boxes


def scale_boxes_batched(img1_shape, rows, counts, img0_shapes, ratio_pads=None):
    """scale_boxes for a whole batch in ONE launch: ``rows`` is the (bs, max_det, 6) fp32 tensor and ``counts`` the device
    int32 counts of `non_max_suppression_batched` (or None); ``img0_shapes[i]`` / ``ratio_pads[i]`` are the per-image arguments the
    reference passes at val.py:397 / detect.py:223.  In place; returns ``rows``."""
    ops.require_gpu(rows, "scale_boxes_batched")
    if rows.dtype != torch.float32 or rows.dim() != 3 or not rows.is_contiguous():
        raise TypeError("scale_boxes_batched expects the contiguous (bs, max_det, 6) fp32 NMS output")
    bs = rows.shape[0]
    tab = []
    for i in range(bs):
        gain, px, py = _gain_pad(img1_shape, img0_shapes[i], None if ratio_pads is None else ratio_pads[i])
        tab.append([gain, px, py, float(img0_shapes[i][1]), float(img0_shapes[i][0])])
    params = torch.tensor(tab, dtype=torch.float32).to(rows.device, non_blocking=True)
    ops.scale_boxes_raw(rows, rows.stride(0), rows.stride(1), counts, bs, rows.shape[1], params)
    return rows


def non_max_suppression_batched(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False, max_det=300):
    """`non_max_suppression` without the Python list: returns (rows (bs, max_det, 6) fp32, counts device int32 (bs), counts list)
    so that scale_boxes_batched / process_batch_batched can consume the result where it lies."""
    assert 0 <= conf_thres <= 1, f"Invalid Confidence threshold {conf_thres}, valid values are between 0.0 and 1.0"
    assert 0 <= iou_thres <= 1, f"Invalid IoU {iou_thres}, valid values are between 0.0 and 1.0"
    if isinstance(prediction, (list, tuple)):
        prediction = prediction[0]
    ops.require_gpu(prediction, "non_max_suppression")
    rows, counts = ops.nms_raw(prediction, conf_thres, iou_thres, classes, agnostic, multi_label, max_det)
    return rows, torch.tensor(counts, dtype=torch.int32).to(rows.device, non_blocking=True), counts


def non_max_suppression(
    prediction,
    conf_thres=0.25,
    iou_thres=0.45,
    classes=None,
    agnostic=False,
    multi_label=False,
    labels=(),
    max_det=300,
    nm=0,
):
    """Drop-in for reference utils/general.py:630 (same arguments, same return: a list with one (n,6) fp32 tensor
    [x1,y1,x2,y2,conf,cls] per image on ``prediction.device``).

    Differences that are deliberate and documented (SURVEY.md 8a'):
      * exact score ties are ordered as torch.sort(stable=True) would (the reference's argsort is unstable);
      * the reference's wall-clock guard (:675,746-748), which silently drops images when NMS is slow, does not exist.
    """
    assert 0 <= conf_thres <= 1, f"Invalid Confidence threshold {conf_thres}, valid values are between 0.0 and 1.0"
    assert 0 <= iou_thres <= 1, f"Invalid IoU {iou_thres}, valid values are between 0.0 and 1.0"
    if isinstance(prediction, (list, tuple)):
        prediction = prediction[0]  # (inference_out, loss_out) -> inference_out   (:660-661)
    if nm:
        raise NotImplementedError("mask coefficients (nm>0) do not exist in YOLOv3 and are not implemented")
    ops.require_gpu(prediction, "non_max_suppression")
    bs, n_rows, no = prediction.shape
    nc = no - 5
    if labels and any(len(lb) for lb in labels):
        # apriori label rows (:689-695) enter the candidate list after the image's own rows: append them as extra
        # prediction rows (obj=1, one-hot class); images with fewer labels get zero rows (obj=0 never passes).
        extra = max(len(lb) for lb in labels)
        v = torch.zeros(bs, extra, no, dtype=prediction.dtype, device=prediction.device)
        for xi, lb in enumerate(labels):
            if len(lb):
                lb = lb.to(prediction.device)
                k = len(lb)
                v[xi, :k, :4] = lb[:, 1:5].to(prediction.dtype)
                v[xi, :k, 4] = 1.0
                v[xi, torch.arange(k, device=prediction.device), lb[:, 0].long() + 5] = 1.0
        prediction = torch.cat((prediction, v), 1)
    rows, counts = ops.nms_raw(prediction, conf_thres, iou_thres, classes, agnostic, multi_label, max_det)
    return [rows[i, :c] for i, c in enumerate(counts)]
