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


def scale_boxes(img1_shape, boxes, img0_shape, ratio_pad=None):
    """Undo letterbox scaling/padding (reference utils/general.py:613-626)."""
    if ratio_pad is None:
        gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
        pad = (img1_shape[1] - img0_shape[1] * gain) / 2, (img1_shape[0] - img0_shape[0] * gain) / 2
    else:
        gain = ratio_pad[0][0]
        pad = ratio_pad[1]
    boxes[..., [0, 2]] -= pad[0]
    boxes[..., [1, 3]] -= pad[1]
    boxes[..., :4] /= gain
    clip_boxes(boxes, img0_shape)
    return boxes


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
