"""Restatement of the third-party arithmetic the reference hot path delegates to.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference imports these from packages that are NOT vendored in /root/reference and
are not installable here (no network):

* ``ultralytics`` (PyPI, floor pin ``>=8.4.110`` -- reference requirements.txt:18,
  pyproject.toml:78): ``utils.metrics.{bbox_iou,box_iou,smooth_bce}``,
  ``utils.ops.{xywh2xyxy,xyxy2xywh,clip_boxes,make_divisible}``,
  ``utils.torch_utils.{fuse_conv_and_bn,initialize_weights,scale_img}``.
* ``torchvision`` (floor pin ``>=0.9.0`` -- reference requirements.txt:17): ``ops.nms``.

Each function below restates the published algorithm and names the reference call site
that anchors it.  Exact upstream versions cannot be checked offline, so these are
"parity unpinned" w.r.t. upstream; parity of everything *inside* /root/reference is
pinned by running the unmodified reference files on top of these functions
(oracle/ref_shim.py + tests/golden/make_golden.py).
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess
from pathlib import Path

import numpy as np
import torch
from torch import nn

_HERE = Path(__file__).resolve().parent


# --------------------------------------------------------------------------- boxes
def xywh2xyxy(x):
    """(cx,cy,w,h) -> (x1,y1,x2,y2), same dtype.  Anchors: reference utils/general.py:705
    (NMS box conversion), val.py:401."""
    out = x.clone() if isinstance(x, torch.Tensor) else np.copy(x)
    half = x[..., 2:4] / 2
    out[..., 0:2] = x[..., 0:2] - half
    out[..., 2:4] = x[..., 0:2] + half
    return out


def xyxy2xywh(x):
    """(x1,y1,x2,y2) -> (cx,cy,w,h).  Anchor: reference utils/general.py:50 (import)."""
    out = x.clone() if isinstance(x, torch.Tensor) else np.copy(x)
    out[..., 0] = (x[..., 0] + x[..., 2]) / 2
    out[..., 1] = (x[..., 1] + x[..., 3]) / 2
    out[..., 2] = x[..., 2] - x[..., 0]
    out[..., 3] = x[..., 3] - x[..., 1]
    return out


def clip_boxes(boxes, shape):
    """Clamp xyxy boxes to an (h, w) image, in place.  Anchor: reference utils/general.py:625."""
    if isinstance(boxes, torch.Tensor):
        boxes[..., 0].clamp_(0, shape[1])
        boxes[..., 1].clamp_(0, shape[0])
        boxes[..., 2].clamp_(0, shape[1])
        boxes[..., 3].clamp_(0, shape[0])
    else:
        boxes[..., [0, 2]] = boxes[..., [0, 2]].clip(0, shape[1])
        boxes[..., [1, 3]] = boxes[..., [1, 3]].clip(0, shape[0])
    return boxes


def make_divisible(x, divisor):
    """Smallest multiple of ``divisor`` >= x.  Anchor: reference models/yolo.py:348."""
    if isinstance(divisor, torch.Tensor):
        divisor = int(divisor.max())
    return math.ceil(x / divisor) * divisor


def box_iou(box1, box2, eps=1e-7):
    """Pairwise IoU of xyxy boxes, (N,4)x(M,4)->(N,M).  Anchors: reference utils/metrics.py:153,
    val.py:176, utils/general.py:737."""
    a1, a2 = box1.float().unsqueeze(1).chunk(2, 2)
    b1, b2 = box2.float().unsqueeze(0).chunk(2, 2)
    inter = (torch.min(a2, b2) - torch.max(a1, b1)).clamp_(0).prod(2)
    return inter / ((a2 - a1).prod(2) + (b2 - b1).prod(2) - inter + eps)


def bbox_iou(box1, box2, xywh=True, GIoU=False, DIoU=False, CIoU=False, eps=1e-7):
    """IoU family between paired boxes, returns (n,1).  Anchor: reference utils/loss.py:151
    (``bbox_iou(pbox, tbox[i], CIoU=True).squeeze()``)."""
    if xywh:
        (x1, y1, w1, h1), (x2, y2, w2, h2) = box1.chunk(4, -1), box2.chunk(4, -1)
        hw1, hh1, hw2, hh2 = w1 / 2, h1 / 2, w2 / 2, h2 / 2
        b1_x1, b1_x2, b1_y1, b1_y2 = x1 - hw1, x1 + hw1, y1 - hh1, y1 + hh1
        b2_x1, b2_x2, b2_y1, b2_y2 = x2 - hw2, x2 + hw2, y2 - hh2, y2 + hh2
    else:
        b1_x1, b1_y1, b1_x2, b1_y2 = box1.chunk(4, -1)
        b2_x1, b2_y1, b2_x2, b2_y2 = box2.chunk(4, -1)
        w1, h1 = b1_x2 - b1_x1, b1_y2 - b1_y1 + eps
        w2, h2 = b2_x2 - b2_x1, b2_y2 - b2_y1 + eps

    inter = (b1_x2.minimum(b2_x2) - b1_x1.maximum(b2_x1)).clamp_(0) * (
        b1_y2.minimum(b2_y2) - b1_y1.maximum(b2_y1)
    ).clamp_(0)
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / union
    if CIoU or DIoU or GIoU:
        cw = b1_x2.maximum(b2_x2) - b1_x1.minimum(b2_x1)
        ch = b1_y2.maximum(b2_y2) - b1_y1.minimum(b2_y1)
        if CIoU or DIoU:
            c2 = cw.pow(2) + ch.pow(2) + eps
            rho2 = ((b2_x1 + b2_x2 - b1_x1 - b1_x2).pow(2) + (b2_y1 + b2_y2 - b1_y1 - b1_y2).pow(2)) / 4
            if CIoU:
                v = (4 / math.pi**2) * ((w2 / h2).atan() - (w1 / h1).atan()).pow(2)
                with torch.no_grad():
                    alpha = v / (v - iou + (1 + eps))
                return iou - (rho2 / c2 + v * alpha)
            return iou - rho2 / c2
        c_area = cw * ch + eps
        return iou - (c_area - union) / c_area
    return iou


def smooth_bce(eps=0.1):
    """Label-smoothing BCE targets (positive, negative).  Anchor: reference utils/loss.py:114."""
    return 1.0 - 0.5 * eps, 0.5 * eps


# --------------------------------------------------------------------------- model helpers
def smooth(y, f=0.05):
    """ultralytics.utils.metrics.smooth (box filter of fraction f), used by the reference's ap_per_class (utils/metrics.py:82) to pick
    the max-F1 operating point.  Restated from the published function: parity unpinned like the rest of this file."""
    nf = round(len(y) * f * 2) // 2 + 1
    p = np.ones(nf // 2)
    yp = np.concatenate((p * y[0], y, p * y[-1]), 0)
    return np.convolve(yp, np.ones(nf) / nf, mode="valid")


def fuse_conv_and_bn(conv, bn):
    """Fold an eval-mode BatchNorm2d into the preceding Conv2d.  Anchor: reference
    models/yolo.py:168 (``m.conv = fuse_conv_and_bn(m.conv, m.bn)``)."""
    fused = (
        nn.Conv2d(
            conv.in_channels,
            conv.out_channels,
            kernel_size=conv.kernel_size,
            stride=conv.stride,
            padding=conv.padding,
            dilation=conv.dilation,
            groups=conv.groups,
            bias=True,
        )
        .requires_grad_(False)
        .to(conv.weight.device)
    )
    w_conv = conv.weight.view(conv.out_channels, -1)
    w_bn = torch.diag(bn.weight.div(torch.sqrt(bn.eps + bn.running_var)))
    fused.weight.copy_(torch.mm(w_bn, w_conv).view(fused.weight.shape))
    b_conv = torch.zeros(conv.weight.shape[0], device=conv.weight.device) if conv.bias is None else conv.bias
    b_bn = bn.bias - bn.weight.mul(bn.running_mean).div(torch.sqrt(bn.running_var + bn.eps))
    fused.bias.copy_(torch.mm(w_bn, b_conv.reshape(-1, 1)).reshape(-1) + b_bn)
    return fused


def initialize_weights(model):
    """BN eps/momentum and in-place activations.  Anchor: reference models/yolo.py:229."""
    for m in model.modules():
        t = type(m)
        if t is nn.BatchNorm2d:
            m.eps = 1e-3
            m.momentum = 0.03
        elif t in (nn.Hardswish, nn.LeakyReLU, nn.ReLU, nn.ReLU6, nn.SiLU):
            m.inplace = True


def scale_img(img, ratio=1.0, same_shape=False, gs=32):
    """Published upstream ``ultralytics.utils.torch_utils.scale_img`` (call site: reference models/yolo.py:246, test-time augmentation): bilinear resize of
    an NCHW batch to (int(h ratio), int(w ratio)), then right / bottom padding with 0.447 (the ImageNet mean) up to the next multiple of ``gs``."""
    if ratio == 1.0:
        return img
    h, w = img.shape[2:]
    s = (int(h * ratio), int(w * ratio))
    img = torch.nn.functional.interpolate(img, size=s, mode="bilinear", align_corners=False)
    if not same_shape:
        h, w = (math.ceil(x * ratio / gs) * gs for x in (h, w))
    return torch.nn.functional.pad(img, [0, w - s[1], 0, h - s[0]], value=0.447)


def one_cycle(y1=0.0, y2=1.0, steps=100):
    """Cosine ramp y1->y2.  Anchor: reference train.py:242."""
    return lambda x: max((1 - math.cos(x * math.pi / steps)) / 2, 0) * (y2 - y1) + y1


def intersect_dicts(da, db, exclude=()):
    """Keys present in both with equal shapes.  Anchor: reference train.py:209."""
    return {k: v for k, v in da.items() if k in db and all(x not in k for x in exclude) and v.shape == db[k].shape}


# --------------------------------------------------------------------------- NMS
def nms_py(boxes: np.ndarray, scores: np.ndarray, thr: float) -> np.ndarray:
    """Greedy NMS, pure numpy (small n only).  torchvision.ops.nms CPU kernel restated:
    stable descending score sort; keep i; suppress later j when
    inter/(area_i+area_j-inter) > thr (strict).  All arithmetic float32.
    Anchor: reference utils/general.py:733."""
    boxes = np.asarray(boxes, dtype=np.float32)
    scores = np.asarray(scores, dtype=np.float32)
    n = boxes.shape[0]
    order = np.argsort(-scores, kind="stable")
    x1, y1, x2, y2 = (boxes[:, k] for k in range(4))
    areas = (x2 - x1) * (y2 - y1)
    dead = np.zeros(n, dtype=bool)
    keep = []
    thr = float(thr)  # torchvision compares the float32 ratio with the double threshold
    zero = np.float32(0)
    for a in range(n):
        i = order[a]
        if dead[i]:
            continue
        keep.append(i)
        rest = order[a + 1 :]
        w = np.maximum(zero, np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]))
        h = np.maximum(zero, np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]))
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (areas[i] + areas[rest] - inter)
        dead[rest[ovr.astype(np.float64) > thr]] = True
    return np.asarray(keep, dtype=np.int64)


_NMS_LIB = None


def _load_nms_c():
    """Build (if needed) and load oracle/_build/libnms_oracle.so (plain C, gcc)."""
    global _NMS_LIB
    if _NMS_LIB is not None:
        return _NMS_LIB
    so = _HERE / "_build" / "libnms_oracle.so"
    src = _HERE / "nms_oracle.c"
    if not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        so.parent.mkdir(exist_ok=True)
        subprocess.check_call(
            ["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC", str(src), "-o", str(so)]
        )
    lib = ctypes.CDLL(str(so))
    lib.y3o_nms.restype = ctypes.c_long
    lib.y3o_nms.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_double, ctypes.c_void_p]
    _NMS_LIB = lib
    return lib


def nms(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    """Drop-in for ``torchvision.ops.nms`` on CPU tensors (C restatement, float32).
    Returns kept indices in descending-score order, int64."""
    b = boxes.detach().to("cpu", torch.float32).contiguous()
    s = scores.detach().to("cpu", torch.float32).contiguous()
    n = b.shape[0]
    out = torch.empty(n, dtype=torch.int64)
    if n == 0:
        return out
    lib = _load_nms_c()
    k = lib.y3o_nms(b.data_ptr(), s.data_ptr(), n, float(iou_threshold), out.data_ptr())
    return out[:k].to(boxes.device)
