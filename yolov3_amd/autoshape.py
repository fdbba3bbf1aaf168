"""``AutoShape``: the reference's input-robust inference wrapper (models/common.py:766-876) on the MI355X path.

numpy / PIL images of any size go in; each is uploaded once as raw HWC uint8 and letterboxed ON THE DEVICE
(csrc/val_edge.hip: cv2.resize INTER_LINEAR + 114-border + HWC->CHW in one pass, reference utils/augmentations.py:104-134),
the uint8 batch feeds the model's first kernel directly (the /255 of models/common.py:868 happens inside the stem
convolution), then batched NMS and batched scale_boxes (one launch each instead of the per-image loop of :871-872).
Rendering helpers of the reference's ``Detections`` (show/save/crop/pandas) are outside the hot path and not provided.
"""
from __future__ import annotations

import time
from pathlib import Path

import numpy as np
import torch
from torch import nn

from . import ops
from .general import non_max_suppression_batched, scale_boxes_batched
from .yolo import make_divisible


def letterbox_geometry(shape, new_shape=(640, 640), auto=False, scaleup=True, stride=32):
    """(new_h, new_w, top, left, ratio, (dw, dh)) exactly as reference utils/augmentations.py:104-134 computes them
    (Python round() = round-half-even, like the reference)."""
    if isinstance(new_shape, int):
        new_shape = (new_shape, new_shape)
    r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
    if not scaleup:
        r = min(r, 1.0)
    new_unpad = round(shape[1] * r), round(shape[0] * r)
    dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
    if auto:
        dw, dh = dw % stride, dh % stride
    dw /= 2
    dh /= 2
    top, bottom = round(dh - 0.1), round(dh + 0.1)
    left, right = round(dw - 0.1), round(dw + 0.1)
    return new_unpad[1], new_unpad[0], top, left, (r, r), (dw, dh), (new_unpad[1] + top + bottom, new_unpad[0] + left + right)


def letterbox_batch(ims, shape1, device, color=114):
    """list of HWC uint8 numpy images -> (n, 3, H1, W1) uint8 device batch, letterboxed to exactly `shape1` (auto=False)."""
    x = torch.empty(len(ims), 3, shape1[0], shape1[1], dtype=torch.uint8, device=device)
    for i, im in enumerate(ims):
        nh, nw, top, left, _, _, full = letterbox_geometry(im.shape[:2], shape1)
        if tuple(full) != tuple(shape1):
            raise ValueError(f"letterbox of {im.shape[:2]} to {shape1} gives {full}")
        src = torch.from_numpy(np.ascontiguousarray(im)).to(device, non_blocking=True)
        ops.letterbox_u8(src, x, i, nh, nw, top, left, color)
    return x


class Detections:
    """Result container with the reference's data attributes (models/common.py:879-903)."""

    def __init__(self, ims, pred, files, times=(0.0, 0.0, 0.0), names=None, shape=None):
        d = pred[0].device if pred else torch.device("cpu")
        gn = [torch.tensor([*(im.shape[i] for i in [1, 0, 1, 0]), 1, 1], device=d) for im in ims]
        self.ims, self.pred, self.names, self.files, self.times = ims, pred, names, files, times
        self.xyxy = pred
        self.xywh = [torch.cat((_xyxy2xywh(x[:, :4]), x[:, 4:]), 1) for x in pred]
        self.xyxyn = [x / g for x, g in zip(self.xyxy, gn)]
        self.xywhn = [x / g for x, g in zip(self.xywh, gn)]
        self.n = len(self.pred)
        self.t = tuple(t * 1e3 / max(self.n, 1) for t in times)
        self.s = tuple(shape) if shape is not None else None

    def __len__(self):
        return self.n

    def tolist(self):
        return [Detections([self.ims[i]], [self.pred[i]], [self.files[i]], self.times, self.names, self.s) for i in range(self.n)]


def _xyxy2xywh(x):
    y = x.clone()
    y[..., 0] = (x[..., 0] + x[..., 2]) / 2
    y[..., 1] = (x[..., 1] + x[..., 3]) / 2
    y[..., 2] = x[..., 2] - x[..., 0]
    y[..., 3] = x[..., 3] - x[..., 1]
    return y


class AutoShape(nn.Module):
    conf = 0.25  # NMS confidence threshold
    iou = 0.45  # NMS IoU threshold
    agnostic = False
    multi_label = False
    classes = None
    max_det = 1000
    amp = False

    def __init__(self, model, verbose=True):
        super().__init__()
        for k in ("yaml", "nc", "hyp", "names", "stride", "abc"):
            if hasattr(model, k):
                setattr(self, k, getattr(model, k))
        self.dmb = type(model).__name__ == "DetectMultiBackend"
        self.pt = True
        self.model = model.eval()
        det = self.model.model.model[-1] if self.dmb else self.model.model[-1]
        det.inplace = False
        det.export = True  # no raw head tensors: the decode kernel then skips writing them

    @torch.no_grad()
    def forward(self, ims, size=640, augment=False, profile=False):
        t0 = time.perf_counter()
        if isinstance(size, int):
            size = (size, size)
        p = next(self.model.parameters())
        if isinstance(ims, torch.Tensor):
            return self.model(ims.to(p.device).type_as(p), augment=augment)
        n, ims = (len(ims), list(ims)) if isinstance(ims, (list, tuple)) else (1, [ims])
        shape0, shape1, files = [], [], []
        for i, im in enumerate(ims):
            f = f"image{i}"
            if isinstance(im, (str, Path)):
                if str(im).startswith("http"):
                    raise ValueError("AutoShape: URLs are not fetched here (no network I/O on the hot path); pass a decoded image")
                from PIL import Image

                im, f = np.asarray(Image.open(im)), im
            elif not isinstance(im, np.ndarray):  # PIL.Image
                im, f = np.asarray(im), getattr(im, "filename", f) or f
            files.append(Path(f).with_suffix(".jpg").name)
            if im.shape[0] < 5:  # CHW
                im = im.transpose((1, 2, 0))
            im = im[..., :3] if im.ndim == 3 else np.stack((im,) * 3, -1)  # enforce 3 channels (cv2.COLOR_GRAY2BGR replicates)
            s = im.shape[:2]
            shape0.append(s)
            g = max(size) / max(s)
            shape1.append([int(y * g) for y in s])
            ims[i] = np.ascontiguousarray(im)
        stride = int(self.stride.max()) if isinstance(self.stride, torch.Tensor) else int(self.stride)
        shape1 = [make_divisible(x, stride) for x in np.array(shape1).max(0)]
        x = letterbox_batch(ims, shape1, p.device)  # uint8; the model's first kernel divides by 255
        t1 = time.perf_counter()
        net = self.model.model if self.dmb else self.model   # the DetectionModel itself: it takes the uint8 batch (a DetectMultiBackend would cast it to half without the /255)
        y = net(x, augment=augment)
        t2 = time.perf_counter()
        rows, counts_t, counts = non_max_suppression_batched(y[0], self.conf, self.iou, self.classes, self.agnostic, self.multi_label, max_det=self.max_det)
        scale_boxes_batched(shape1, rows, counts_t, shape0)
        pred = [rows[i, :c] for i, c in enumerate(counts)]
        t3 = time.perf_counter()
        return Detections(ims, pred, files, (t1 - t0, t2 - t1, t3 - t2), getattr(self, "names", None), x.shape)
