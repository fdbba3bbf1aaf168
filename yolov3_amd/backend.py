"""``DetectMultiBackend`` for the PyTorch-checkpoint branch (reference models/common.py:432-768; the ``.pt`` branch
:471-476, ``forward`` :647-727, ``warmup`` :735).  The other eleven runtimes (ONNX, TensorRT, OpenVINO, ...) are foreign
inference engines and out of scope: a non-``.pt`` weights path raises."""
from __future__ import annotations

from pathlib import Path

import torch
from torch import nn

from .compat import attempt_load


class DetectMultiBackend(nn.Module):
    def __init__(self, weights="yolov3.pt", device=torch.device("cuda"), dnn=False, data=None, fp16=False, fuse=True):
        super().__init__()
        w = str(weights[0] if isinstance(weights, (list, tuple)) else weights)
        if isinstance(weights, nn.Module):
            model = weights.to(device)
        elif Path(w).suffix == ".pt":
            model = attempt_load(w, device=device, inplace=True, fuse=fuse)  # models/common.py:472
        else:
            raise NotImplementedError(f"{w}: only PyTorch .pt checkpoints run on the MI355X path (the other backends are foreign runtimes)")
        self.stride = max(int(model.stride.max()), 32)
        self.names = model.names if hasattr(model, "names") else {i: f"class{i}" for i in range(1000)}
        if fp16:
            model.half()  # models/common.py:475
        self.model = model
        self.pt, self.jit, self.onnx, self.engine, self.triton = True, False, False, False, False
        self.fp16, self.device, self.nhwc = fp16, device, False

    def forward(self, im, augment=False, visualize=False):
        if self.fp16 and im.dtype != torch.float16:
            im = im.half()  # models/common.py:650-651
        y = self.model(im, augment=augment, visualize=visualize)
        return list(y) if isinstance(y, (list, tuple)) else y  # tuple -> list, models/common.py:724-725

    def warmup(self, imgsz=(1, 3, 640, 640)):
        if self.device.type != "cpu":
            im = torch.empty(*imgsz, dtype=torch.half if self.fp16 else torch.float, device=self.device)
            self.forward(im)
