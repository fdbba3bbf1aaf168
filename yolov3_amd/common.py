"""Layer containers mirroring reference models/common.py (Conv :57-81, Bottleneck :150-165, SPP :267-290,
Concat :416-428): same constructor arguments, same sub-module names, hence the same ``state_dict`` keys
(``conv.weight``, ``bn.*``, ``cv1.*``, ``cv2.*``) so reference checkpoints map one-to-one.

These modules hold parameters only.  The arithmetic runs in the HIP engine (yolov3_amd/engine.py) which
executes the whole graph from a static plan; a module called on its own runs a one-layer plan through the
same kernels.  Nothing here computes with PyTorch ops."""
from __future__ import annotations

import torch
from torch import nn


def autopad(k, p=None, d=1):
    """'same' padding for kernel k (reference models/common.py:48-54)."""
    if d > 1:
        k = d * (k - 1) + 1 if isinstance(k, int) else [d * (x - 1) + 1 for x in k]
    if p is None:
        p = k // 2 if isinstance(k, int) else [x // 2 for x in k]
    return p


class _EngineLayer(nn.Module):
    """Stand-alone call support: run this layer alone through the HIP engine on an NCHW tensor."""

    def forward(self, x):
        from .engine import run_single_layer

        return run_single_layer(self, x)


class Conv(_EngineLayer):
    """Conv2d(bias=False) + BatchNorm2d + SiLU (reference models/common.py:57-81).  After ``fuse`` the BN is
    folded: ``conv`` carries a bias and ``bn`` is gone (reference models/yolo.py:163-172)."""

    default_act = nn.SiLU()

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, d=1, act=True):
        super().__init__()
        if g != 1 or d != 1:
            raise NotImplementedError("grouped / dilated Conv is not used by the yolov3 models and has no HIP kernel")
        if k not in (1, 3) or s not in (1, 2):
            raise NotImplementedError(f"Conv k={k}, s={s}: the HIP path implements k in (1,3), s in (1,2) (all yolov3*.yaml shapes)")
        self.conv = nn.Conv2d(c1, c2, k, s, autopad(k, p, d), groups=g, dilation=d, bias=False)
        self.bn = nn.BatchNorm2d(c2)
        self.act = self.default_act if act is True else act if isinstance(act, nn.Module) else nn.Identity()
        if not isinstance(self.act, (nn.SiLU, nn.Identity)):
            raise NotImplementedError("only SiLU / Identity activations have a fused HIP epilogue")

    @property
    def fused(self) -> bool:
        return not hasattr(self, "bn")


class Bottleneck(_EngineLayer):
    """x + cv2(cv1(x)) when shortcut and c1 == c2 (reference models/common.py:150-165)."""

    def __init__(self, c1, c2, shortcut=True, g=1, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_, c2, 3, 1, g=g)
        self.add = shortcut and c1 == c2


class SPP(_EngineLayer):
    """cv2(cat([x, mp5(x), mp9(x), mp13(x)])) with x = cv1(x) (reference models/common.py:267-290)."""

    def __init__(self, c1, c2, k=(5, 9, 13)):
        super().__init__()
        if tuple(k) != (5, 9, 13):
            raise NotImplementedError("the HIP SPP pyramid kernel implements k=(5,9,13) (yolov3-spp.yaml)")
        c_ = c1 // 2
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_ * (len(k) + 1), c2, 1, 1)
        self.k = tuple(k)


class Concat(nn.Module):
    """Channel concatenation (reference models/common.py:416-428).  In the engine it is zero-copy: producers
    write straight into channel slices of the destination."""

    def __init__(self, dimension=1):
        super().__init__()
        if dimension != 1:
            raise NotImplementedError("Concat along channels only")
        self.d = dimension


class Upsample(nn.Module):
    """nn.Upsample(None, 2, 'nearest') stand-in (reference models/yolov3.yaml:43,51); parameter-free."""

    def __init__(self, size=None, scale_factor=2, mode="nearest"):
        super().__init__()
        if size is not None or int(scale_factor) != 2 or mode != "nearest":
            raise NotImplementedError("only nearest x2 upsampling is used by yolov3 and implemented in HIP")
        self.scale_factor, self.mode = 2, "nearest"


class MaxPool2d(nn.Module):
    """nn.MaxPool2d(k, s, p) stand-in (reference models/yolov3-tiny.yaml:21-32); parameter-free."""

    def __init__(self, kernel_size, stride=None, padding=0):
        super().__init__()
        self.kernel_size, self.stride, self.padding = kernel_size, stride or kernel_size, padding


class ZeroPad2d(nn.Module):
    """nn.ZeroPad2d([l, r, t, b]) stand-in (reference models/yolov3-tiny.yaml:31); only right/bottom padding."""

    def __init__(self, padding):
        super().__init__()
        l, r, t, b = padding
        if l or t:
            raise NotImplementedError("ZeroPad2d: only right/bottom padding is implemented (yolov3-tiny)")
        self.padding = (l, r, t, b)
