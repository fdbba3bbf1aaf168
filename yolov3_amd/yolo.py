"""Host-side mirror of reference models/yolo.py for the detection hot path: ``Detect`` (:69-123),
``DetectionModel`` (:190-292) and ``parse_model`` (:298-380) with the reference's constructor signatures,
attributes (``model``, ``save``, ``stride``, ``names``, ``yaml``, ``inplace``) and ``state_dict`` keys.
``forward`` executes on an MI355X through the HIP engine (yolov3_amd/engine.py); there is no PyTorch
compute path and CPU tensors are rejected loudly."""
from __future__ import annotations

import math
from copy import deepcopy
from pathlib import Path

import torch
import yaml
from torch import nn

from .common import SPP, Bottleneck, Concat, Conv, MaxPool2d, Upsample, ZeroPad2d

CFG_DIR = Path(__file__).resolve().parent / "cfg"

_MODULES = {
    "Conv": Conv,
    "Bottleneck": Bottleneck,
    "SPP": SPP,
    "Concat": Concat,
    "nn.Upsample": Upsample,
    "Upsample": Upsample,
    "nn.MaxPool2d": MaxPool2d,
    "MaxPool2d": MaxPool2d,
    "nn.ZeroPad2d": ZeroPad2d,
    "ZeroPad2d": ZeroPad2d,
}


def make_divisible(x, divisor):
    """ceil to a multiple of divisor (upstream ultralytics.utils.ops.make_divisible; reference models/yolo.py:348)."""
    if isinstance(divisor, torch.Tensor):
        divisor = int(divisor.max())
    return math.ceil(x / divisor) * divisor


class Detect(nn.Module):
    """Detection head (reference models/yolo.py:69-123): per level a 1x1 Conv2d(bias) to na*(nc+5) channels,
    reshaped to (bs, na, ny, nx, no); in eval mode additionally decoded to (bs, sum(na*ny*nx), no)."""

    stride = None
    dynamic = False
    export = False

    def __init__(self, nc=80, anchors=(), ch=(), inplace=True):
        super().__init__()
        self.nc = nc
        self.no = nc + 5
        self.nl = len(anchors)
        self.na = len(anchors[0]) // 2
        self.register_buffer("anchors", torch.tensor(anchors).float().view(self.nl, -1, 2))
        self.m = nn.ModuleList(nn.Conv2d(x, self.no * self.na, 1) for x in ch)
        self.inplace = inplace

    def forward(self, x):
        """x: list of nl NCHW feature maps -> training: list of raw (bs,na,ny,nx,no); eval: (z, raw list) or
        (z,) when ``export``.  Runs the head convolutions and the decode on the GPU through the HIP engine."""
        from .engine import run_detect

        return run_detect(self, x)


def parse_model(d, ch):
    """Model dict -> (nn.Sequential, save list); follows reference models/yolo.py:298-380 for the module
    kinds the yolov3*.yaml files use.  Each module gets ``.i/.f/.type/.np`` like the reference."""
    d = deepcopy(d)
    anchors, nc, gd, gw = d["anchors"], d["nc"], d["depth_multiple"], d["width_multiple"]
    if d.get("activation"):
        raise NotImplementedError("custom activations are not on the yolov3 hot path (default SiLU only)")
    na = (len(anchors[0]) // 2) if isinstance(anchors, list) else anchors
    no = na * (nc + 5)
    layers, save, c2 = [], [], ch[-1]
    for i, (f, n, m, args) in enumerate(d["backbone"] + d["head"]):
        args = [nc if a == "nc" else anchors if a == "anchors" else None if a == "None" else a for a in args]
        n = n_ = max(round(n * gd), 1) if n > 1 else n
        if m in ("Conv", "Bottleneck", "SPP"):
            c1, c2 = ch[f], args[0]
            if c2 != no:
                c2 = make_divisible(c2 * gw, 8)
            args = [c1, c2, *args[1:]]
        elif m == "Concat":
            c2 = sum(ch[x] for x in f)
        elif m == "Detect":
            args.append([ch[x] for x in f])
            if isinstance(args[1], int):
                args[1] = [list(range(args[1] * 2))] * len(f)
        elif m in _MODULES:
            c2 = ch[f]
        else:
            raise NotImplementedError(f"module '{m}' is not used by any yolov3*.yaml and has no MI355X implementation")
        cls = Detect if m == "Detect" else _MODULES[m]
        m_ = nn.Sequential(*(cls(*args) for _ in range(n))) if n > 1 else cls(*args)
        m_.i, m_.f, m_.type = i, f, ("models.yolo.Detect" if m == "Detect" else f"models.common.{m}" if not m.startswith("nn.") else f"torch.nn.modules.{m[3:]}")
        m_.np = sum(x.numel() for x in m_.parameters())
        save.extend(x % i for x in ([f] if isinstance(f, int) else f) if x != -1)
        layers.append(m_)
        if i == 0:
            ch = []
        ch.append(c2)
    return nn.Sequential(*layers), sorted(save)


def check_anchor_order(m: Detect):
    """Flip anchor order if it disagrees with the stride order (reference utils/autoanchor.py:16-23)."""
    a = m.anchors.prod(-1).mean(-1).view(-1)
    da = a[-1] - a[0]
    ds = m.stride[-1] - m.stride[0]
    if da and (da.sign() != ds.sign()):
        m.anchors[:] = m.anchors.flip(0)


def fuse_conv_and_bn(conv: nn.Conv2d, bn: nn.BatchNorm2d) -> nn.Conv2d:
    """Fold eval-mode BN into the conv, fp32 (upstream ultralytics.utils.torch_utils.fuse_conv_and_bn; reference
    models/yolo.py:168): w' = diag(g/sqrt(var+eps)) w ; b' = b_conv*g/sqrt(var+eps) + beta - g*mean/sqrt(var+eps)."""
    fused = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding, conv.dilation, conv.groups, bias=True)
    fused = fused.requires_grad_(False).to(conv.weight.device)
    w_conv = conv.weight.detach().view(conv.out_channels, -1)
    w_bn = torch.diag(bn.weight.detach().div(torch.sqrt(bn.eps + bn.running_var)))
    fused.weight.copy_(torch.mm(w_bn, w_conv).view(fused.weight.shape))
    b_conv = torch.zeros(conv.weight.shape[0], device=conv.weight.device) if conv.bias is None else conv.bias.detach()
    b_bn = bn.bias.detach() - bn.weight.detach().mul(bn.running_mean).div(torch.sqrt(bn.running_var + bn.eps))
    fused.bias.copy_(torch.mm(w_bn, b_conv.reshape(-1, 1)).reshape(-1) + b_bn)
    return fused


class BaseModel(nn.Module):
    """reference models/yolo.py:126-187."""

    @property
    def _plans(self):
        """compiled execution plans of this model: a view of the engine's cache, which lives OUTSIDE the module so that
        deepcopy / pickle / torch.save of the model (reference train.py:470-488, ModelEMA) never see device plans"""
        from .engine import plan_cache

        return plan_cache(self).plans

    def _drop_plans(self):
        from .engine import drop_plans

        drop_plans(self)

    def forward(self, x, profile=False, visualize=False):
        return self._forward_once(x, profile, visualize)

    def _forward_once(self, x, profile=False, visualize=False):
        """Whole-graph execution on the GPU (reference :135-147 walks modules one by one; here the graph is
        compiled once per input shape into a static plan of HIP launches)."""
        if visualize:
            raise NotImplementedError("feature visualisation is outside the accelerated hot path")
        from .engine import run_model

        return run_model(self, x, profile=profile)

    def fuse(self):
        """Fold every Conv's BatchNorm (reference :163-172)."""
        for m in self.modules():
            if isinstance(m, Conv) and hasattr(m, "bn"):
                m.conv = fuse_conv_and_bn(m.conv, m.bn)
                delattr(m, "bn")
        self._drop_plans()
        return self

    def info(self, verbose=False, img_size=640):
        n_p = sum(x.numel() for x in self.parameters())
        n_l = len(list(self.modules()))
        print(f"{type(self).__name__} summary: {n_l} modules, {n_p} parameters")

    def _apply(self, fn):
        """Also move Detect.stride (reference :178-187) and drop compiled plans (weights moved / re-typed)."""
        super()._apply(fn)
        m = self.model[-1]
        if isinstance(m, Detect) and m.stride is not None:
            m.stride = fn(m.stride)
        self._drop_plans()
        return self


class DetectionModel(BaseModel):
    """reference models/yolo.py:190-292: DetectionModel(cfg, ch=3, nc=None, anchors=None)."""

    def __init__(self, cfg="yolov3.yaml", ch=3, nc=None, anchors=None):
        super().__init__()
        if isinstance(cfg, dict):
            self.yaml = deepcopy(cfg)
        else:
            p = Path(cfg)
            if not p.exists() and (CFG_DIR / p.name).exists():
                p = CFG_DIR / p.name
            self.yaml_file = p.name
            with open(p, encoding="ascii", errors="ignore") as f:
                self.yaml = yaml.safe_load(f)
        ch = self.yaml["ch"] = self.yaml.get("ch", ch)
        if nc and nc != self.yaml["nc"]:
            self.yaml["nc"] = nc
        if anchors:
            self.yaml["anchors"] = round(anchors)
        self.model, self.save = parse_model(deepcopy(self.yaml), ch=[ch])
        self.names = [str(i) for i in range(self.yaml["nc"])]
        self.inplace = self.yaml.get("inplace", True)

        m = self.model[-1]
        if isinstance(m, Detect):
            m.inplace = self.inplace
            # the reference dry-runs a 256x256 zero image (:219-222); the strides follow from shape bookkeeping
            from .engine import graph_hw

            s = 256
            hw = graph_hw(self, s, s)
            m.stride = torch.tensor([s / hw[j][0] for j in m.f])
            check_anchor_order(m)
            m.anchors /= m.stride.view(-1, 1, 1)
            self.stride = m.stride
            self._initialize_biases()
        # upstream initialize_weights (:229): BN eps / momentum
        for mod in self.modules():
            if isinstance(mod, nn.BatchNorm2d):
                mod.eps = 1e-3
                mod.momentum = 0.03

    def forward(self, x, augment=False, profile=False, visualize=False):
        if augment:
            return self._forward_augment(x)
        return self._forward_once(x, profile, visualize)

    def _forward_augment(self, x):
        """Test-time augmentation (reference :239-276): the batch at scales 1 / 0.83 / 0.67, the middle one mirrored left-right, each through the
        engine; the decoded predictions de-scaled / de-mirrored and concatenated without the stride-32 rows of the full-size pass and the stride-8
        rows of the smallest one (_clip_augmented).  The mirror + resize + pad and the de-scale + row selection are one kernel each
        (csrc/val_edge.hip); the concatenated tensor is written once."""
        if self.training:
            raise RuntimeError("augmented inference is an eval-mode path (the reference indexes the decoded prediction of eval mode)")
        from . import ops

        img_size = x.shape[-2:]
        scales, flips = [1, 0.83, 0.67], [None, 3, None]
        gs = int(self.stride.max())
        det = self.model[-1]
        nl, no = det.nl, det.no
        g = sum(4**k for k in range(nl))
        strides = [int(v) for v in self.stride.tolist()]
        # the row windows first (the decoded prediction of an (h, w) pass has na * sum_l (h / s_l)(w / s_l) rows): the result is allocated once and every
        # pass is de-scaled into its window as soon as it is through the engine
        windows = []
        for i, si in enumerate(scales):
            h, w = (img_size if si == 1 else (math.ceil(d * si / gs) * gs for d in img_size))
            rows = det.na * sum((h // s) * (w // s) for s in strides)
            lo, hi = 0, rows
            if i == 0:
                hi = rows - (rows // g) * 1                    # without the tail: the coarsest level of the full-size pass
            if i == len(scales) - 1:
                lo = (rows // g) * 4 ** (nl - 1)               # without the head: the finest level of the smallest pass
            windows.append((rows, lo, hi - lo))
        out, off = None, 0
        for (rows, lo, n), si, fi in zip(windows, scales, flips):
            yi = self._forward_once(ops.scale_img(x, si, gs=gs, flip_lr=fi == 3))[0]
            if yi.shape[1] != rows or yi.shape[2] != no:
                raise RuntimeError(f"augmented pass at scale {si}: {tuple(yi.shape)} rows, expected {rows} x {no}")
            if out is None:
                out = torch.empty(yi.shape[0], sum(w[2] for w in windows), no, dtype=yi.dtype, device=yi.device)
            ops.descale_pred_into(yi.contiguous(), lo, n, si, fi, img_size, out, off)
            off += n
        return out, None

    def _initialize_biases(self, cf=None):
        """Detect bias prior (reference :282-292): obj += log(8/(640/s)^2), cls += log(0.6/(nc-0.99999))."""
        m = self.model[-1]
        for mi, s in zip(m.m, m.stride):
            b = mi.bias.view(m.na, -1)
            b.data[:, 4] += math.log(8 / (640 / s) ** 2)
            b.data[:, 5 : 5 + m.nc] += math.log(0.6 / (m.nc - 0.99999)) if cf is None else torch.log(cf / cf.sum())
            mi.bias = torch.nn.Parameter(b.view(-1), requires_grad=True)


Model = DetectionModel  # reference alias (models/yolo.py:295)
