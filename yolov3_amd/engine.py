"""Static execution plans for the YOLOv3 graph on MI355X.

The reference walks ``nn.Sequential`` module by module (models/yolo.py:135-147) and lets ATen launch ~220
kernels per forward with NCHW tensors, separate BN/SiLU passes, ``torch.cat`` and ``nn.Upsample`` copies.
Here the graph is compiled ONCE per (batch, height, width, dtype) into a flat list of C-ABI launches over
pre-planned NHWC buffers:

* Conv+BN+SiLU(+residual) is one kernel (BN folded into the packed filter / bias at plan time, fp32 fold);
* ``Concat`` is zero-copy: producers write into channel slices of the destination buffer;
* ``nn.Upsample`` disappears: the producing 1x1 conv scatters each pixel to its 2x2 block of the concat slice;
* ``ZeroPad2d`` + ``MaxPool2d(2,1)`` (yolov3-tiny) is one pooling launch; SPP's three pools are one launch;
* activation buffers are recycled by lifetime so the working set stays inside the 256 MiB Infinity Cache as
  far as possible; weights are packed once and stay resident in HBM.

torch is used for device memory and the current stream only.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
import time
import weakref
from collections import OrderedDict
from dataclasses import dataclass, field

import torch
from torch import nn

from . import _lib, ops
from ._lib import Y3ConvDesc, Y3Tensor
from .common import SPP, Bottleneck, Concat, Conv, MaxPool2d, Upsample, ZeroPad2d


# ------------------------------------------------------------------------------------------- shape bookkeeping
def _conv_hw(hw, k, s):
    p = k // 2
    return tuple((v + 2 * p - k) // s + 1 for v in hw)


def _sources(i, f):
    fl = [f] if isinstance(f, int) else list(f)
    return [i + j if j < 0 else j for j in fl]


def graph_hw(model, h, w):
    """(h, w) of every top-level layer output for an (h, w) input (the reference finds strides by a dry run,
    models/yolo.py:219-222)."""
    sizes = []
    for i, m in enumerate(model.model):
        src = _sources(i, m.f)
        hw = (h, w) if src[0] < 0 else sizes[src[0]]
        mm = m[0] if isinstance(m, nn.Sequential) else m
        if isinstance(mm, Conv):
            hw = _conv_hw(hw, mm.conv.kernel_size[0], mm.conv.stride[0])
        elif isinstance(mm, Upsample):
            hw = (hw[0] * 2, hw[1] * 2)
        elif isinstance(mm, MaxPool2d):
            k, s, p = mm.kernel_size, mm.stride, mm.padding
            hw = tuple((v + 2 * p - k) // s + 1 for v in hw)
        elif isinstance(mm, ZeroPad2d):
            l, r, t, b = mm.padding
            hw = (hw[0] + t + b, hw[1] + l + r)
        sizes.append(hw)
    return sizes


def _out_channels(m):
    mm = m[-1] if isinstance(m, nn.Sequential) else m
    if isinstance(mm, Conv):
        return mm.conv.out_channels
    if isinstance(mm, Bottleneck):
        return mm.cv2.conv.out_channels
    if isinstance(mm, SPP):
        return mm.cv2.conv.out_channels
    return None


# ------------------------------------------------------------------------------------------- symbolic buffers
class Buf:
    """A flat activation buffer with a lifetime [first, last] in op indices; storage is assigned at finalize."""

    def __init__(self, numel, name=""):
        self.numel, self.name = int(numel), name
        self.first, self.last = None, None
        self.tensor = None

    def touch(self, t):
        if self.first is None:
            self.first = t
        self.last = t


@dataclass
class SView:
    buf: Buf
    n: int
    h: int
    w: int
    c: int
    pitch: int
    coff: int = 0

    def slice(self, coff, c):
        return SView(self.buf, self.n, self.h, self.w, c, self.pitch, self.coff + coff)

    def real(self) -> ops.View:
        return ops.View(self.buf.tensor, self.n, self.h, self.w, self.c, self.pitch, self.coff)


def _pad8(c):
    return (c + 7) // 8 * 8


# ------------------------------------------------------------------------------------------- weights
class ConvWeights:
    """Packed filter bank + fp32 bias of one (fused) convolution, resident on the device."""

    def __init__(self, filt, bias, cin, cout, k, s, act):
        self.filt, self.bias, self.cin, self.cout, self.k, self.s, self.act = filt, bias, cin, cout, k, s, act


def _fold(conv: nn.Conv2d, bn):
    """fp32 (weight OIHW, bias) of conv(+bn): the algebra of upstream fuse_conv_and_bn (reference models/yolo.py:168),
    evaluated on the device at plan time."""
    w = conv.weight.detach().float()
    b = conv.bias.detach().float() if conv.bias is not None else torch.zeros(w.shape[0], device=w.device)
    if bn is not None:
        scale = bn.weight.detach().float().div(torch.sqrt(bn.eps + bn.running_var.float()))
        w = torch.mm(torch.diag(scale), w.view(w.shape[0], -1)).view(w.shape)
        b_bn = bn.bias.detach().float() - bn.weight.detach().float().mul(bn.running_mean.float()).div(torch.sqrt(bn.running_var.float() + bn.eps))
        b = torch.mm(torch.diag(scale), b.reshape(-1, 1)).reshape(-1) + b_bn
    return w, b


KEEP_FOLDED = False  # tests set this to keep the fp32 folded (w, b) next to the packed bank


def make_conv_weights(conv: nn.Conv2d, bn, act: bool, dtype, cin_pad=None, cache: dict | None = None) -> ConvWeights:
    """Fold + pack one conv.  `cache` (PlanCache.weights) shares the packed bank between the plans of a model: rect-batch
    validation compiles dozens of (h, w) shapes (reference utils/dataloaders.py:548-570) and each used to re-pack 124 MB."""
    key = (id(conv), id(bn), bool(act), dtype, cin_pad)
    if cache is not None and key in cache:
        return cache[key]
    cw = _make_conv_weights(conv, bn, act, dtype, cin_pad)
    if cache is not None:
        cache[key] = cw
    return cw


def _make_conv_weights(conv: nn.Conv2d, bn, act: bool, dtype, cin_pad=None) -> ConvWeights:
    w, b = _fold(conv, bn)
    co, ci, k, _ = w.shape
    cin = cin_pad or _pad8(ci)
    cout = _pad8(co)
    filt = ops.pack_filter(w, cout, cin, dtype)
    bias = torch.zeros(cout, dtype=torch.float32, device=w.device)
    bias[:co] = b
    cw = ConvWeights(filt, bias, cin, cout, k, conv.stride[0], act)
    if KEEP_FOLDED:
        cw.folded = (w, b)
    return cw


# ------------------------------------------------------------------------------------------- plan
@dataclass
class _Launch:
    fn: object
    args: tuple
    keep: tuple = ()  # python objects that must outlive the launch (ctypes structs, tensors)
    label: str = ""
    flops: float = 0.0
    bytes: float = 0.0
    kernel: str = ""  # set when the launch is not a y3_conv2d_fwd (bench.py groups by it)


class Plan:
    """Compiled forward for fixed (N, H, W, dtype).  ``run`` issues the launches on the current stream."""

    def __init__(self, device, dtype, n, h, w):
        self.device, self.dtype, self.n, self.h, self.w = device, dtype, n, h, w
        self.bufs: list[Buf] = []
        self.steps: list = []  # (kind, payload) symbolic until finalize
        self.launches: list[_Launch] = []
        self.detect = None
        self.input_view = None
        self.out_views: dict = {}
        self.param_refs = []
        self.param_version = 0
        self.stem_x = None  # not None when layer 0 runs as the stem kernel and reads the caller's NCHW tensor (arguments are built per call)
        self.lock = threading.Lock()  # launches of one plan are enqueued atomically (its buffers are ordered by ONE stream)

    # -- building -----------------------------------------------------------------------------
    def new_buf(self, n, h, w, pitch, name=""):
        b = Buf(n * h * w * pitch, name)
        self.bufs.append(b)
        return b

    def new_view(self, n, h, w, c, name=""):
        return SView(self.new_buf(n, h, w, c, name), n, h, w, c, c, 0)

    def add(self, kind, reads, writes, **kw):
        t = len(self.steps)
        for v in reads:
            if v is not None:
                v.buf.touch(t)
        for v in writes:
            v.buf.touch(t)
        self.steps.append((kind, kw))

    def conv(self, x: SView, wts: ConvWeights, y: SView, residual: SView = None, ups=False, label=""):
        assert x.c == wts.cin, (label, x.c, wts.cin)
        assert y.c == wts.cout, (label, y.c, wts.cout)
        self.add("conv", [x, residual], [y], x=x, w=wts, y=y, res=residual, ups=ups, label=label)

    # -- storage + launch list ----------------------------------------------------------------
    def finalize(self):
        # lifetime-based storage assignment: a buffer is recycled once its last reader has been issued
        esz = torch.empty(0, dtype=self.dtype).element_size()
        free: list[torch.Tensor] = []
        by_first = sorted([b for b in self.bufs if b.first is not None], key=lambda b: b.first)
        releases: dict[int, list[Buf]] = {}
        for b in by_first:
            releases.setdefault(b.last, []).append(b)
        pending = list(by_first)
        total = 0
        pi = 0
        for t in range(len(self.steps) + 1):
            while pi < len(pending) and pending[pi].first == t:
                b = pending[pi]
                pi += 1
                best = None
                for k, tns in enumerate(free):
                    if tns.numel() >= b.numel and (best is None or tns.numel() < free[best].numel()):
                        best = k
                if best is not None and free[best].numel() <= 2 * b.numel + 4096:
                    b.tensor = free.pop(best)
                else:
                    b.tensor = torch.empty(b.numel, dtype=self.dtype, device=self.device)
                    total += b.numel * esz
            for b in releases.get(t, []):
                if not getattr(b, "pinned", False):
                    free.append(b.tensor)
        self.activation_bytes = total
        L = _lib.lib()
        dcode = ops.dtype_code(self.dtype)
        # scratch of the K-split conv form (small launches): one per plan (the plan's launches run in order on one stream)
        self.workspace = ops.conv_workspace(self.device) if self.dtype != torch.float32 and any(k == "conv" for k, _ in self.steps) else None
        ws_ptr, ws_bytes = (self.workspace.data_ptr(), self.workspace.numel()) if self.workspace is not None else (None, 0)
        for kind, kw in self.steps:
            if kind == "conv":
                x, y, w, res = kw["x"].real(), kw["y"].real(), kw["w"], kw["res"]
                d = Y3ConvDesc(dcode, w.k, w.s, _lib.Y3_ACT_SILU if w.act else _lib.Y3_ACT_NONE, int(kw["ups"]), _lib.Y3_ALGO_AUTO, w.cin, w.cout, 0, w.filt.numel())
                xt, yt = x.y3(), y.y3()
                rt = res.real().y3() if res is not None else None
                ho = (x.h + 2 * (w.k // 2) - w.k) // w.s + 1
                wo = (x.w + 2 * (w.k // 2) - w.k) // w.s + 1
                m = x.n * ho * wo
                self.launches.append(
                    _Launch(
                        L.y3_conv2d_fwd_ws,
                        (C.byref(d), C.byref(xt), w.filt.data_ptr(), w.bias.data_ptr(), C.byref(rt) if rt is not None else None, C.byref(yt), ws_ptr, ws_bytes),
                        keep=(d, xt, yt, rt, w),
                        label=kw["label"],
                        flops=2.0 * m * w.cout * w.cin * w.k * w.k,
                        bytes=esz * (x.n * x.h * x.w * w.cin + m * w.cout * (4 if kw["ups"] else 1) + (m * w.cout if res is not None else 0) + w.cout * w.cin * w.k * w.k),
                    )
                )
            elif kind == "stem":
                w, yv = kw["w"], kw["y"].real()
                yt = yv.y3()
                self.stem_x = True
                m = yv.n * yv.h * yv.w
                self.launches.append(
                    _Launch(
                        L.y3_stem_conv_fwd,
                        (_STEM_X, _STEM_SDT, yv.n, kw["cin"], yv.h, yv.w, _STEM_DIV, w.filt.data_ptr(), w.bias.data_ptr(), dcode,
                         _lib.Y3_ACT_SILU if w.act else _lib.Y3_ACT_NONE, C.byref(yt)),
                        keep=(yt, w),
                        label=kw["label"],
                        flops=2.0 * m * w.cout * _pad8(kw["cin"]) * 9,   # counted like the generic path (8 padded channels) so GFLOP/img stays comparable
                        bytes=esz * (m * kw["cin"] + m * w.cout + w.cout * kw["cin"] * 9),
                        kernel="stem_conv",
                    )
                )
            elif kind == "stem_pair":
                w0, w1, yv = kw["w0"], kw["w1"], kw["y"].real()
                yt = yv.y3()
                self.stem_x = True
                m1 = yv.n * yv.h * yv.w
                m0 = yv.n * kw["h"] * kw["w"]
                self.launches.append(
                    _Launch(
                        L.y3_stem_pair_fwd,
                        (_STEM_X, _STEM_SDT, yv.n, kw["cin"], kw["h"], kw["w"], _STEM_DIV, w0.filt.data_ptr(), w0.bias.data_ptr(),
                         _lib.Y3_ACT_SILU if w0.act else _lib.Y3_ACT_NONE, w1.filt.data_ptr(), w1.bias.data_ptr(), _lib.Y3_ACT_SILU if w1.act else _lib.Y3_ACT_NONE, dcode, C.byref(yt)),
                        keep=(yt, w0, w1),
                        label=kw["label"],
                        flops=2.0 * m0 * w0.cout * _pad8(kw["cin"]) * 9 + 2.0 * m1 * w1.cout * w1.cin * 9,   # counted like the two generic launches
                        bytes=esz * (m0 * kw["cin"] + m1 * w1.cout + w0.cout * kw["cin"] * 9 + w1.cout * w1.cin * 9),
                        kernel="stem_pair",
                    )
                )
            elif kind == "bneck_pair":
                xv, yv, w1, w2 = kw["x"].real(), kw["y"].real(), kw["w1"], kw["w2"]
                xt, yt = xv.y3(), yv.y3()
                m = xv.n * xv.h * xv.w
                cc, cm = xv.c, xv.c // 2
                self.launches.append(
                    _Launch(
                        L.y3_bneck_pair_fwd,
                        (C.byref(xt), w1.filt.data_ptr(), w1.bias.data_ptr(), _lib.Y3_ACT_SILU if w1.act else _lib.Y3_ACT_NONE, w2.filt.data_ptr(), w2.bias.data_ptr(),
                         _lib.Y3_ACT_SILU if w2.act else _lib.Y3_ACT_NONE, int(kw["add"]), dcode, C.byref(yt)),
                        keep=(xt, yt, w1, w2),
                        label=kw["label"],
                        flops=2.0 * m * (cm * cc + cc * cm * 9),                                  # counted like the two generic launches
                        bytes=esz * (m * cc + m * cc + cm * cc + cc * cm * 9),                   # x once, y once: the intermediate never leaves the CU
                        kernel="bneck_pair",
                    )
                )
            elif kind == "maxpool":
                xt, yt = kw["x"].real().y3(), kw["y"].real().y3()
                self.launches.append(_Launch(L.y3_maxpool2d, (C.byref(xt), C.byref(yt), dcode, kw["k"], kw["s"], kw["p"], kw["zr"], kw["zb"]), keep=(xt, yt), label=kw["label"]))
            elif kind == "spp":
                xt, yt = kw["x"].real().y3(), kw["y"].real().y3()
                self.launches.append(_Launch(L.y3_spp_pyramid, (C.byref(xt), C.byref(yt), dcode), keep=(xt, yt), label=kw["label"]))
            elif kind == "upsample":
                xt, yt = kw["x"].real().y3(), kw["y"].real().y3()
                self.launches.append(_Launch(L.y3_upsample2x, (C.byref(xt), C.byref(yt), dcode), keep=(xt, yt), label=kw["label"]))
            elif kind == "copy":
                xt, yt = kw["x"].real().y3(), kw["y"].real().y3()
                self.launches.append(_Launch(L.y3_copy_slice, (C.byref(xt), C.byref(yt), dcode), keep=(xt, yt), label=kw["label"]))
            else:
                raise AssertionError(kind)
        self.trace, self.steps = self.steps, None  # symbolic steps kept for the host-logic tests

    def conv_variant(self, ln) -> str:
        """kernel variant the library dispatches a conv launch of this plan to (asked from the library, nothing is launched)"""
        d, xt, yt, rt, _w = ln.keep
        name = C.create_string_buffer(64)
        _lib.check(_lib.lib().y3_conv2d_fwd_variant(C.byref(d), C.byref(xt), C.byref(yt), int(rt is not None), self.workspace.numel() if self.workspace is not None else 0, name, 64),
                   "y3_conv2d_fwd_variant")
        return name.value.decode()

    # -- execution -----------------------------------------------------------------------------
    def run_body(self, stream, stem=None):
        """Enqueue every launch on `stream`.  `stem` = (pointer, dtype code, divisor) of the caller's NCHW image for the stem
        kernels: substituted per call, so nothing shared is mutated (two threads may hold the same plan object)."""
        with self.lock:
            self.last_stem = stem   # profiling tools replay single launches (bench.per_kernel_times, _profile)
            for ln in self.launches:
                st = ln.fn(*self.resolved_args(ln, stem), stream)
                if st != 0:
                    _lib.check(st, ln.label or "launch")

    def resolved_args(self, ln, stem=None):
        """argument tuple of a launch with the per-call stem placeholders filled in"""
        if ln.kernel not in ("stem_conv", "stem_pair"):
            return ln.args
        stem = stem if stem is not None else getattr(self, "last_stem", None)
        if stem is None:
            raise RuntimeError("stem launch replayed before the plan ran once")
        return tuple(stem[0] if a is _STEM_X else stem[1] if a is _STEM_SDT else stem[2] if a is _STEM_DIV else a for a in ln.args)


_STEM_X, _STEM_SDT, _STEM_DIV = object(), object(), object()   # placeholders in a stem launch's argument tuple (filled per call)


def _param_version(params):
    return sum(p._version for p in params)


# ------------------------------------------------------------------------------------------- graph compiler
class _Compiler:
    def __init__(self, plan: Plan, dtype, training=False, wcache=None):
        self.plan, self.dtype, self.training, self.wcache = plan, dtype, training, wcache
        if training:
            raise NotImplementedError(
                "training-mode forward (batch-statistics BatchNorm + backward) is not implemented on the MI355X path yet; "
                "call model.eval() -- there is no PyTorch fallback"
            )

    def conv_unit(self, m: Conv, x: SView, y: SView = None, residual=None, ups=False, label="", cin_pad=None):
        p = self.plan
        w = make_conv_weights(m.conv, getattr(m, "bn", None), isinstance(m.act, nn.SiLU), self.dtype, cin_pad=cin_pad, cache=self.wcache)
        if y is None:
            ho, wo = _conv_hw((x.h, x.w), w.k, w.s)
            y = p.new_view(x.n, ho, wo, w.cout, label)
        p.conv(x, w, y, residual, ups, label)
        return y

    def bottleneck(self, m: Bottleneck, x: SView, y: SView = None, label=""):
        if _bneck_pair_eligible(m, x, self.dtype):
            # Bottleneck(64, 64) / Bottleneck(128, 128) (yolov3 layers 2 and 4, the 320x320 / 160x160 maps): one kernel, the C/2-channel
            # intermediate stays in LDS and x is read once
            p = self.plan
            w1 = make_conv_weights(m.cv1.conv, getattr(m.cv1, "bn", None), isinstance(m.cv1.act, nn.SiLU), self.dtype, cin_pad=x.c, cache=self.wcache)
            w2 = make_conv_weights(m.cv2.conv, getattr(m.cv2, "bn", None), isinstance(m.cv2.act, nn.SiLU), self.dtype, cin_pad=x.c // 2, cache=self.wcache)
            if y is None:
                y = p.new_view(x.n, x.h, x.w, x.c, label)
            p.add("bneck_pair", [x], [y], x=x, y=y, w1=w1, w2=w2, add=bool(m.add), label=label)
            return y
        t = self.conv_unit(m.cv1, x, label=label + ".cv1")
        return self.conv_unit(m.cv2, t, y=y, residual=x if m.add else None, label=label + ".cv2")

    def spp(self, m: SPP, x: SView, y: SView = None, label=""):
        p = self.plan
        c_ = m.cv1.conv.out_channels
        cat = p.new_view(x.n, x.h, x.w, 4 * c_, label + ".cat")
        self.conv_unit(m.cv1, x, y=cat.slice(0, c_), label=label + ".cv1")
        p.add("spp", [cat.slice(0, c_)], [cat.slice(c_, 3 * c_)], x=cat.slice(0, c_), y=cat.slice(c_, 3 * c_), label=label + ".pools")
        return self.conv_unit(m.cv2, cat, y=y, label=label + ".cv2")


def _bneck_pair_eligible(m, x, dtype) -> bool:
    """Bottleneck(C, C), C = 64 or 128: cv1 = Conv(C, C/2, 1, 1), cv2 = Conv(C/2, C, 3, 1) on a C-channel view: csrc/stem.hip
    (y3_bneck_pair_fwd) computes both with the intermediate kept in LDS.  Y3_BNECK_PAIR=0 restores the two generic launches, =64 keeps the
    kernel for C = 64 only (A/B runs)."""
    import os

    flag = os.environ.get("Y3_BNECK_PAIR", "1")
    if flag == "0" or dtype not in (torch.float16, torch.bfloat16) or x.c not in (64, 128) or (flag == "64" and x.c != 64):
        return False
    if x.n * x.h * x.w * x.pitch * 2 >= 2**31:   # one buffer descriptor per tensor in y3_bneck_pair_fwd (the generic path chunks the batch instead)
        return False
    c1, c2 = m.cv1.conv, m.cv2.conv
    c = x.c

    def plain(c, k):
        return c.kernel_size == (k, k) and c.stride == (1, 1) and c.padding == (k // 2, k // 2) and c.dilation == (1, 1) and c.groups == 1

    return (c1.in_channels == c and c1.out_channels == c // 2 and plain(c1, 1) and c2.in_channels == c // 2 and c2.out_channels == c and plain(c2, 3)
            and isinstance(m.cv1.act, (nn.SiLU, nn.Identity)) and isinstance(m.cv2.act, (nn.SiLU, nn.Identity)))


def _stem_eligible(m, dtype, srcs, input_consumers) -> bool:
    """Layer 0 = Conv(ch <= 4, <= 64 filters, 3, 1) fed by the image alone: csrc/stem.hip computes it straight from the
    caller's NCHW tensor (no NHWC copy of the image).  Y3_STEM=0 restores ingest + generic conv (A/B runs)."""
    import os

    c = m.conv
    return (os.environ.get("Y3_STEM", "1") != "0" and dtype in (torch.float16, torch.bfloat16) and list(srcs) == [-1] and list(input_consumers) == [0]
            and c.in_channels <= 4 and c.out_channels <= 64 and c.kernel_size == (3, 3) and c.stride == (1, 1) and c.padding == (1, 1)
            and c.dilation == (1, 1) and c.groups == 1)


def _stem_pair_eligible(layers, src, consumers, placed, fused_ups) -> bool:
    """Layers 0 and 1 = Conv(<=4, 32, 3, 1) -> Conv(32, 64, 3, 2), layer 0 consumed by layer 1 alone: csrc/stem.hip computes both
    with layer 0's output kept in LDS.  Y3_STEM_PAIR=0 restores stem + generic conv (A/B runs)."""
    import os

    if os.environ.get("Y3_STEM_PAIR", "1") == "0" or len(layers) < 2:
        return False
    m0, m1 = layers[0], layers[1]
    if not (isinstance(m0, Conv) and isinstance(m1, Conv)) or list(src[1]) != [0] or list(consumers[0]) != [1] or 0 in placed or 1 in fused_ups or 0 in fused_ups:
        return False
    c0, c1 = m0.conv, m1.conv
    return (c0.out_channels == 32 and c1.in_channels == 32 and c1.out_channels == 64 and c1.kernel_size == (3, 3) and c1.stride == (2, 2) and c1.padding == (1, 1)
            and c1.dilation == (1, 1) and c1.groups == 1 and isinstance(m0.act, (nn.SiLU, nn.Identity)) and isinstance(m1.act, (nn.SiLU, nn.Identity)))


def compile_model(model, n, h, w, dtype, device, wcache=None) -> Plan:
    from .yolo import Detect

    plan = Plan(device, dtype, n, h, w)
    comp = _Compiler(plan, dtype, training=model.training, wcache=wcache)
    layers = list(model.model)
    nl = len(layers)
    hw = graph_hw(model, h, w)
    src = [_sources(i, m.f) for i, m in enumerate(layers)]
    consumers = {i: [] for i in range(-1, nl)}
    for i, s in enumerate(src):
        for j in s:
            consumers[j].append(i)

    def kind(m):
        return m[0] if isinstance(m, nn.Sequential) else m

    ch = {}
    cin0 = _pad8(model.yaml.get("ch", 3))
    for i, m in enumerate(layers):
        k = kind(m)
        oc = _out_channels(m)
        if oc is None:
            if isinstance(k, Concat):
                oc = sum(ch[j] for j in src[i])
            elif isinstance(k, Detect):
                oc = 0
            else:
                oc = ch[src[i][0]] if src[i][0] >= 0 else cin0
        ch[i] = oc

    # ---- placement: where each layer's output lives (Concat destinations claim their sources) ----
    placed: dict[int, SView] = {}
    extra_copy = []  # (src layer, destination view) when a tensor feeds a second Concat
    for i, m in enumerate(layers):
        if isinstance(kind(m), Concat):
            ctot = ch[i]
            cbuf = placed.get(i) or plan.new_view(n, hw[i][0], hw[i][1], ctot, f"L{i}.concat")
            placed[i] = cbuf
            off = 0
            for j in src[i]:
                sl = cbuf.slice(off, ch[j])
                if j in placed or j < 0 or (ch[j] % 8) or (off % 8):
                    extra_copy.append((j, i, sl))
                else:
                    placed[j] = sl
                off += ch[j]

    # Upsample fed by a single-consumer Conv: the conv scatters straight into the upsample's home
    fused_ups = {}
    for i, m in enumerate(layers):
        if isinstance(kind(m), Upsample):
            j = src[i][0]
            if j >= 0 and isinstance(layers[j], Conv) and consumers[j] == [i] and j not in placed:
                fused_ups[j] = i
    # ZeroPad2d consumed only by a MaxPool2d: folded into the pool launch
    fused_pad = {}
    for i, m in enumerate(layers):
        if isinstance(kind(m), ZeroPad2d):
            cons = consumers[i]
            if len(cons) == 1 and isinstance(kind(layers[cons[0]]), MaxPool2d) and i not in placed:
                fused_pad[i] = cons[0]
            else:
                raise NotImplementedError("ZeroPad2d is only supported directly in front of a MaxPool2d (yolov3-tiny)")

    def home(i):
        if i not in placed:
            placed[i] = plan.new_view(n, hw[i][0], hw[i][1], ch[i], f"L{i}")
        return placed[i]

    x_in = plan.new_view(n, h, w, cin0, "input")
    x_in.buf.pinned = True
    plan.input_view = x_in
    out = {-1: x_in}

    pair = None   # layer-0 weights while layers 0 + 1 are emitted as one stem_pair launch
    pair_ok = not model.training and _stem_pair_eligible(layers, src, consumers, placed, fused_ups)   # before home() claims a buffer for layer 0
    for i, m in enumerate(layers):
        k = kind(m)
        ins = [out[j] for j in src[i]]
        lab = f"L{i}"
        if isinstance(k, Detect):
            heads = []
            for lvl, xv in enumerate(ins):
                conv = k.m[lvl]
                wts = make_conv_weights(conv, None, False, dtype, cin_pad=xv.c, cache=wcache)
                hv = plan.new_view(xv.n, xv.h, xv.w, wts.cout, f"{lab}.head{lvl}")
                hv.buf.pinned = True
                plan.conv(xv, wts, hv, label=f"{lab}.m{lvl}")
                heads.append(hv)
            plan.detect = (k, heads)
            continue
        if isinstance(k, Concat):
            for (j, ci, sl) in extra_copy:
                if ci == i:
                    plan.add("copy", [out[j]], [sl], x=out[j], y=sl, label=f"{lab}.copy{j}")
            out[i] = placed[i]
            continue
        if isinstance(k, Upsample):
            j = src[i][0]
            if j in fused_ups:
                out[i] = home(i)  # already written by the producing conv
            else:
                y = home(i)
                plan.add("upsample", [ins[0]], [y], x=ins[0], y=y, label=lab)
                out[i] = y
            continue
        if isinstance(k, ZeroPad2d):
            out[i] = ins[0]  # folded into the following pool
            continue
        if isinstance(k, MaxPool2d):
            j = src[i][0]
            zr = zb = 0
            if j in fused_pad:
                zr, zb = kind(layers[j]).padding[1], kind(layers[j]).padding[3]
            y = home(i)
            plan.add("maxpool", [ins[0]], [y], x=ins[0], y=y, k=k.kernel_size, s=k.stride, p=k.padding, zr=zr, zb=zb, label=lab)
            out[i] = y
            continue
        # conv-type layers
        if i in fused_ups:
            dst = home(fused_ups[i])
            comp.conv_unit(k, ins[0], y=dst, ups=True, label=lab, cin_pad=ins[0].c)
            out[i] = None
            continue
        y = home(i)
        if isinstance(m, nn.Sequential):
            x = ins[0]
            for r, sub in enumerate(m):
                last = r == len(m) - 1
                x = comp.bottleneck(sub, x, y=y if last else None, label=f"{lab}.{r}")
            out[i] = y
        elif isinstance(k, Conv) and i == 1 and pair:
            w0 = pair
            w1 = make_conv_weights(k.conv, getattr(k, "bn", None), isinstance(k.act, nn.SiLU), dtype, cin_pad=32, cache=wcache)
            plan.add("stem_pair", [], [y], w0=w0, w1=w1, y=y, cin=w0.cin, h=h, w=w, label="L0+L1")
            out[i] = y
        elif isinstance(k, Conv) and i == 0 and _stem_eligible(k, dtype, src[0], consumers[-1]):
            skey = ("stem", id(k.conv), id(getattr(k, "bn", None)), dtype)
            wts = wcache.get(skey) if wcache is not None else None
            if wts is None:
                cw, cb = _fold(k.conv, getattr(k, "bn", None))
                co = cw.shape[0]
                bias = torch.zeros(_pad8(co), dtype=torch.float32, device=cw.device)
                bias[:co] = cb
                wts = ConvWeights(ops.pack_filter_stem(cw, _pad8(co), dtype), bias, cw.shape[1], _pad8(co), 3, 1, isinstance(k.act, nn.SiLU))
                if KEEP_FOLDED:
                    wts.folded = (cw, cb)
                if wcache is not None:
                    wcache[skey] = wts
            if pair_ok:
                pair = wts          # emitted together with layer 1
                out[i] = None
            else:
                plan.add("stem", [], [y], w=wts, y=y, cin=wts.cin, label=lab)
                out[i] = y
        elif isinstance(k, Conv):
            out[i] = comp.conv_unit(k, ins[0], y=y, label=lab, cin_pad=ins[0].c)
        elif isinstance(k, Bottleneck):
            out[i] = comp.bottleneck(k, ins[0], y=y, label=lab)
        elif isinstance(k, SPP):
            out[i] = comp.spp(k, ins[0], y=y, label=lab)
        else:
            raise NotImplementedError(type(k).__name__)
    plan.out_views = out
    plan.finalize()
    plan.param_refs = list(model.parameters()) + list(model.buffers())
    plan.param_version = _param_version(plan.param_refs)
    return plan


# ------------------------------------------------------------------------------------------- entry points
def _engine_dtype(model) -> torch.dtype:
    p = next(model.parameters())
    if p.dtype not in (torch.float16, torch.bfloat16, torch.float32):
        raise TypeError(f"unsupported parameter dtype {p.dtype}")
    return p.dtype


def _decode_consts(det, dtype):
    """Host copies of (anchor w,h in pixels per level, stride per level), rounded through `dtype` exactly as
    Detect._make_grid does on tensors (reference models/yolo.py:112-123).  Computed once per plan."""
    stride = det.stride.detach().to("cpu", dtype)
    anchors_px = (det.anchors.detach().to("cpu", dtype) * stride.view(-1, 1, 1)).float()
    return [(anchors_px[l].reshape(-1).tolist(), float(stride[l])) for l in range(det.nl)]


def _decode_outputs(plan: Plan, det, heads, n, export=False, training=False):
    dtype, dev = plan.dtype, plan.device
    na, no = det.na, det.no
    rows = [na * hv.h * hv.w for hv in heads]
    total = sum(rows)
    consts = plan.__dict__.get("decode_consts")
    if consts is None:
        consts = plan.decode_consts = _decode_consts(det, dtype)
    z = None if training else torch.empty(n, total, no, dtype=dtype, device=dev)
    raws = []
    off = 0
    for lvl, hv in enumerate(heads):
        raw = None if export else torch.empty(n, na, hv.h, hv.w, no, dtype=dtype, device=dev)
        apx, stride = consts[lvl]
        ops.detect_decode(hv.real(), na, no, apx, stride, raw, z, off, total)
        raws.append(raw)
        off += rows[lvl]
    if training:
        return raws
    return (z,) if export else (z, raws)


class PlanCache:
    """Compiled plans of ONE model, kept OUT of the module (a plan holds ctypes argument blocks and tens of MB of activations:
    in `model.__dict__` it broke `deepcopy(model)` / `torch.save` / ModelEMA -- the reference's checkpoint path, train.py:470-488).
    Plans are keyed by (batch, h, w, dtype, mode, device, stream) and evicted least-recently-used; the packed filter banks are
    shared by all eval plans and live as long as the parameters are not modified."""

    MAX_EVAL = int(os.environ.get("Y3_MAX_PLANS", "6"))
    MAX_TRAIN = int(os.environ.get("Y3_MAX_TRAIN_PLANS", "2"))          # training slots = forwards that may be outstanding at once (train_engine.TrainSlot)
    MAX_TRAIN_SHAPES = int(os.environ.get("Y3_MAX_TRAIN_SHAPES", "24"))  # distinct (batch, h, w, dtype) training shapes kept compiled: the reference's multi-scale training
                                                                          # draws from ~21 sizes (train.py:394-399); a plan is views into its slot's arena, not buffers

    def __init__(self):
        self.plans: "OrderedDict" = OrderedDict()
        self.weights: dict = {}
        self.weights_version = None
        self.slots: dict = {}   # (dtype, device index) -> [TrainSlot] * MAX_TRAIN
        self.lock = threading.Lock()

    def clear(self):
        with self.lock:
            self.plans.clear()
            self.weights.clear()
            self.slots.clear()
            self.weights_version = None

    def get(self, key):
        plan = self.plans.get(key)
        if plan is not None:
            self.plans.move_to_end(key)
        return plan

    def train_slots(self, dtype, device, slot_cls):
        """the MAX_TRAIN training slots of (activation dtype, device): each owns one activation arena, sized for the largest shape it has run, and the filter banks"""
        key = (dtype, device.index)
        sl = self.slots.get(key)
        if sl is None:
            sl = self.slots[key] = [slot_cls(dtype, device, i) for i in range(self.MAX_TRAIN)]
        return sl

    def drop_train_slot(self, slot):
        """forget the compiled plans of one training slot that are stale (built on an arena / for Parameter objects the slot has replaced since)"""
        for k in [k for k, p in self.plans.items() if k[0] == "train" and getattr(p, "slot", None) is slot and p.slot_generation != slot.generation]:
            del self.plans[k]

    def put(self, key, plan):
        self.plans[key] = plan
        if key[0] != "train":
            same = [k for k in self.plans if k[0] != "train"]
            for k in same[: max(0, len(same) - self.MAX_EVAL)]:
                del self.plans[k]   # the activation pool / workspace go back to torch's allocator once the last launch that uses them has run
            return
        # training plans: one per (shape, slot) and at most MAX_TRAIN_SHAPES shapes (multi-scale training, a short last batch), least recently used shape first
        shapes = []
        for k in self.plans:
            if k[0] == "train" and k[1:6] not in shapes:
                shapes.append(k[1:6])
        for shp in shapes[: max(0, len(shapes) - self.MAX_TRAIN_SHAPES)]:
            for k in [k for k in self.plans if k[0] == "train" and k[1:6] == shp]:
                del self.plans[k]
        mine = [k for k in self.plans if k[0] == "train" and k[1:6] == key[1:6]]
        for k in mine[: max(0, len(mine) - self.MAX_TRAIN)]:
            del self.plans[k]


_PLAN_CACHES: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()
_PLAN_CACHES_LOCK = threading.Lock()


def plan_cache(model) -> PlanCache:
    with _PLAN_CACHES_LOCK:
        pc = _PLAN_CACHES.get(model)
        if pc is None:
            pc = _PLAN_CACHES[model] = PlanCache()
        return pc


def drop_plans(model):
    """forget compiled plans and packed filters (weights moved / re-typed / fused)"""
    with _PLAN_CACHES_LOCK:
        pc = _PLAN_CACHES.get(model)
    if pc is not None:
        pc.clear()


def run_model(model, x: torch.Tensor, profile=False):
    ops.require_gpu(x, "DetectionModel.forward")
    if x.dim() != 4:
        raise ValueError(f"expected a (bs, ch, h, w) image batch, got shape {tuple(x.shape)}")
    ch = model.yaml.get("ch", 3) if isinstance(getattr(model, "yaml", None), dict) else None
    if ch is not None and x.shape[1] != ch:   # the stem kernel is handed the raw pointer: a wrong channel count would read past the buffer
        raise ValueError(f"expected {ch} input channels, got a batch of shape {tuple(x.shape)}")
    if model.training:
        from .train_engine import run_model_train

        return run_model_train(model, x)
    dtype = _engine_dtype(model)
    n, c, h, w = x.shape
    stream = ops.stream_ptr()
    pc = plan_cache(model)
    key = ("eval", n, h, w, dtype, x.device.index, stream)
    with pc.lock:
        refs = list(model.parameters()) + list(model.buffers())
        version = (_param_version(refs), len(refs))
        if pc.weights_version != version:   # parameters were modified in place (or replaced) since the filters were packed
            for k in [k for k in pc.plans if k[0] != "train"]:   # training plans re-pack their banks from the fp32 masters every forward: they stay
                del pc.plans[k]
            pc.weights.clear()
            pc.weights_version = version
        plan = pc.get(key)
        if plan is None:
            with torch.no_grad():
                plan = compile_model(model, n, h, w, dtype, x.device, wcache=pc.weights)
            pc.put(key, plan)
    # uint8 images are normalised inside the first kernel: the `im.half(); im /= 255` of reference val.py:358-359 / detect.py:187-189 /
    # models/common.py:868 without a separate pass over the batch
    div = 255.0 if x.dtype == torch.uint8 else 1.0
    stem = None
    if plan.stem_x is not None:
        xc = x.contiguous()   # the launch reads the caller's tensor directly (stream-ordered: xc stays alive until its kernels are enqueued)
        stem = (C.c_void_p(xc.data_ptr()), C.c_int32(ops.dtype_code(xc.dtype)), C.c_float(div))
    else:
        ops.nchw_to_nhwc(x, plan.input_view.real(), div)
    if profile:
        _profile(plan, stream, stem)
    else:
        plan.run_body(stream, stem)
    det, heads = plan.detect
    return _decode_outputs(plan, det, heads, n, export=det.export, training=model.training)


def _profile(plan: Plan, stream, stem=None):
    """Per-launch timing table (the reference's model(x, profile=True), models/yolo.py:149-161)."""
    torch.cuda.synchronize()
    rows = []
    for ln in plan.launches:
        args = plan.resolved_args(ln, stem)
        t0 = time.perf_counter()
        for _ in range(10):
            ln.fn(*args, stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        rows.append((ln.label, dt * 1e3, ln.flops / dt / 1e12 if dt else 0, ln.bytes / dt / 1e9 if dt else 0))
    print(f"{'launch':>24s} {'ms':>9s} {'TFLOP/s':>9s} {'GB/s':>9s}")
    for r in rows:
        print(f"{r[0]:>24s} {r[1]:9.4f} {r[2]:9.1f} {r[3]:9.1f}")
    print(f"{'total':>24s} {sum(r[1] for r in rows):9.4f}")
    plan.last_profile = rows


def run_detect(det, xs):
    """Detect.forward on a list of NCHW maps (reference models/yolo.py:89-110)."""
    dtype = det.m[0].weight.dtype
    dev = xs[0].device
    ops.require_gpu(xs[0], "Detect.forward")
    n = xs[0].shape[0]
    plan = Plan(dev, dtype, n, 0, 0)
    heads, ins = [], []
    with torch.no_grad():
        for lvl, x in enumerate(xs):
            _, c, h, w = x.shape
            xin = plan.new_view(n, h, w, _pad8(c), f"in{lvl}")
            xin.buf.pinned = True
            wts = make_conv_weights(det.m[lvl], None, False, dtype, cin_pad=xin.c)
            hv = plan.new_view(n, h, w, wts.cout, f"head{lvl}")
            hv.buf.pinned = True
            plan.conv(xin, wts, hv, label=f"detect.m{lvl}")
            heads.append(hv)
            ins.append(xin)
        plan.finalize()
    for x, xin in zip(xs, ins):
        ops.nchw_to_nhwc(x.to(dtype), xin.real(), 1.0)
    plan.run_body(ops.stream_ptr())
    return _decode_outputs(plan, det, heads, n, export=det.export, training=det.training)


def run_single_layer(module, x: torch.Tensor):
    """Run one Conv / Bottleneck / SPP alone (NCHW in, NCHW out) through the same kernels (eval semantics)."""
    ops.require_gpu(x, type(module).__name__ + ".forward")
    dtype = next(module.parameters()).dtype
    n, c, h, w = x.shape
    plan = Plan(x.device, dtype, n, h, w)
    comp = _Compiler(plan, dtype, training=False)
    xin = plan.new_view(n, h, w, _pad8(c), "input")
    xin.buf.pinned = True
    with torch.no_grad():
        if isinstance(module, Conv):
            y = comp.conv_unit(module, xin, label="conv", cin_pad=xin.c)
        elif isinstance(module, Bottleneck):
            y = comp.bottleneck(module, xin, label="bottleneck")
        elif isinstance(module, SPP):
            y = comp.spp(module, xin, label="spp")
        else:
            raise NotImplementedError(type(module).__name__)
        y.buf.pinned = True
        plan.finalize()
    ops.nchw_to_nhwc(x.to(dtype), xin.real(), 1.0)
    plan.run_body(ops.stream_ptr())
    full = ops.nhwc_to_nchw(y.real())
    co = _out_channels(module)
    return full[:, :co]
