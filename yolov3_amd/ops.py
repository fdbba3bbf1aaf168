"""Thin host wrappers over the C ABI (include/yolov3_hip.h): torch supplies device memory and the stream,
nothing else.  Activations are NHWC views (`View`) over flat torch buffers."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import math

import torch

from . import _lib
from ._lib import Y3ConvDesc, Y3NmsParams, Y3Tensor, check

DTYPE_CODE = {torch.float16: _lib.Y3_F16, torch.bfloat16: _lib.Y3_BF16, torch.float32: _lib.Y3_F32, torch.uint8: _lib.Y3_U8}


def dtype_code(dt: torch.dtype) -> int:
    try:
        return DTYPE_CODE[dt]
    except KeyError:
        raise TypeError(f"dtype {dt} is not supported by the MI355X path (float16 / bfloat16 / float32)") from None


def require_gpu(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(
            f"{what}: tensor is on {t.device}; the yolov3_amd hot path runs only on an MI355X (HIP) device. "
            "There is no CPU / PyTorch fallback."
        )


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def tune_set(key: str, value: int):
    """run-time knob of the library (y3_tune_set: A/B hooks, test coverage of size-gated forms); process-wide"""
    check(_lib.lib().y3_tune_set(key.encode(), int(value)), "y3_tune_set")


def tune_get(key: str) -> int:
    v = _lib.lib().y3_tune_get(key.encode())
    if v == -(2**63):
        raise _lib.Y3Error(_lib.lib().y3_last_error().decode(errors="replace"))
    return int(v)


def tune_reset():
    _lib.lib().y3_tune_reset()


@dataclass
class View:
    """NHWC view: channels [coff, coff+c) of a (n, h, w, pitch) buffer."""

    buf: torch.Tensor  # flat storage, at least n*h*w*pitch elements
    n: int
    h: int
    w: int
    c: int
    pitch: int
    coff: int = 0

    def y3(self) -> Y3Tensor:
        ptr = self.buf.data_ptr() + self.coff * self.buf.element_size()
        return Y3Tensor(ptr, self.n, self.h, self.w, self.c, self.pitch)

    def slice(self, coff: int, c: int) -> "View":
        assert coff + c <= self.c
        return View(self.buf, self.n, self.h, self.w, c, self.pitch, self.coff + coff)

    def as_nhwc(self) -> torch.Tensor:
        """torch view (n,h,w,c) for tests/debug."""
        full = self.buf[: self.n * self.h * self.w * self.pitch].view(self.n, self.h, self.w, self.pitch)
        return full[..., self.coff : self.coff + self.c]

    @staticmethod
    def alloc(n, h, w, c, dtype, device, pitch=None) -> "View":
        pitch = pitch or c
        return View(torch.empty(n * h * w * pitch, dtype=dtype, device=device), n, h, w, c, pitch, 0)


def packed_filter_elems(cout: int, cin: int, k: int) -> int:
    return int(_lib.lib().y3_packed_filter_elems(cout, cin, k))


def pack_filter(w_oihw: torch.Tensor, cout: int, cin: int, dtype: torch.dtype) -> torch.Tensor:
    """OIHW fp32 weights (device) -> packed filter bank for y3_conv2d_fwd, zero padded to (cout, cin)."""
    require_gpu(w_oihw, "pack_filter")
    w = w_oihw.detach().to(torch.float32).contiguous()
    co, ci, k, _ = w.shape
    out = torch.empty(packed_filter_elems(cout, cin, k), dtype=dtype, device=w.device)
    check(_lib.lib().y3_pack_filter(w.data_ptr(), co, ci, k, cout, cin, dtype_code(dtype), out.data_ptr(), stream_ptr()), "y3_pack_filter")
    return out


def conv_workspace(device) -> torch.Tensor:
    """Scratch of the K-split form of the persistent 3x3 kernel (y3_conv2d_fwd_ws: fp32 slabs for small launches): allocated once here, owned by whoever runs convs on ONE
    stream at a time (a compiled plan keeps its own)."""
    return torch.zeros(int(_lib.lib().y3_conv_workspace_bytes()), dtype=torch.uint8, device=device)


def conv2d(x: View, filt: torch.Tensor, bias: torch.Tensor, y: View, k: int, stride: int, act: bool, residual: View | None = None, upsample2x: bool = False,
           algo: int = _lib.Y3_ALGO_AUTO, in_dilation: int = 0, workspace: torch.Tensor | None = None):
    d = Y3ConvDesc(dtype_code(x.buf.dtype), k, stride, _lib.Y3_ACT_SILU if act else _lib.Y3_ACT_NONE, int(upsample2x), algo, x.c, y.c, in_dilation, filt.numel())
    xt, yt = x.y3(), y.y3()
    rt = residual.y3() if residual is not None else None
    if workspace is not None:
        check(
            _lib.lib().y3_conv2d_fwd_ws(C.byref(d), C.byref(xt), filt.data_ptr(), bias.data_ptr(), C.byref(rt) if rt is not None else None, C.byref(yt), workspace.data_ptr(),
                                        workspace.numel(), stream_ptr()),
            "y3_conv2d_fwd_ws",
        )
        return
    check(
        _lib.lib().y3_conv2d_fwd(C.byref(d), C.byref(xt), filt.data_ptr(), bias.data_ptr(), C.byref(rt) if rt is not None else None, C.byref(yt), stream_ptr()),
        "y3_conv2d_fwd",
    )


def conv_variant(x: View, y: View, k: int, stride: int, residual: bool = False, upsample2x: bool = False, algo: int = _lib.Y3_ALGO_AUTO, in_dilation: int = 0,
                 workspace_bytes: int = 0) -> str:
    """Name of the kernel variant the library's dispatcher picks for this problem (nothing is launched)."""
    d = Y3ConvDesc(dtype_code(x.buf.dtype), k, stride, _lib.Y3_ACT_NONE, int(upsample2x), algo, x.c, y.c, in_dilation)
    xt, yt = x.y3(), y.y3()
    name = C.create_string_buffer(64)
    check(_lib.lib().y3_conv2d_fwd_variant(C.byref(d), C.byref(xt), C.byref(yt), int(residual), workspace_bytes, name, 64), "y3_conv2d_fwd_variant")
    return name.value.decode()


def pack_filter_stem(w_oihw: torch.Tensor, cout: int, dtype: torch.dtype) -> torch.Tensor:
    """OIHW fp32 weights (cin <= 4, 3x3) -> the stem kernel's [cout_pad32][3][16] bank."""
    require_gpu(w_oihw, "pack_filter_stem")
    w = w_oihw.detach().to(torch.float32).contiguous()
    co, ci, k, _ = w.shape
    assert k == 3 and ci <= 4
    out = torch.empty(int(_lib.lib().y3_packed_filter_stem_elems(cout)), dtype=dtype, device=w.device)
    check(_lib.lib().y3_pack_filter_stem(w.data_ptr(), co, ci, cout, dtype_code(dtype), out.data_ptr(), stream_ptr()), "y3_pack_filter_stem")
    return out


def stem_conv(x_nchw: torch.Tensor, filt: torch.Tensor, bias: torch.Tensor, y: View, act: bool, divisor: float = 1.0):
    """First-layer 3x3 s1 conv straight from the NCHW image (u8 / f16 / bf16 / f32) into the NHWC view y."""
    require_gpu(x_nchw, "stem_conv")
    x = x_nchw.contiguous()
    n, c, h, w = x.shape
    yt = y.y3()
    check(_lib.lib().y3_stem_conv_fwd(x.data_ptr(), dtype_code(x.dtype), n, c, h, w, float(divisor), filt.data_ptr(), bias.data_ptr() if bias is not None else None,
                                      dtype_code(y.buf.dtype), _lib.Y3_ACT_SILU if act else _lib.Y3_ACT_NONE, C.byref(yt), stream_ptr()), "y3_stem_conv_fwd")


def stem_conv_stats_rows(n: int, h: int, w: int) -> int:
    return int(_lib.lib().y3_stem_conv_stats_rows(n, h, w))


def stem_conv_stats(x_nchw: torch.Tensor, filt: torch.Tensor, bias: torch.Tensor, y: View, stat_rows: torch.Tensor, capacity_rows: int, divisor: float = 1.0) -> int:
    """stem_conv without activation + one row of (sum, sum of squares) per filter and block in stat_rows (fp32): the training form of layer 0."""
    require_gpu(x_nchw, "stem_conv_stats")
    x = x_nchw.contiguous()
    n, c, h, w = x.shape
    yt = y.y3()
    rows = C.c_int64(0)
    check(_lib.lib().y3_stem_conv_fwd_stats(x.data_ptr(), dtype_code(x.dtype), n, c, h, w, float(divisor), filt.data_ptr(), bias.data_ptr() if bias is not None else None,
                                            dtype_code(y.buf.dtype), _lib.Y3_ACT_NONE, C.byref(yt), stat_rows.data_ptr(), int(capacity_rows), C.byref(rows), stream_ptr()),
          "y3_stem_conv_fwd_stats")
    return int(rows.value)


def stem_conv_stats_only(x_nchw: torch.Tensor, filt: torch.Tensor, like: View, stat_rows: torch.Tensor, capacity_rows: int, divisor: float = 1.0) -> int:
    """the statistics rows of stem_conv_stats WITHOUT writing the conv output (`like` gives its shape and dtype): layer 0 of the training step by recomputation"""
    require_gpu(x_nchw, "stem_conv_stats_only")
    x = x_nchw.contiguous()
    n, c, h, w = x.shape
    lt = like.y3()
    rows = C.c_int64(0)
    check(_lib.lib().y3_stem_conv_stats_only(x.data_ptr(), dtype_code(x.dtype), n, c, h, w, float(divisor), filt.data_ptr(), dtype_code(like.buf.dtype), C.byref(lt),
                                             stat_rows.data_ptr(), int(capacity_rows), C.byref(rows), stream_ptr()), "y3_stem_conv_stats_only")
    return int(rows.value)


def stem_conv_bn(x_nchw: torch.Tensor, filt: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, act: int, y: View, divisor: float = 1.0):
    """y = act(scale * round(conv0(x)) + shift) straight from the image: what bn_act_fwd would write from the stored conv output, which is never stored"""
    require_gpu(x_nchw, "stem_conv_bn")
    x = x_nchw.contiguous()
    n, c, h, w = x.shape
    yt = y.y3()
    check(_lib.lib().y3_stem_conv_fwd_bn(x.data_ptr(), dtype_code(x.dtype), n, c, h, w, float(divisor), filt.data_ptr(), scale.data_ptr(), shift.data_ptr(), int(act),
                                         dtype_code(y.buf.dtype), C.byref(yt), stream_ptr()), "y3_stem_conv_fwd_bn")


def stem_bn_bwd_wgrad_recompute(x_nchw: torch.Tensor, filt: torch.Tensor, dy: View, scale, shift, mean, invstd, act: int, sums: torch.Tensor, dgamma, dbeta, dw: torch.Tensor,
                                workspace: torch.Tensor, divisor: float = 1.0):
    """stem_bn_bwd_wgrad without a stored u: both passes recompute it from the image (filt: the stem-packed filters of the forward)"""
    require_gpu(x_nchw, "stem_bn_bwd_wgrad_recompute")
    x = x_nchw.contiguous()
    n, c, h, w = x.shape
    if dw.dtype != torch.float32 or not dw.is_contiguous() or tuple(dw.shape) != (32, c, 3, 3):
        raise TypeError("stem_bn_bwd_wgrad_recompute: dw must be a contiguous fp32 (32, cin, 3, 3) tensor")
    gt = dy.y3()
    check(_lib.lib().y3_stem_bn_bwd_wgrad_recompute(x.data_ptr(), dtype_code(x.dtype), n, c, h, w, float(divisor), filt.data_ptr(), C.byref(gt), scale.data_ptr(), shift.data_ptr(),
                                                    mean.data_ptr(), invstd.data_ptr(), dtype_code(dy.buf.dtype), int(act), sums.data_ptr(),
                                                    dgamma.data_ptr() if dgamma is not None else None, dbeta.data_ptr() if dbeta is not None else None, dw.data_ptr(),
                                                    workspace.data_ptr(), workspace.numel(), stream_ptr()), "y3_stem_bn_bwd_wgrad_recompute")


def stem_bwd_workspace(device) -> torch.Tensor:
    return torch.empty(int(_lib.lib().y3_stem_bn_bwd_wgrad_workspace_bytes()), dtype=torch.uint8, device=device)


def stem_bn_bwd_wgrad(x_nchw: torch.Tensor, u: View, dy: View, scale, shift, mean, invstd, act: int, sums: torch.Tensor, dgamma, dbeta, dw: torch.Tensor,
                      workspace: torch.Tensor, divisor: float = 1.0):
    """Layer 0 backward (no data gradient): BatchNorm + activation backward of dy and the filter gradient dw (fp32 OIHW) in one pass over
    (u, dy) after the reduction pass -- du is never written.  32 filters, <= 3 image channels."""
    require_gpu(x_nchw, "stem_bn_bwd_wgrad")
    x = x_nchw.contiguous()
    n, c, h, w = x.shape
    if dw.dtype != torch.float32 or not dw.is_contiguous() or tuple(dw.shape) != (32, c, 3, 3):
        raise TypeError("stem_bn_bwd_wgrad: dw must be a contiguous fp32 (32, cin, 3, 3) tensor")
    ut, gt = u.y3(), dy.y3()
    check(_lib.lib().y3_stem_bn_bwd_wgrad(x.data_ptr(), dtype_code(x.dtype), n, c, h, w, float(divisor), C.byref(ut), C.byref(gt), scale.data_ptr(), shift.data_ptr(),
                                          mean.data_ptr(), invstd.data_ptr(), dtype_code(u.buf.dtype), int(act), sums.data_ptr(),
                                          dgamma.data_ptr() if dgamma is not None else None, dbeta.data_ptr() if dbeta is not None else None, dw.data_ptr(),
                                          workspace.data_ptr(), workspace.numel(), stream_ptr()), "y3_stem_bn_bwd_wgrad")


def pack_filter_dgrad(w_oihw: torch.Tensor, cout: int, cin: int, dtype: torch.dtype) -> torch.Tensor:
    """OIHW fp32 weights -> filter bank of the data-gradient conv (cin filters over (kh, kw, cout), flipped taps)."""
    require_gpu(w_oihw, "pack_filter_dgrad")
    w = w_oihw.detach().to(torch.float32).contiguous()
    co, ci, k, _ = w.shape
    out = torch.empty(packed_filter_elems(cin, cout, k), dtype=dtype, device=w.device)
    check(_lib.lib().y3_pack_filter_dgrad(w.data_ptr(), co, ci, k, cout, cin, dtype_code(dtype), out.data_ptr(), stream_ptr()), "y3_pack_filter_dgrad")
    return out


def bneck_pair(x: View, filt1: torch.Tensor, bias1: torch.Tensor, act1: bool, filt2: torch.Tensor, bias2: torch.Tensor, act2: bool, add: bool, y: View):
    """Bottleneck(C, C), C = 64 or 128: y = [x +] cv2(cv1(x)), cv1 1x1 C -> C/2, cv2 3x3 C/2 -> C, the intermediate kept in LDS (csrc/stem.hip)."""
    xt, yt = x.y3(), y.y3()
    check(_lib.lib().y3_bneck_pair_fwd(C.byref(xt), filt1.data_ptr(), bias1.data_ptr(), _lib.Y3_ACT_SILU if act1 else _lib.Y3_ACT_NONE, filt2.data_ptr(), bias2.data_ptr(),
                                       _lib.Y3_ACT_SILU if act2 else _lib.Y3_ACT_NONE, int(bool(add)), dtype_code(x.buf.dtype), C.byref(yt), stream_ptr()), "y3_bneck_pair_fwd")


def last_conv_variant() -> str:
    """variant name of the last conv / data-gradient launch of this thread (tests)"""
    buf = C.create_string_buffer(64)
    check(_lib.lib().y3_conv_last_variant(buf, 64), "y3_conv_last_variant")
    return buf.value.decode()


def conv2d_dgrad_s2(w_oihw: torch.Tensor, du: View, gx: View, accumulate: bool):
    """Data gradient of a 3x3 stride-2 conv through the four output-parity class convolutions (f16/bf16)."""
    dt = du.buf.dtype
    L = _lib.lib()
    w = w_oihw.detach().to(torch.float32).contiguous()
    co, ci, k, _ = w.shape
    assert k == 3
    packed = torch.empty(int(L.y3_packed_filter_dgrad_s2_elems(du.c, gx.c)), dtype=dt, device=w.device)
    check(L.y3_pack_filter_dgrad_s2(w.data_ptr(), co, ci, du.c, gx.c, dtype_code(dt), packed.data_ptr(), stream_ptr()), "y3_pack_filter_dgrad_s2")
    dut, gxt = du.y3(), gx.y3()
    check(L.y3_conv2d_dgrad_s2(dtype_code(dt), C.byref(dut), packed.data_ptr(), C.byref(gxt) if accumulate else None, C.byref(gxt), stream_ptr()), "y3_conv2d_dgrad_s2")


def conv2d_wgrad_workspace_bytes(x: View, cout: int, k: int, stride: int) -> int:
    d = Y3ConvDesc(dtype_code(x.buf.dtype), k, stride, 0, 0, 0, x.c, cout, 0)
    xt = x.y3()
    return int(_lib.lib().y3_conv2d_wgrad_workspace_bytes(C.byref(d), C.byref(xt)))


def conv2d_wgrad(x: View, du: View, k: int, stride: int, cout_real: int, cin_real: int, want_bias: bool = False, alloc=None, workspace=None):
    """Filter gradient (cout_real, cin_real, k, k) fp32 (+ bias gradient) of a conv with input x and output-gradient du.
    `alloc(shape)` supplies the output tensors (the training plan's per-backward gradient arena); default torch.empty.  `workspace`: a caller-owned uint8 buffer
    for the split-K slabs (the training slot keeps one for all its layers: launches of one stream use it in order); default a fresh allocation per call."""
    d = Y3ConvDesc(dtype_code(x.buf.dtype), k, stride, 0, 0, 0, x.c, du.c, 0)
    if alloc is None:
        def alloc(shape):
            return torch.empty(shape, dtype=torch.float32, device=x.buf.device)
    dw = alloc((cout_real, cin_real, k, k))
    db = alloc((cout_real,)) if want_bias else None
    xt, dt = x.y3(), du.y3()
    need = int(_lib.lib().y3_conv2d_wgrad_workspace_bytes(C.byref(d), C.byref(xt)))
    ws = workspace if workspace is not None and workspace.numel() >= need else torch.empty(need, dtype=torch.uint8, device=x.buf.device)
    check(_lib.lib().y3_conv2d_wgrad(C.byref(d), C.byref(xt), C.byref(dt), cout_real, cin_real, dw.data_ptr(), db.data_ptr() if db is not None else None,
                                     ws.data_ptr(), need, stream_ptr()),
          "y3_conv2d_wgrad")
    return dw, db


def conv2d_wgrad_plan(x: View, cout: int, k: int, stride: int):
    """(tile edge, pixel slices, xcd-grouped) of the filter-gradient launch for this shape (y3_conv2d_wgrad_plan: dry run)"""
    d = Y3ConvDesc(dtype_code(x.buf.dtype), k, stride, 0, 0, 0, x.c, cout, 0)
    xt = x.y3()
    tile, slices, xg = C.c_int32(0), C.c_int64(0), C.c_int32(0)
    check(_lib.lib().y3_conv2d_wgrad_plan(C.byref(d), C.byref(xt), C.byref(tile), C.byref(slices), C.byref(xg)), "y3_conv2d_wgrad_plan")
    return int(tile.value), int(slices.value), int(xg.value)


BN_PARTIAL_ROWS = 512


def bn_scratch(c: int, device) -> torch.Tensor:
    """fp64 scratch for y3_bn_stats / y3_bn_act_bwd: totals + per-block partial rows (Y3_BN_SCRATCH_DOUBLES)."""
    return torch.zeros((1 + BN_PARTIAL_ROWS) * 2 * c, dtype=torch.float64, device=device)


def nchw_to_nhwc(src: torch.Tensor, out: View, divisor: float = 1.0):
    require_gpu(src, "nchw_to_nhwc")
    src = src.contiguous()
    n, c, h, w = src.shape
    ot = out.y3()
    check(_lib.lib().y3_nchw_to_nhwc(src.data_ptr(), dtype_code(src.dtype), n, c, h, w, float(divisor), dtype_code(out.buf.dtype), C.byref(ot), stream_ptr()), "y3_nchw_to_nhwc")


def nhwc_to_nchw(src: View) -> torch.Tensor:
    dst = torch.empty(src.n, src.c, src.h, src.w, dtype=src.buf.dtype, device=src.buf.device)
    st = src.y3()
    check(_lib.lib().y3_nhwc_to_nchw(C.byref(st), dtype_code(src.buf.dtype), dst.data_ptr(), stream_ptr()), "y3_nhwc_to_nchw")
    return dst


def maxpool2d(x: View, y: View, k: int, stride: int, pad: int, zpad_r: int = 0, zpad_b: int = 0):
    xt, yt = x.y3(), y.y3()
    check(_lib.lib().y3_maxpool2d(C.byref(xt), C.byref(yt), dtype_code(x.buf.dtype), k, stride, pad, zpad_r, zpad_b, stream_ptr()), "y3_maxpool2d")


def maxpool2d_bwd(x: View, dy: View, dx: View, k: int, stride: int, pad: int, zpad_r: int = 0, zpad_b: int = 0, accumulate: bool = False):
    """dx (+)= backward of MaxPool2d(k, stride, pad) (+ right / bottom zero pad) through the indexed two-pass form (a byte of scratch per output element)"""
    xt, gt, dt = x.y3(), dy.y3(), dx.y3()
    need = int(_lib.lib().y3_maxpool2d_bwd_workspace_bytes(C.byref(xt), k, stride, pad, zpad_r, zpad_b))
    ws = torch.empty(max(need, 1), dtype=torch.uint8, device=x.buf.device)
    check(_lib.lib().y3_maxpool2d_bwd_ws(C.byref(xt), C.byref(gt), C.byref(dt), dtype_code(x.buf.dtype), k, stride, pad, zpad_r, zpad_b, int(bool(accumulate)), ws.data_ptr(), need,
                                         stream_ptr()), "y3_maxpool2d_bwd_ws")


def spp_pyramid(x: View, y3c: View):
    xt, yt = x.y3(), y3c.y3()
    check(_lib.lib().y3_spp_pyramid(C.byref(xt), C.byref(yt), dtype_code(x.buf.dtype), stream_ptr()), "y3_spp_pyramid")


def upsample2x(x: View, y: View):
    xt, yt = x.y3(), y.y3()
    check(_lib.lib().y3_upsample2x(C.byref(xt), C.byref(yt), dtype_code(x.buf.dtype), stream_ptr()), "y3_upsample2x")


def copy_slice(x: View, y: View):
    xt, yt = x.y3(), y.y3()
    check(_lib.lib().y3_copy_slice(C.byref(xt), C.byref(yt), dtype_code(x.buf.dtype), stream_ptr()), "y3_copy_slice")


def detect_decode(head: View, na: int, no: int, anchors_px, stride: float, raw: torch.Tensor | None, z: torch.Tensor | None, row_offset: int, total_rows: int):
    ht = head.y3()
    arr = (C.c_float * (na * 2))(*[float(v) for v in anchors_px])
    check(
        _lib.lib().y3_detect_decode(
            C.byref(ht), dtype_code(head.buf.dtype), na, no, arr, float(stride), raw.data_ptr() if raw is not None else None, z.data_ptr() if z is not None else None,
            row_offset, total_rows, stream_ptr(),
        ),
        "y3_detect_decode",
    )


_nms_ws_cache: dict = {}


def nms_raw(pred: torch.Tensor, conf_thres: float, iou_thres: float, classes, agnostic: bool, multi_label: bool, max_det: int, max_nms: int = 30000,
            max_wh: float = 7680.0):
    """Batched NMS on device.  Returns (rows (bs,max_det,6) fp32, counts list[int]).  One D2H copy (counts+status)."""
    require_gpu(pred, "non_max_suppression")
    pred = pred.contiguous()
    bs, n_rows, no = pred.shape
    nc = no - 5
    dev = pred.device
    cls_t = torch.tensor(list(classes), dtype=torch.int32, device=dev) if classes is not None else None
    p = Y3NmsParams(float(iou_thres), float(conf_thres), int(bool(multi_label)), int(bool(agnostic)), int(max_det), int(max_nms), float(max_wh),
                    0 if cls_t is None else cls_t.numel())
    L = _lib.lib()
    rows = torch.empty(bs, max_det, 6, dtype=torch.float32, device=dev)
    meta = torch.empty(bs + 2, dtype=torch.int32, device=dev)
    capacity = 0
    for attempt in range(3):
        need = int(L.y3_nms_workspace_bytes(bs, n_rows, nc, C.byref(p), capacity))
        key = (dev.index, need)
        ws = _nms_ws_cache.get(key)
        if ws is None:
            ws = torch.empty(need, dtype=torch.uint8, device=dev)
            if need <= (1 << 30):
                _nms_ws_cache[key] = ws
        check(
            L.y3_nms(pred.data_ptr(), dtype_code(pred.dtype), bs, n_rows, nc, C.byref(p), cls_t.data_ptr() if cls_t is not None else None, rows.data_ptr(),
                     meta.data_ptr(), meta.data_ptr() + bs * 4, capacity, ws.data_ptr(), need, stream_ptr()),
            "y3_nms",
        )
        m = meta.tolist()  # the single device->host sync of the call
        nms_raw.last_candidates = m[bs + 1]   # candidates (rows above conf_thres, one per class under multi_label) of the whole batch: bench.py reports it
        if m[bs] == 0:
            return rows, m[:bs]
        capacity = max(m[bs + 1], 1)  # overflow: rerun with room for every candidate
    raise _lib.Y3Error("y3_nms: candidate capacity overflow persisted")


def scale_boxes_raw(rows: torch.Tensor, img_stride: int, row_stride: int, counts: torch.Tensor | None, bs: int, max_rows: int, params: torch.Tensor):
    require_gpu(rows, "scale_boxes")
    check(_lib.lib().y3_scale_boxes(rows.data_ptr(), int(img_stride), int(row_stride), counts.data_ptr() if counts is not None else None, int(bs), int(max_rows),
                                    params.data_ptr(), stream_ptr()), "y3_scale_boxes")


def match_detections_raw(dets: torch.Tensor, img_stride: int, row_stride: int, counts: torch.Tensor | None, bs: int, max_det: int, labels: torch.Tensor,
                         offsets: torch.Tensor, iouv: torch.Tensor) -> torch.Tensor:
    require_gpu(dets, "process_batch")
    correct = torch.empty(bs, max_det, iouv.numel(), dtype=torch.uint8, device=dets.device)
    check(_lib.lib().y3_match_detections(dets.data_ptr(), int(img_stride), int(row_stride), counts.data_ptr() if counts is not None else None, int(bs), int(max_det),
                                         labels.data_ptr() if labels.numel() else None, offsets.data_ptr(), iouv.data_ptr(), int(iouv.numel()), correct.data_ptr(), stream_ptr()),
          "y3_match_detections")
    return correct


def scale_img(img: torch.Tensor, ratio: float = 1.0, same_shape: bool = False, gs: int = 32, flip_lr: bool = False) -> torch.Tensor:
    """upstream scale_img (the resize + pad of reference models/yolo.py:246), fused with the left-right mirror of the same line: (n, c, h, w) ->
    (n, c, ceil(h ratio / gs) gs, ceil(w ratio / gs) gs).  ratio 1.0 without a mirror returns `img` itself, like upstream."""
    if ratio == 1.0 and not flip_lr:
        return img
    require_gpu(img, "scale_img")
    if img.dim() != 4 or img.dtype not in (torch.float32, torch.float16, torch.bfloat16):
        raise TypeError("scale_img expects a floating (n, c, h, w) batch")
    img = img.contiguous()
    n, c, h, w = img.shape
    ih, iw = (h, w) if ratio == 1.0 else (int(h * ratio), int(w * ratio))
    oh, ow = (ih, iw) if (same_shape or ratio == 1.0) else (math.ceil(h * ratio / gs) * gs, math.ceil(w * ratio / gs) * gs)
    out = torch.empty(n, c, oh, ow, dtype=img.dtype, device=img.device)
    check(_lib.lib().y3_scale_img(img.data_ptr(), dtype_code(img.dtype), n, c, h, w, ih, iw, oh, ow, int(bool(flip_lr)), 0.447, out.data_ptr(), stream_ptr()), "y3_scale_img")
    return out


def descale_pred_into(pred: torch.Tensor, row0: int, nrows: int, scale: float, flip, img_size, out: torch.Tensor, out_row0: int):
    """rows [row0, row0 + nrows) of the decoded prediction (bs, rows, no) of one augmentation pass, de-scaled / de-mirrored (reference models/yolo.py:253-267),
    into rows [out_row0, ...) of the concatenated result."""
    require_gpu(pred, "_descale_pred")
    if pred.dim() != 3 or out.dim() != 3 or pred.dtype != out.dtype or not pred.is_contiguous() or not out.is_contiguous() or pred.shape[0] != out.shape[0] or pred.shape[2] != out.shape[2]:
        raise TypeError("descale_pred_into expects contiguous (bs, rows, no) tensors of one dtype")
    check(_lib.lib().y3_descale_pred(pred.data_ptr(), dtype_code(pred.dtype), pred.shape[0], pred.shape[1], pred.shape[2], int(row0), int(nrows), float(scale), int(flip or 0),
                                     float(img_size[0]), float(img_size[1]), out.data_ptr(), out.shape[1], int(out_row0), stream_ptr()), "y3_descale_pred")


def letterbox_u8(src_hwc: torch.Tensor, dst_batch: torch.Tensor, index: int, new_h: int, new_w: int, top: int, left: int, color: int = 114):
    """One image (h0, w0, >=3) uint8 on the device -> image `index` of the (n, 3, H1, W1) uint8 batch (resize + pad + transpose)."""
    require_gpu(src_hwc, "letterbox")
    if src_hwc.dtype != torch.uint8 or dst_batch.dtype != torch.uint8 or src_hwc.dim() != 3 or dst_batch.dim() != 4 or not src_hwc.is_contiguous() or not dst_batch.is_contiguous():
        raise TypeError("letterbox expects a contiguous (h, w, c) uint8 image and a contiguous (n, 3, H, W) uint8 batch")
    h0, w0, cs = src_hwc.shape
    check(_lib.lib().y3_letterbox_u8(src_hwc.data_ptr(), h0, w0, cs, dst_batch.data_ptr(), int(index), dst_batch.shape[2], dst_batch.shape[3], int(new_h), int(new_w), int(top),
                                     int(left), int(color), stream_ptr()), "y3_letterbox_u8")


def conv2d_stats_rows(x: View, y: View, k: int, stride: int, workspace: torch.Tensor | None = None) -> int:
    """rows of the statistics buffer a conv2d_stats launch of this shape writes (depends on the dispatched tile variant)."""
    d = Y3ConvDesc(dtype_code(x.buf.dtype), k, stride, _lib.Y3_ACT_NONE, 0, _lib.Y3_ALGO_AUTO, x.c, y.c, 0)
    xt, yt = x.y3(), y.y3()
    if workspace is not None:
        rows = int(_lib.lib().y3_conv2d_fwd_stats_rows_ws(C.byref(d), C.byref(xt), C.byref(yt), workspace.numel()))
    else:
        rows = int(_lib.lib().y3_conv2d_fwd_stats_rows(C.byref(d), C.byref(xt), C.byref(yt)))
    if rows < 0:
        check(-1, "y3_conv2d_fwd_stats_rows")
    return rows


def conv1x1_bnin_rows(u_in: View, y_in: View, y: View, has_shortcut: bool) -> int:
    """statistics rows of a conv1x1_bnin_stats launch of this shape, or -1 when the library's input-transform form (csrc/conv_1x1s.h) does not cover it"""
    d = Y3ConvDesc(dtype_code(u_in.buf.dtype), 1, 1, _lib.Y3_ACT_NONE, 0, _lib.Y3_ALGO_AUTO, u_in.c, y.c, 0)
    # geometry only (nothing is dereferenced): the views may not be bound to memory yet (TrainPlan builds before its arena exists)
    ut, it, yt = (Y3Tensor(4096, v.n, v.h, v.w, v.c, v.pitch) for v in (u_in, y_in, y))
    return int(_lib.lib().y3_conv2d_fwd_bnin_rows(C.byref(d), C.byref(ut), C.byref(it), C.byref(yt), int(bool(has_shortcut))))


def conv1x1_bnin_stats(u_in: View, in_scale: torch.Tensor, in_shift: torch.Tensor, in_act: int, shortcut: View | None, y_in: View, filt: torch.Tensor, bias: torch.Tensor, y: View,
                       stat_rows: torch.Tensor, capacity_rows: int) -> int:
    """y_in = act(in_scale * u_in + in_shift) (+ shortcut), stored once, and y = conv1x1(y_in) with statistics rows -- one launch (the producing layer's BatchNorm applied on the
    way into its 1x1 consumer: no normalise pass, no second read of y_in)."""
    d = Y3ConvDesc(dtype_code(u_in.buf.dtype), 1, 1, _lib.Y3_ACT_NONE, 0, _lib.Y3_ALGO_AUTO, u_in.c, y.c, 0, filt.numel())
    ut, it, yt = u_in.y3(), y_in.y3(), y.y3()
    st = shortcut.y3() if shortcut is not None else None
    n = C.c_int64(0)
    check(_lib.lib().y3_conv2d_fwd_bnin_stats(C.byref(d), C.byref(ut), in_scale.data_ptr(), in_shift.data_ptr(), int(in_act), C.byref(st) if st is not None else None, C.byref(it),
                                              filt.data_ptr(), bias.data_ptr(), C.byref(yt), stat_rows.data_ptr(), int(capacity_rows), C.byref(n), stream_ptr()), "y3_conv2d_fwd_bnin_stats")
    return int(n.value)


def conv2d_stats(x: View, filt: torch.Tensor, bias: torch.Tensor, y: View, k: int, stride: int, stat_rows: torch.Tensor, capacity_rows: int,
                 workspace: torch.Tensor | None = None) -> int:
    """y = conv(x) (no activation) + per-(pixel tile, wave) rows of (sum, sum of squares) per filter in stat_rows (fp32)."""
    d = Y3ConvDesc(dtype_code(x.buf.dtype), k, stride, _lib.Y3_ACT_NONE, 0, _lib.Y3_ALGO_AUTO, x.c, y.c, 0, filt.numel())
    xt, yt = x.y3(), y.y3()
    n = C.c_int64(0)
    if workspace is not None:
        check(_lib.lib().y3_conv2d_fwd_stats_ws(C.byref(d), C.byref(xt), filt.data_ptr(), bias.data_ptr(), C.byref(yt), stat_rows.data_ptr(), int(capacity_rows), C.byref(n),
                                                workspace.data_ptr(), workspace.numel(), stream_ptr()), "y3_conv2d_fwd_stats_ws")
        return int(n.value)
    check(_lib.lib().y3_conv2d_fwd_stats(C.byref(d), C.byref(xt), filt.data_ptr(), bias.data_ptr(), C.byref(yt), stat_rows.data_ptr(), int(capacity_rows), C.byref(n), stream_ptr()),
          "y3_conv2d_fwd_stats")
    return int(n.value)


class PackJobs:
    """Every layer's filter banks in one launch (y3_pack_filter_jobs): `add` registers a layer and returns its persistent (forward bank,
    data-gradient bank) tensors, `run` re-packs all of them from the current fp32 weights.  The job table lives on the device and is
    rebuilt only when a weight tensor moved (data_ptr changed)."""

    def __init__(self, dtype: torch.dtype, device):
        self.dtype, self.device = dtype, device
        self.jobs = []          # (weight param, fwd bank | None, dgrad bank | None, cout, cin)
        self._table, self._ptrs, self._blocks = None, None, 0

    def add(self, w: torch.Tensor, cout: int, cin: int, want_fwd: bool = True, want_dgrad: bool = True):
        co, ci, k, _ = w.shape
        # zero-filled ONCE: y3_pack_filter_jobs writes only the elements that come from a weight (the row / K padding of a bank stays zero)
        fwd = torch.zeros(packed_filter_elems(cout, cin, k), dtype=self.dtype, device=self.device) if want_fwd else None
        dg = torch.zeros(packed_filter_elems(cin, cout, k), dtype=self.dtype, device=self.device) if want_dgrad else None
        self.jobs.append((w, fwd, dg, cout, cin))
        self._table = None
        return fwd, dg

    def run(self):
        if not self.jobs:
            return
        ptrs = [w.data_ptr() for w, *_ in self.jobs]
        if self._table is None or ptrs != self._ptrs:
            import struct

            L = _lib.lib()
            rows, first = [], 0
            for (w, fwd, dg, cout, cin), ptr in zip(self.jobs, ptrs):
                if w.dtype != torch.float32 or not w.is_contiguous():
                    raise TypeError("PackJobs expects contiguous fp32 master weights")
                co, ci, k, _ = w.shape
                rows.append(struct.pack("<QQQ6i", ptr, fwd.data_ptr() if fwd is not None else 0, dg.data_ptr() if dg is not None else 0, co, ci, k, cout, cin, first))
                first += int(L.y3_pack_job_blocks(k, cout, cin, int(fwd is not None), int(dg is not None)))
            host = torch.frombuffer(bytearray(b"".join(rows)), dtype=torch.uint8)
            self._table = host.to(self.device)
            self._ptrs, self._blocks = ptrs, first
        check(_lib.lib().y3_pack_filter_jobs(self._table.data_ptr(), len(self.jobs), self._blocks, dtype_code(self.dtype), stream_ptr()), "y3_pack_filter_jobs")


def stem_pair(x_nchw: torch.Tensor, filt0: torch.Tensor, bias0: torch.Tensor, act0: bool, filt1: torch.Tensor, bias1: torch.Tensor, act1: bool, y: View, divisor: float = 1.0):
    """Conv(3->32, 3, 1) -> Conv(32->64, 3, 2) straight from the NCHW image into the NHWC view y (layer 0's output stays in LDS)."""
    require_gpu(x_nchw, "stem_pair")
    x = x_nchw.contiguous()
    n, c, h, w = x.shape
    yt = y.y3()
    check(_lib.lib().y3_stem_pair_fwd(x.data_ptr(), dtype_code(x.dtype), n, c, h, w, float(divisor), filt0.data_ptr(), bias0.data_ptr(), _lib.Y3_ACT_SILU if act0 else _lib.Y3_ACT_NONE,
                                      filt1.data_ptr(), bias1.data_ptr(), _lib.Y3_ACT_SILU if act1 else _lib.Y3_ACT_NONE, dtype_code(y.buf.dtype), C.byref(yt), stream_ptr()),
          "y3_stem_pair_fwd")


def pack_filter_pair(w_oihw: torch.Tensor, cout: int, cin: int, dtype: torch.dtype):
    """(forward bank, data-gradient bank) of one layer from its OIHW fp32 weights in one launch (training step)."""
    require_gpu(w_oihw, "pack_filter_pair")
    w = w_oihw.detach().to(torch.float32).contiguous()
    co, ci, k, _ = w.shape
    fwd = torch.empty(packed_filter_elems(cout, cin, k), dtype=dtype, device=w.device)
    dg = torch.empty(packed_filter_elems(cin, cout, k), dtype=dtype, device=w.device)
    check(_lib.lib().y3_pack_filter_pair(w.data_ptr(), co, ci, k, cout, cin, dtype_code(dtype), fwd.data_ptr(), dg.data_ptr(), stream_ptr()), "y3_pack_filter_pair")
    return fwd, dg
