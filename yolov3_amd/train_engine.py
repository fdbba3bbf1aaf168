"""Training-mode execution of the YOLOv3 graph on MI355X: forward with batch-statistics BatchNorm and the full
backward (data gradients, filter gradients, BN parameter gradients) as HIP launches, wrapped in ONE
torch.autograd.Function so that ``loss.backward()``, torch optimizers and DistributedDataParallel work unchanged
(the reference's training loop, train.py:402-422, only needs ``model(imgs)`` + autograd).

Reference semantics: models/common.py:75 (Conv = act(bn(conv(x))) with batch stats, eps 1e-3, momentum 0.03),
:165 (residual add), :428 (cat), nn.Upsample / nn.MaxPool2d, models/yolo.py:96-98 (Detect head conv + view/permute).

Layout/plan: same NHWC activations and zero-copy Concat as the inference engine; every activation (pre-BN ``u`` and
post-activation ``y``) is kept for the backward -- 288 GB of HBM makes recomputation unnecessary at batch 64.
Gradient buffers mirror the activation buffers (Concat sources are channel slices of the Concat's gradient buffer),
are zero-filled at the start of the backward and every producer ACCUMULATES into them, so fan-out (a tensor feeding
two consumers, residual connections) needs no special casing.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref

import torch
from torch import nn

from . import _lib, ops
from ._lib import Y3Tensor, check
from .common import SPP, Bottleneck, Concat, Conv, MaxPool2d, Upsample, ZeroPad2d
from .engine import _pad8, _sources, graph_hw
from .ops import View


class Act:
    """An activation tensor and (during backward) its gradient, both NHWC views with identical geometry."""

    def __init__(self, view: View):
        self.view = view
        self.g: View | None = None
        self.ready = False  # gradient buffer holds a value (first producer WRITES, later ones accumulate: no memset)
        # a gradient contribution that is still another tensor (the shortcut of a Bottleneck: d out / d x = identity, so the contribution IS the output's
        # gradient): the next producer adds it while it writes (`take_deferred`), anyone else sees it materialised by `grad()`
        self.deferred: View | None = None

    def slice(self, coff, c):
        a = Act(self.view.slice(coff, c))
        a.parent, a.coff = self, coff
        return a

    def defer(self, contribution: View) -> bool:
        """note `contribution` as the (only, so far) gradient of this tensor without copying it; False when that is not possible here"""
        if self.ready or self.deferred is not None or self.g is not None or getattr(self, "parent", None) is not None:
            return False
        v = self.view
        if (contribution.n, contribution.h, contribution.w, contribution.c) != (v.n, v.h, v.w, v.c):
            return False
        self.deferred = contribution
        return True

    def take_deferred(self) -> View | None:
        """the deferred contribution, handed to a producer that will WRITE grad() = its own result + this (and then mark_ready)"""
        d, self.deferred = self.deferred, None
        return d

    def grad(self) -> View:
        if self.deferred is not None:   # someone needs the buffer itself: copy the deferred contribution in
            d = self.take_deferred()
            g = self.grad()
            ops.copy_slice(d, g)
            self.ready = True
            return g
        if self.g is None:
            parent = getattr(self, "parent", None)
            if parent is not None:
                pg = parent.grad()
                if not parent.is_ready():
                    # a slice is touched before the whole buffer received its first gradient (a consumer of the slice that
                    # comes later in the graph than the Concat's consumer): fall back to zero-fill + accumulate
                    pg.buf.zero_()
                    parent.mark_ready()
                self.g = pg.slice(self.coff, self.view.c)
            else:
                v = self.view
                self.g = View(torch.empty(v.n * v.h * v.w * v.pitch, dtype=v.buf.dtype, device=v.buf.device), v.n, v.h, v.w, v.c, v.pitch, 0)
        return self.g

    def is_ready(self) -> bool:
        parent = getattr(self, "parent", None)
        return self.ready or self.deferred is not None or (parent is not None and parent.is_ready())

    def mark_ready(self):
        self.ready = True

    def drop_grad(self):
        self.g = None
        self.ready = False
        self.deferred = None


def _f32(t):
    return t.detach().float().contiguous()


class _Unit:
    def fwd(self):
        raise NotImplementedError

    def bwd(self, grads: dict):
        raise NotImplementedError


class ConvUnit(_Unit):
    """conv -> BN(batch stats) -> SiLU (+ residual)."""

    def __init__(self, plan, m: Conv, x: Act, y: Act, res: Act | None, need_dx=True, label=""):
        self.plan, self.m, self.x, self.y, self.res, self.need_dx, self.label = plan, m, x, y, res, need_dx, label
        conv = m.conv
        self.k, self.s = conv.kernel_size[0], conv.stride[0]
        self.cin, self.cout = x.view.c, _pad8(conv.out_channels)
        self.co_real, self.ci_real = conv.out_channels, conv.in_channels
        v = y.view
        dev, dt = plan.device, plan.dtype
        self.u = plan.alloc_view(v.n, v.h, v.w, self.cout)
        C_ = self.cout
        self.sums = plan.bn_sums(C_)   # shared by every unit of the plan: each pass consumes its sums before the next launch is queued (one stream)
        self.scale, self.shift, self.mean, self.invstd = (torch.empty(C_, dtype=torch.float32, device=dev) for _ in range(4))
        self.zero_bias = torch.zeros(C_, dtype=torch.float32, device=dev)
        self.act = _lib.Y3_ACT_SILU if isinstance(m.act, nn.SiLU) else _lib.Y3_ACT_NONE
        self.count = v.n * v.h * v.w
        self.use_stem = False
        self.stat_rows = None
        self.filt_d = None
        # the data gradient of this unit runs through the generic dgrad bank (not the stride-2 parity-class banks, not layer 0)
        self.pair_pack = need_dx and plan.dtype in (torch.float16, torch.bfloat16) and not (self.s == 2 and self.k == 3)
        self.bank_fwd = self.bank_dgrad = None   # persistent filter banks filled by the plan's one-launch packing (TrainPlan.pack_jobs)
        # consumer-side BatchNorm (TrainPlan._pair_bn_consumers): `bn_in` = the unit whose normalise + activation (+ shortcut) this 1x1 unit applies on the way in;
        # `act_in_consumer` = this unit's own normalise pass runs inside its 1x1 consumer's launch
        self.bn_in: ConvUnit | None = None
        self.act_in_consumer = False
        self.bnin_rows = -1
        self.stem_filt = None   # layer 0 by recomputation (stem_recompute): the stem-packed bank of this step's forward, kept for the backward

    def sync_group(self):
        """the process group of a torch.nn.SyncBatchNorm layer (reference train.py:270-272: --sync-bn converts every BatchNorm2d before DDP wraps the
        model) when there is more than one rank in it, else False: the layer then behaves like nn.BatchNorm2d, as torch's does"""
        bn = self.m.bn
        if not isinstance(bn, nn.SyncBatchNorm) or not (torch.distributed.is_available() and torch.distributed.is_initialized()):
            return False
        group = bn.process_group   # None = the default group
        return (group,) if torch.distributed.get_world_size(group) > 1 else False

    def _sync_forward_stats(self, group):
        """sums[0 .. 2C) hold this rank's (sum, sum of squares): all-reduce them together with the element count, finalize with the global count"""
        bn, c2 = self.m.bn, 2 * self.cout
        pack = torch.empty(c2 + 1, dtype=torch.float64, device=self.plan.device)
        pack[:c2].copy_(self.sums[:c2])
        pack[c2] = float(self.count)
        torch.distributed.all_reduce(pack, group=group[0])
        self.sums[:c2].copy_(pack[:c2])
        self.count_all = pack[c2:]   # kept for the backward of this step
        check(
            _lib.lib().y3_bn_finalize_devcount(self.sums.data_ptr(), self.count_all.data_ptr(), self.cout, bn.weight.data_ptr(), bn.bias.data_ptr(), float(bn.eps),
                                               float(bn.momentum), bn.running_mean.data_ptr(), bn.running_var.data_ptr(), self.scale.data_ptr(), self.shift.data_ptr(),
                                               self.mean.data_ptr(), self.invstd.data_ptr(), ops.stream_ptr()),
            "y3_bn_finalize_devcount",
        )

    def generic_dgrad(self) -> bool:
        """the data gradient runs as ONE launch of the forward conv kernels on the flipped bank (not the stride-2 parity classes)"""
        return self.need_dx and not (self.s == 2 and self.k == 3 and self.plan.dtype != torch.float32)

    def fused_stem_bwd(self) -> bool:
        """layer 0 through y3_stem_bn_bwd_wgrad (Y3_STEM_BWD=0: BatchNorm backward + generic filter gradient, for A/B runs)"""
        return (self.use_stem and self.plan.x_nchw is not None and not self.need_dx and self.res is None and self.cout == 32 and self.ci_real <= 3
                and os.environ.get("Y3_STEM_BWD", "1") != "0" and not self.sync_group())   # (SyncBatchNorm: the two-phase backward below)

    def stem_recompute(self) -> bool:
        """layer 0 without its pre-BatchNorm tensor (round 6, Y3_STEM_RECOMPUTE=1; default off): statistics-only pass, one pass that writes act(bn(u)), and a backward that
        rebuilds u from the image -- the 1.68 GB tensor (batch 64) is neither written nor read back three times, and the step needs that much less memory.  Same bits in the
        forward (tests), but NOT faster: the four passes are bound by their patch staging / elementwise VALU work, not by HBM -- 1.18 + 1.08 + 0.98 ms against 0.74 + 0.66 +
        0.69 + 0.98 ms of the stored path, whole step 55.87 -> 56.11 ms on one box (profiles/r06_stem_recompute_ab.txt).  Kept as a switch for memory-bound configurations."""
        return (self.fused_stem_bwd() and self.plan.epilogue_stats and not self.act_in_consumer and self.plan.dtype in (torch.float16, torch.bfloat16)
                and os.environ.get("Y3_STEM_RECOMPUTE", "0") == "1")

    def fwd(self):
        m, bn = self.m, self.m.bn
        L = _lib.lib()
        st = ops.stream_ptr()
        if self.cout != self.co_real:
            raise NotImplementedError("BatchNorm over a channel-padded conv")
        dcode = ops.dtype_code(self.plan.dtype)
        ut = self.u.y3()
        stats_in_epilogue = False
        sync = self.sync_group()
        if self.use_stem and self.plan.x_nchw is not None:
            # layer 0 straight from the caller's NCHW image (csrc/stem.hip); the NHWC copy is still made for the filter gradient
            filt = ops.pack_filter_stem(m.conv.weight, self.cout, self.plan.dtype)
            if self.plan.epilogue_stats:   # BatchNorm statistics from the same launch: one row per block, no reduction pass over the 64 B/pixel tensor
                xi = self.plan.x_nchw
                rows = ops.stem_conv_stats_rows(xi.shape[0], xi.shape[2], xi.shape[3])
                buf = self.plan.stat_buffer(rows * 2 * self.cout)
                recompute = self.stem_recompute()
                self.stem_filt = filt if recompute else None   # the backward multiplies with the same bank
                n_rows = ops.stem_conv_stats_only(xi, filt, self.u, buf, rows) if recompute else ops.stem_conv_stats(xi, filt, self.zero_bias, self.u, buf, rows)
                if sync:
                    check(L.y3_bn_sum_rows(buf.data_ptr(), n_rows, self.cout, self.sums.data_ptr(), st), "y3_bn_sum_rows")
                    self._sync_forward_stats(sync)
                else:
                    check(
                        L.y3_bn_finalize_rows(buf.data_ptr(), n_rows, self.count, self.cout, self.sums.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(), float(bn.eps),
                                              float(bn.momentum), bn.running_mean.data_ptr(), bn.running_var.data_ptr(), self.scale.data_ptr(), self.shift.data_ptr(),
                                              self.mean.data_ptr(), self.invstd.data_ptr(), st),
                        "y3_bn_finalize_rows",
                    )
                stats_in_epilogue = True
            else:
                ops.stem_conv(self.plan.x_nchw, filt, self.zero_bias, self.u, act=False)
        else:
            if self.plan.banks_fresh and self.bank_fwd is not None:   # packed with every other layer at the top of this forward
                filt, self.filt_d = self.bank_fwd, self.bank_dgrad
            elif self.pair_pack:   # forward and data-gradient banks in one launch; the weights do not change before the backward
                filt, self.filt_d = ops.pack_filter_pair(m.conv.weight, self.cout, self.cin, self.plan.dtype)
            else:
                filt = ops.pack_filter(m.conv.weight, self.cout, self.cin, self.plan.dtype)
            if self.bn_in is not None:
                # the producer's BatchNorm + activation (+ shortcut) applied on the way in (csrc/conv_1x1s.h IN form): its normalised output is written once by THIS launch
                pr = self.bn_in
                buf = self.plan.stat_buffer(self.bnin_rows * 2 * self.cout)
                n_rows = ops.conv1x1_bnin_stats(pr.u, pr.scale, pr.shift, pr.act, pr.res.view if pr.res is not None else None, pr.y.view, filt, self.zero_bias, self.u, buf,
                                                self.bnin_rows)
                check(
                    L.y3_bn_finalize_rows(buf.data_ptr(), n_rows, self.count, self.cout, self.sums.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(), float(bn.eps),
                                          float(bn.momentum), bn.running_mean.data_ptr(), bn.running_var.data_ptr(), self.scale.data_ptr(), self.shift.data_ptr(),
                                          self.mean.data_ptr(), self.invstd.data_ptr(), st),
                    "y3_bn_finalize_rows",
                )
                stats_in_epilogue = True
            elif self.plan.epilogue_stats:
                # BatchNorm statistics taken in the conv epilogue (per-tile rows of sum / sum of squares): no separate pass over u
                ws = self.plan.conv_ws
                if self.stat_rows is None:
                    self.stat_rows = ops.conv2d_stats_rows(self.x.view, self.u, self.k, self.s, workspace=ws)
                buf = self.plan.stat_buffer(self.stat_rows * 2 * self.cout)
                n_rows = ops.conv2d_stats(self.x.view, filt, self.zero_bias, self.u, self.k, self.s, buf, self.stat_rows, workspace=ws)
                if sync:
                    check(L.y3_bn_sum_rows(buf.data_ptr(), n_rows, self.cout, self.sums.data_ptr(), st), "y3_bn_sum_rows")
                    self._sync_forward_stats(sync)
                else:
                    check(
                        L.y3_bn_finalize_rows(buf.data_ptr(), n_rows, self.count, self.cout, self.sums.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(), float(bn.eps),
                                              float(bn.momentum), bn.running_mean.data_ptr(), bn.running_var.data_ptr(), self.scale.data_ptr(), self.shift.data_ptr(),
                                              self.mean.data_ptr(), self.invstd.data_ptr(), st),
                        "y3_bn_finalize_rows",
                    )
                stats_in_epilogue = True
            else:
                ops.conv2d(self.x.view, filt, self.zero_bias, self.u, self.k, self.s, act=False, workspace=self.plan.conv_ws)
        if not stats_in_epilogue and sync:
            check(L.y3_bn_stats(C.byref(ut), dcode, self.sums.data_ptr(), st), "y3_bn_stats")
            self._sync_forward_stats(sync)
        elif not stats_in_epilogue:
            check(
                L.y3_bn_stats_finalize(C.byref(ut), dcode, self.sums.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(), float(bn.eps), float(bn.momentum),
                                       bn.running_mean.data_ptr(), bn.running_var.data_ptr(), self.scale.data_ptr(), self.shift.data_ptr(), self.mean.data_ptr(),
                                       self.invstd.data_ptr(), st),
                "y3_bn_stats_finalize",
            )
        if self.act_in_consumer:
            return   # y = act(scale u + shift) (+ shortcut) is computed and stored by the 1x1 consumer's launch (the next unit)
        if self.stem_filt is not None:
            ops.stem_conv_bn(self.plan.x_nchw, self.stem_filt, self.scale, self.shift, self.act, self.y.view)   # u recomputed from the image, never stored
            return
        yt = self.y.view.y3()
        rt = self.res.view.y3() if self.res is not None else None
        check(L.y3_bn_act_fwd(C.byref(ut), self.scale.data_ptr(), self.shift.data_ptr(), C.byref(rt) if rt is not None else None, C.byref(yt), dcode, self.act, st),
              "y3_bn_act_fwd")

    def bwd(self, grads):
        m = self.m
        L = _lib.lib()
        st = ops.stream_ptr()
        dcode = ops.dtype_code(self.plan.dtype)
        gy = self.y.grad()
        dgamma = self.plan.grad_alloc((self.cout,))
        dbeta = self.plan.grad_alloc((self.cout,))
        if self.fused_stem_bwd():
            # layer 0: no data gradient, so du has one consumer -- the filter gradient; both in one pass over (u, dy), du never stored
            xi = self.plan.x_nchw
            if xi._version != self.plan.x_version:
                raise RuntimeError("the input batch was modified in place between the forward and the backward of this training step")
            dw = self.plan.grad_alloc(tuple(m.conv.weight.shape))
            if self.stem_filt is not None:
                ops.stem_bn_bwd_wgrad_recompute(xi, self.stem_filt, gy, self.scale, self.shift, self.mean, self.invstd, self.act, self.sums, dgamma, dbeta, dw, self.plan.stem_bwd_ws())
            else:
                ops.stem_bn_bwd_wgrad(xi, self.u, gy, self.scale, self.shift, self.mean, self.invstd, self.act, self.sums, dgamma, dbeta, dw, self.plan.stem_bwd_ws())
            grads[m.bn.weight] = dgamma   # handed over in the order the arena slices were taken: a bucket is then one contiguous range (parallel.GradBuckets)
            grads[m.bn.bias] = dbeta
            grads[m.conv.weight] = dw
            return
        du = self.plan.scratch_like(self.u)
        ut, gt, dt = self.u.y3(), gy.y3(), du.y3()
        sync = self.sync_group()
        if sync:
            # SyncBatchNorm: du needs the means of (dz, dz xhat) over EVERY rank's pixels; dgamma / dbeta stay this rank's sums (the gradient
            # exchange averages them with the other parameter gradients, as DDP does around torch's SyncBatchNorm)
            c2 = 2 * self.cout
            check(
                L.y3_bn_act_bwd_reduce(C.byref(ut), C.byref(gt), self.scale.data_ptr(), self.shift.data_ptr(), self.mean.data_ptr(), self.invstd.data_ptr(), dcode, self.act,
                                       self.sums.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), st),
                "y3_bn_act_bwd_reduce",
            )
            tot = self.sums[:c2].clone()
            torch.distributed.all_reduce(tot, group=sync[0])
            self.sums[c2 : 2 * c2].copy_(tot / self.count_all)
            grt = None
            if self.res is not None:
                gr = self.res.grad()
                grt = gr.y3()
            check(
                L.y3_bn_act_bwd_apply(C.byref(ut), C.byref(gt), self.scale.data_ptr(), self.shift.data_ptr(), self.mean.data_ptr(), self.invstd.data_ptr(), dcode, self.act,
                                      self.sums.data_ptr(), C.byref(dt), C.byref(grt) if grt is not None else None, int(self.res.is_ready()) if self.res is not None else 0, st),
                "y3_bn_act_bwd_apply",
            )
            if self.res is not None:
                self.res.mark_ready()
        elif self.res is not None and not self.res.is_ready() and os.environ.get("Y3_DEFER_SHORTCUT", "1") != "0" and self.res.defer(gy):
            # out = act(bn(conv)) + res and nothing has written d res yet: d res = d out + (what cv1's data gradient adds).  Nothing is stored here: the
            # data-gradient launch of the other consumer of res (cv1 of the Bottleneck) takes d out as its residual operand and writes d res once --
            # one pass over the tensor less per Bottleneck, 4.4 GB of the batch-64 step (Y3_DEFER_SHORTCUT=0: the stored form, for A/B runs)
            check(
                L.y3_bn_act_bwd(C.byref(ut), C.byref(gt), self.scale.data_ptr(), self.shift.data_ptr(), self.mean.data_ptr(), self.invstd.data_ptr(), dcode, self.act,
                                self.sums.data_ptr(), C.byref(dt), dgamma.data_ptr(), dbeta.data_ptr(), st),
                "y3_bn_act_bwd",
            )
        elif self.res is not None:  # out = act(bn(conv)) + res  ->  d res (+)= d out, written by the pass that reads d out anyway
            gr = self.res.grad()
            grt = gr.y3()
            check(
                L.y3_bn_act_bwd_res(C.byref(ut), C.byref(gt), self.scale.data_ptr(), self.shift.data_ptr(), self.mean.data_ptr(), self.invstd.data_ptr(), dcode, self.act,
                                    self.sums.data_ptr(), C.byref(dt), dgamma.data_ptr(), dbeta.data_ptr(), C.byref(grt), int(self.res.is_ready()), st),
                "y3_bn_act_bwd_res",
            )
            self.res.mark_ready()
        else:
            check(
                L.y3_bn_act_bwd(C.byref(ut), C.byref(gt), self.scale.data_ptr(), self.shift.data_ptr(), self.mean.data_ptr(), self.invstd.data_ptr(), dcode, self.act,
                                self.sums.data_ptr(), C.byref(dt), dgamma.data_ptr(), dbeta.data_ptr(), st),
                "y3_bn_act_bwd",
            )
        grads[m.bn.weight] = dgamma   # cout == co_real (checked in fwd): whole tensors, so autograd takes them without a copy; handed over in
        grads[m.bn.bias] = dbeta      # arena order (dgamma, dbeta, then the filter gradient below): buckets stay contiguous, disjoint ranges
        self.plan.wgrad(grads, m.conv.weight, None, self.x.view, du, self.k, self.s, self.co_real, self.ci_real)
        self._dgrad(du, grads)

    def _dgrad(self, du: View, grads=None):
        m = self.m
        if not self.need_dx:
            return
        if not self.generic_dgrad():
            gx = self.x.grad()
            ops.conv2d_dgrad_s2(m.conv.weight, du, gx, accumulate=self.x.is_ready())   # no zero-tap waste
        else:
            deferred = self.x.take_deferred()   # (a Bottleneck's shortcut gradient: added by this launch, never stored on its own)
            gx = self.x.grad()
            filt_d, self.filt_d = self.filt_d, None
            if filt_d is None:
                filt_d = ops.pack_filter_dgrad(m.conv.weight, self.cout, self.cin, self.plan.dtype)
            zb = self.plan.zeros_f32(self.cin)
            res = deferred if deferred is not None else (gx if self.x.is_ready() else None)
            ops.conv2d(du, filt_d, zb, gx, self.k, 1, act=False, residual=res, in_dilation=self.s, workspace=self.plan.conv_ws)
        self.x.mark_ready()


class HeadUnit(_Unit):
    """Detect's 1x1 Conv2d(bias) + view/permute to (bs, na, ny, nx, no) (reference models/yolo.py:96-98)."""

    def __init__(self, plan, conv: nn.Conv2d, det, x: Act, label=""):
        self.plan, self.conv, self.det, self.x, self.label = plan, conv, det, x, label
        v = x.view
        self.cout = _pad8(conv.out_channels)
        self.head = plan.alloc_view(v.n, v.h, v.w, self.cout)
        self.raw = None
        self.bank_fwd = self.bank_dgrad = None
        self.filt_d = None

    def fwd(self):
        w = self.conv.weight
        if self.plan.banks_fresh and self.bank_fwd is not None:
            filt, self.filt_d = self.bank_fwd, self.bank_dgrad
        else:
            filt, self.filt_d = ops.pack_filter(w, self.cout, self.x.view.c, self.plan.dtype), None
        bias = torch.zeros(self.cout, dtype=torch.float32, device=self.plan.device)
        bias[: self.conv.out_channels] = _f32(self.conv.bias)
        ops.conv2d(self.x.view, filt, bias, self.head, 1, 1, act=False)
        v = self.head
        det = self.det
        self.raw = torch.empty(v.n, det.na, v.h, v.w, det.no, dtype=self.plan.dtype, device=self.plan.device)
        ops.detect_decode(self.head, det.na, det.no, [0.0] * (det.na * 2), 1.0, self.raw, None, 0, 0)
        return self.raw

    def bwd_from(self, graw, grads):
        L = _lib.lib()
        det = self.det
        v = self.head
        ghead = self.plan.scratch_like(self.head)
        gt = ghead.y3()
        graw = graw.contiguous().to(self.plan.dtype)
        check(L.y3_detect_raw_bwd(graw.data_ptr(), ops.dtype_code(self.plan.dtype), v.n, det.na, v.h, v.w, det.no, C.byref(gt), ops.stream_ptr()), "y3_detect_raw_bwd")
        self.plan.wgrad(grads, self.conv.weight, self.conv.bias, self.x.view, ghead, 1, 1, self.conv.out_channels, self.conv.in_channels)
        gx = self.x.grad()
        filt_d, self.filt_d = self.filt_d, None
        if filt_d is None:
            filt_d = ops.pack_filter_dgrad(self.conv.weight, self.cout, self.x.view.c, self.plan.dtype)
        res = gx if self.x.is_ready() else None
        ops.conv2d(ghead, filt_d, self.plan.zeros_f32(self.x.view.c), gx, 1, 1, act=False, residual=res)
        self.x.mark_ready()


class UpsampleUnit(_Unit):
    def __init__(self, plan, x: Act, y: Act):
        self.plan, self.x, self.y = plan, x, y

    def fwd(self):
        ops.upsample2x(self.x.view, self.y.view)

    def bwd(self, grads):
        gy, gx = self.y.grad().y3(), self.x.grad().y3()
        check(_lib.lib().y3_upsample2x_bwd(C.byref(gy), C.byref(gx), ops.dtype_code(self.plan.dtype), int(self.x.is_ready()), ops.stream_ptr()), "y3_upsample2x_bwd")
        self.x.mark_ready()


class MaxPoolUnit(_Unit):
    def __init__(self, plan, x: Act, y: Act, k, s, p, zr, zb):
        self.plan, self.x, self.y, self.k, self.s, self.p, self.zr, self.zb = plan, x, y, k, s, p, zr, zb

    def fwd(self):
        ops.maxpool2d(self.x.view, self.y.view, self.k, self.s, self.p, self.zr, self.zb)

    def bwd(self, grads):
        # the indexed two-pass form, 16 bytes of channels per thread (round 6; y3_maxpool2d_bwd's gather moved 2 bytes per thread: 8.5 ms of yolov3-tiny's 16.4 ms step)
        ops.maxpool2d_bwd(self.x.view, self.y.grad(), self.x.grad(), self.k, self.s, self.p, self.zr, self.zb, accumulate=self.x.is_ready())
        self.x.mark_ready()


class SPPPoolUnit(_Unit):
    """the 5/9/13 stride-1 max-pools of SPP (reference models/common.py:287-290), one forward launch; the backward is
    three max-pool backward launches pairs (first-maximum index per window, then k^2 look-ups per element) accumulating into the pooled tensor's gradient."""

    def __init__(self, plan, x: Act, y3c: Act):
        self.plan, self.x, self.y = plan, x, y3c

    def fwd(self):
        ops.spp_pyramid(self.x.view, self.y.view)

    def bwd(self, grads):
        c = self.x.view.c
        for j, k in enumerate((5, 9, 13)):   # (round 6: the indexed two-pass backward -- the gather form cost 210 ms per batch-64 step on these three pools)
            ops.maxpool2d_bwd(self.x.view, self.y.grad().slice(j * c, c), self.x.grad(), k, 1, k // 2, accumulate=self.x.is_ready())
            self.x.mark_ready()


class TrainSlot:
    """What the training plans of one (model, activation dtype, device, slot index) share: ONE activation arena sized for the largest shape seen, the persistent
    filter banks (re-packed from the fp32 masters at every forward, whatever the shape), the conv workspace and the small scratch buffers.

    Why: the reference's multi-scale training (train.py:394-399) draws a new size from ~21 (imgsz 640: 320 .. 960 in steps of 32) for EVERY batch.  A plan that owned its
    activations cost 12-55 GB per shape, so only three shapes could stay compiled and every step rebuilt a plan and re-allocated its buffers.  Now a plan is a list of
    views into the slot's arena plus a few KB of per-layer vectors: two dozen shapes stay compiled (PlanCache.MAX_TRAIN_SHAPES) and a change of shape allocates nothing
    once the largest shape has been seen.  A slot runs one forward at a time: while the backward of a grad-enabled forward is outstanding the slot is busy and
    another forward (a second micro-batch, a no_grad pass) takes the next slot (PlanCache.MAX_TRAIN of them) -- run_model_train."""

    def __init__(self, dtype, device, index=0):
        self.dtype, self.device, self.index = dtype, device, index
        self.arena: torch.Tensor | None = None
        self.generation = 0          # bumped when the arena is replaced: plans built on the old one are stale
        self.arena_allocations = 0   # how often the arena was (re)allocated (tests: no growth after the largest shape)
        self.active = None           # weakref to the plan whose forward ran last in this slot
        self.last_forward = -1
        self.param_ids = None
        self.pack_jobs = None
        self._banks: dict = {}
        self._conv_ws = None
        self._bn_sums = self._stat_buf = self._stem_bwd_ws = None
        self._wgrad_ws = None        # split-K slabs of the filter gradients: one buffer for every layer of the slot (the launches that use it are ordered on one stream)
        self._zeros: dict = {}
        self._ones: dict = {}

    def busy(self) -> bool:
        pl = self.active() if self.active is not None else None
        return pl is not None and pl.outstanding

    def take_over(self, plan):
        """`plan` runs its forward in this slot now: whatever another plan saved here is gone (its backward raises instead of computing garbage)"""
        old = self.active() if self.active is not None else None
        if old is not None and old is not plan and old.outstanding:
            old.generation += 1
            old.outstanding = False
        self.active = weakref.ref(plan)

    def reset(self, param_ids):
        """the model's Parameter objects changed: banks keyed by the old tensors are dropped, every plan of the slot is stale"""
        self.param_ids = param_ids
        self.pack_jobs, self._banks = None, {}
        self.generation += 1

    def grow(self, nbytes: int):
        self.arena = None   # (release first: the old arena may be most of the free memory)
        self.arena = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        self.generation += 1
        self.arena_allocations += 1

    def carve(self, off: int, nbytes: int, dtype):
        if self.arena is None or off + nbytes > self.arena.numel():
            return None
        return self.arena[off:off + nbytes].view(dtype)

    def banks(self, w, cout, cin, want_fwd, want_dgrad):
        """persistent (forward bank, data-gradient bank) of weight tensor `w`, registered once per slot with the one-launch packer"""
        if self.pack_jobs is None:
            self.pack_jobs = ops.PackJobs(self.dtype, self.device)
        key = (id(w), cout, cin, bool(want_fwd), bool(want_dgrad))
        b = self._banks.get(key)
        if b is None:
            b = self._banks[key] = self.pack_jobs.add(w, cout, cin, want_fwd, want_dgrad)
        return b

    def conv_ws(self):
        if self._conv_ws is None:
            self._conv_ws = ops.conv_workspace(self.device)
        return self._conv_ws


_FORWARD_TICK = 0
PLAN_BUILDS = 0   # TrainPlan constructions in this process (tests of the multi-scale path count them)


class TrainPlan:
    def __init__(self, model, n, h, w, dtype, device, slot: "TrainSlot | None" = None):
        from .yolo import Detect

        global PLAN_BUILDS
        PLAN_BUILDS += 1
        self.model, self.n, self.h, self.w, self.dtype, self.device = model, n, h, w, dtype, device
        self.slot = slot if slot is not None else TrainSlot(dtype, device)   # (on its own -- tests, tools -- use TrainPlan.build(..., TrainSlot(dtype, device)): a private slot)
        self.slot_generation = self.slot.generation
        self._act_off, self._overflow, self._carved = 0, False, []   # bump allocation inside the slot's arena (alloc_view)
        self.units: list[_Unit] = []
        self.heads: list[HeadUnit] = []
        self.acts: list[Act] = []
        layers = list(model.model)
        hw = graph_hw(model, h, w)
        src = [_sources(i, m.f) for i, m in enumerate(layers)]

        def kind(m):
            return m[0] if isinstance(m, nn.Sequential) else m

        def out_ch(i, m):
            k = m[-1] if isinstance(m, nn.Sequential) else m
            if isinstance(k, Conv):
                return k.conv.out_channels
            if isinstance(k, Bottleneck):
                return k.cv2.conv.out_channels
            if isinstance(k, Concat):
                return sum(ch[j] for j in src[i])
            if isinstance(k, SPP):
                return k.cv2.conv.out_channels
            return ch[src[i][0]]

        ch = {}
        for i, m in enumerate(layers):
            if not isinstance(kind(m), Detect):
                ch[i] = out_ch(i, m)
        self._max_c = max(_pad8(mod.out_channels) for mod in model.modules() if isinstance(mod, nn.Conv2d))

        def new_act(i_hw, c):
            a = Act(self.alloc_view(n, i_hw[0], i_hw[1], c))
            self.acts.append(a)
            return a

        # placement: Concat sources live in slices of the Concat buffer (zero-copy, forward and backward)
        placed: dict[int, Act] = {}
        for i, m in enumerate(layers):
            if isinstance(kind(m), Concat):
                cat = new_act(hw[i], ch[i])
                placed[i] = cat
                off = 0
                for j in src[i]:
                    if j in placed or (ch[j] % 8) or (off % 8):
                        raise NotImplementedError("a tensor feeding two Concats / unaligned Concat offsets is not supported in training")
                    placed[j] = cat.slice(off, ch[j])
                    self.acts.append(placed[j])
                    off += ch[j]

        def home(i):
            if i not in placed:
                placed[i] = new_act(hw[i], ch[i])
            return placed[i]

        cin0 = _pad8(model.yaml.get("ch", 3))
        self.x_in = Act(self.alloc_view(n, h, w, cin0))
        out: dict[int, Act] = {-1: self.x_in}
        pad_of = {}
        for i, m in enumerate(layers):
            k = kind(m)
            ins = [out[j] for j in src[i]]
            if isinstance(k, Detect):
                self.det = k
                for lvl, a in enumerate(ins):
                    self.heads.append(HeadUnit(self, k.m[lvl], k, a, f"L{i}.m{lvl}"))
                continue
            if isinstance(k, Concat):
                out[i] = placed[i]
            elif isinstance(k, Upsample):
                out[i] = home(i)
                self.units.append(UpsampleUnit(self, ins[0], out[i]))
            elif isinstance(k, ZeroPad2d):
                pad_of[i] = (k.padding[1], k.padding[3])
                out[i] = ins[0]
            elif isinstance(k, MaxPool2d):
                zr, zb = pad_of.get(src[i][0], (0, 0))
                out[i] = home(i)
                self.units.append(MaxPoolUnit(self, ins[0], out[i], k.kernel_size, k.stride, k.padding, zr, zb))
            elif isinstance(m, nn.Sequential) or isinstance(k, Bottleneck):
                subs = list(m) if isinstance(m, nn.Sequential) else [k]
                x = ins[0]
                for r, sub in enumerate(subs):
                    last = r == len(subs) - 1
                    t = new_act(hw[i], sub.cv1.conv.out_channels)
                    self.units.append(ConvUnit(self, sub.cv1, x, t, None, True, f"L{i}.{r}.cv1"))
                    y = home(i) if last else new_act(hw[i], sub.cv2.conv.out_channels)
                    self.units.append(ConvUnit(self, sub.cv2, t, y, x if sub.add else None, True, f"L{i}.{r}.cv2"))
                    x = y
                out[i] = x
            elif isinstance(k, SPP):
                c_ = k.cv1.conv.out_channels
                cat = new_act(hw[i], 4 * c_)
                s0, s3 = cat.slice(0, c_), cat.slice(c_, 3 * c_)
                self.acts += [s0, s3]
                self.units.append(ConvUnit(self, k.cv1, ins[0], s0, None, True, f"L{i}.cv1"))
                self.units.append(SPPPoolUnit(self, s0, s3))
                out[i] = home(i)
                self.units.append(ConvUnit(self, k.cv2, cat, out[i], None, True, f"L{i}.cv2"))
            elif isinstance(k, Conv):
                out[i] = home(i)
                self.units.append(ConvUnit(self, k, ins[0], out[i], None, need_dx=src[i][0] >= 0, label=f"L{i}"))
            else:
                raise NotImplementedError(type(k).__name__)
        self._pair_bn_consumers()
        self.params = list(model.parameters())
        self.param_ids = tuple(id(p) for p in self.params)   # run_model_train rebuilds the plan when a Parameter object is replaced
        self.x_nchw = None
        self.x_version = 0

        # Filter gradients on a second HIP stream (Y3_WGRAD_STREAM=0: one stream).  Nothing downstream in the backward needs them, so the data-gradient / BatchNorm chain
        # does not wait for them.  Round 5 measured -1.0 ... -1.15 ms per batch-64 step once warm but +5 ms over the first ten steps (every launch took a fresh workspace
        # under the side stream: a second allocator pool to fill) and left it opt-in; with the slot-owned workspace (wgrad_ws) nothing is allocated under the side stream
        # and fresh processes gain from the first step: 55.87 -> 55.03 ms, same box, interleaved (profiles/r06_wgrad_stream_ab.txt).  Default since round 6.
        self.wgrad_stream = None
        self._bwd_stream = None   # the stream backward() runs on while a side stream is in use (grad_alloc records the arena on both)
        if device.type == "cuda" and os.environ.get("Y3_WGRAD_STREAM", "1") == "1":
            self.wgrad_stream = torch.cuda.Stream(device=device)   # (a lab tool may swap in a CU-masked stream: tools/lab/cu_mask.py, tools/wgrad_overlap_ab.py)
        # Y3_BN_EPILOGUE=0: statistics by a separate reduction pass over u (A/B runs); fp32 plans always take that path
        self.epilogue_stats = dtype in (torch.float16, torch.bfloat16) and os.environ.get("Y3_BN_EPILOGUE", "1") != "0"

        u0 = self.units[0] if self.units else None
        if (isinstance(u0, ConvUnit) and u0.x is self.x_in and u0.k == 3 and u0.s == 1 and u0.ci_real <= 4 and u0.cout <= 64 and u0.cout % 8 == 0
                and dtype in (torch.float16, torch.bfloat16) and os.environ.get("Y3_STEM", "1") != "0"):
            u0.use_stem = True
        self._bn_counters = [u.m.bn.num_batches_tracked for u in self.units if isinstance(u, ConvUnit) and u.m.bn.num_batches_tracked is not None]
        # scratch of the persistent conv kernel (forward + data-gradient launches of the 3x3 layers with >= 256 channels); one per plan:
        # every conv launch of the plan runs on the compute stream
        self.conv_ws = self.slot.conv_ws() if dtype in (torch.float16, torch.bfloat16) else None   # (one per slot: a slot runs one plan at a time, on one stream)
        self._arena, self._arena_off = None, 0
        self._arena_numel = sum((p.numel() + 63) // 64 * 64 for p in self.params)
        # all filter banks of a step in one launch (Y3_PACK_JOBS=0: one launch per layer, as before); the banks belong to the slot: every shape packs into the same ones
        self.pack_jobs, self.banks_fresh = None, False
        if dtype in (torch.float16, torch.bfloat16) and os.environ.get("Y3_PACK_JOBS", "1") != "0":
            for u in self.units:
                if isinstance(u, ConvUnit) and not u.use_stem:
                    u.bank_fwd, u.bank_dgrad = self.slot.banks(u.m.conv.weight, u.cout, u.cin, True, u.pair_pack)
            for hd in self.heads:
                hd.bank_fwd, hd.bank_dgrad = self.slot.banks(hd.conv.weight, hd.cout, hd.x.view.c, True, True)
            self.pack_jobs = self.slot.pack_jobs
        self.last_forward = 0
        self.generation = 0        # bumped by every forward: the saved activations belong to exactly one forward
        self.outstanding = False   # a grad-enabled forward ran and its backward has not: the saved state must not be overwritten

    def _pair_bn_consumers(self):
        """Consumer-side BatchNorm (round 5; Y3_BN_IN_CONSUMER=0: the separate passes, for A/B runs).  Where a Conv unit is IMMEDIATELY followed by a 1x1 Conv unit that reads its
        output (Bottleneck.cv2 -> the next Bottleneck's cv1, a stride-2 Conv -> the first cv1 of its stage; reference models/common.py:150-165) and the library's input-transform
        form covers the shape, the producer's normalise + activation (+ shortcut) pass is dropped: the consumer's launch reads the producer's pre-BatchNorm tensor, applies
        scale / shift / SiLU / shortcut on the way in, stores the result once for the other consumers (the next shortcut, the filter gradient) and multiplies it.  One read of
        the activation and one launch less per pair.  The backward is unchanged (it needs u and y as before)."""
        if self.dtype not in (torch.float16, torch.bfloat16) or os.environ.get("Y3_BN_IN_CONSUMER", "1") == "0" or os.environ.get("Y3_BN_EPILOGUE", "1") == "0":
            return
        for pr, cu in zip(self.units, self.units[1:]):
            if not (isinstance(pr, ConvUnit) and isinstance(cu, ConvUnit)) or cu.k != 1 or cu.s != 1 or cu.x is not pr.y or cu.use_stem or cu.res is not None:
                continue
            if getattr(pr.y, "parent", None) is not None or pr.sync_group() or cu.sync_group() or pr.act_in_consumer or pr.cout != pr.co_real:
                continue
            rows = ops.conv1x1_bnin_rows(pr.u, pr.y.view, cu.u, pr.res is not None)   # (a dry run of the library's plan: -1 = shape not covered)
            if rows <= 0:
                continue
            cu.bn_in, cu.bnin_rows, pr.act_in_consumer = pr, rows, True

    # -- helpers ---------------------------------------------------------------------------------
    @classmethod
    def build(cls, model, n, h, w, dtype, device, slot: "TrainSlot", siblings=()):
        """a plan on `slot`.  When the slot's arena is too small for this shape it is re-allocated at this shape's size and every plan of the slot (`siblings`: the
        compiled plans of the other shapes) is pointed at the new one -- same offsets, nothing is rebuilt"""
        plan = cls(model, n, h, w, dtype, device, slot)
        if plan._overflow:
            others = [p for p in siblings if p is not plan and p.slot is slot]
            for p in others:
                p.unbind()           # (the old arena is released BEFORE the new one is allocated: it may be most of the free memory)
            slot.grow(plan._act_off)
            for p in others + [plan]:
                p.rebind()
        return plan

    def alloc_view(self, n, h, w, c) -> View:
        """an NHWC activation buffer of this plan: the next 256-byte aligned range of the slot's arena.  Beyond the arena's end the view stays unbound and the plan
        only counts (`_overflow`): build() then grows the arena to the counted size and binds the views (rebind)"""
        esz = torch.empty(0, dtype=self.dtype).element_size()
        nbytes = n * h * w * c * esz
        off = self._act_off
        self._act_off = off + (nbytes + 255) // 256 * 256
        buf = self.slot.carve(off, nbytes, self.dtype)
        if buf is None:
            self._overflow = True
            buf = torch.empty(0, dtype=self.dtype, device=self.device)
        v = View(buf, n, h, w, c, c, 0)
        self._carved.append((v, off, nbytes))
        return v

    def unbind(self):
        empty = torch.empty(0, dtype=self.dtype, device=self.device)
        for v, _, _ in self._carved:
            v.buf = empty
        for a in self.acts:
            if getattr(a, "parent", None) is not None:
                a.view.buf = empty

    def rebind(self):
        """point every activation view at the slot's CURRENT arena (same offsets): after the arena grew for a larger shape.  Nothing caches a device pointer of an
        activation between launches (the y3_tensor descriptors are made per call), so this is all there is to moving a plan"""
        for v, off, nbytes in self._carved:
            v.buf = self.slot.carve(off, nbytes, self.dtype)
            assert v.buf is not None, "the arena is smaller than a plan that was built on it"
        for a in self.acts:   # channel slices of a Concat buffer share their parent's storage
            if getattr(a, "parent", None) is not None:
                a.view.buf = a.parent.view.buf
        self._overflow = False
        self.slot_generation = self.slot.generation

    def bn_sums(self, c):
        """fp64 scratch of the BatchNorm reductions (totals + per-block partial rows), one buffer for the whole slot (every pass consumes its sums before the next
        launch is queued: one stream)."""
        sl = self.slot
        if sl._bn_sums is None or sl._bn_sums.numel() < (1 + ops.BN_PARTIAL_ROWS) * 2 * max(c, self._max_c):
            sl._bn_sums = ops.bn_scratch(max(c, self._max_c), self.device)
        return sl._bn_sums

    def grad_alloc(self, shape):
        """fp32 gradient tensor of `shape`: a slice of this backward's flat arena, handed out in the order the backward produces
        gradients.  Consecutive gradients are therefore adjacent in memory and parallel.GradBuckets all-reduces whole ranges of the
        arena in place -- no flatten / copy-back passes over the 248 MB.  The arena is a fresh allocation per backward (no memset;
        torch's caching allocator recycles it), so gradients handed to autograd never alias a later backward's."""
        n = 1
        for d_ in shape:
            n *= int(d_)
        n_al = (n + 63) // 64 * 64   # 256-byte aligned slices
        if self._arena is None or self._arena_off + n_al > self._arena.numel():
            self._arena = torch.empty(max(self._arena_numel, n_al), dtype=torch.float32, device=self.device)
            self._arena_off = 0
            if self.wgrad_stream is not None:   # written on the side stream, read on the compute stream, whichever pool it came from
                self._arena.record_stream(self.wgrad_stream)
                self._arena.record_stream(self._bwd_stream if self._bwd_stream is not None else torch.cuda.current_stream())
        t = self._arena[self._arena_off:self._arena_off + n].view(shape)
        self._arena_off += n_al
        return t

    def wgrad(self, grads, w_param, b_param, x: View, du: View, k, s, co_real, ci_real):
        """Filter (and bias) gradient of one layer.  Nothing downstream in the backward needs it, so it CAN go to a second HIP
        stream (Y3_WGRAD_STREAM=1; see __init__ for the measurement).  `du` was produced on the current stream: the side
        stream waits for an event recorded here."""
        side = self.wgrad_stream
        ws = self.wgrad_ws(ops.conv2d_wgrad_workspace_bytes(x, du.c, k, s))   # (taken on the compute stream: nothing is ever allocated under the side stream)
        if side is None:
            dw, db = ops.conv2d_wgrad(x, du, k, s, co_real, ci_real, want_bias=b_param is not None, alloc=self.grad_alloc, workspace=ws)
            grads[w_param] = dw
            if b_param is not None:
                grads[b_param] = db
            return
        cur = torch.cuda.current_stream()
        if self._arena is None:
            self.grad_alloc((0,))     # the backward's gradient arena comes from the compute stream's pool too
        ev = torch.cuda.Event()
        ev.record(cur)
        du.buf.record_stream(side)   # scratch of this layer: keep it from being recycled while the side stream still reads it
        with torch.cuda.stream(side):
            side.wait_event(ev)
            dw, db = ops.conv2d_wgrad(x, du, k, s, co_real, ci_real, want_bias=b_param is not None, alloc=self.grad_alloc, workspace=ws)   # arena slices in host order, as on one stream
            dw.record_stream(cur)
            grads[w_param] = dw      # handed over under the side stream: a gradient sink that launches collectives waits for the right stream
            if b_param is not None:
                db.record_stream(cur)
                grads[b_param] = db

    def wgrad_ws(self, nbytes):
        """the slot's filter-gradient workspace, grown to the largest layer's need (a step allocates nothing for the slabs; round 5 took a fresh buffer per launch --
        on the side stream that filled a second allocator pool over the first ten steps)"""
        sl = self.slot
        if sl._wgrad_ws is None or sl._wgrad_ws.numel() < nbytes:
            old = sl._wgrad_ws
            if old is not None and self.wgrad_stream is not None:
                old.record_stream(self.wgrad_stream)   # (launches of an earlier backward may still be using it there)
            sl._wgrad_ws = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        return sl._wgrad_ws

    def stem_bwd_ws(self):
        sl = self.slot
        if sl._stem_bwd_ws is None:
            sl._stem_bwd_ws = ops.stem_bwd_workspace(self.device)
        return sl._stem_bwd_ws

    def stat_buffer(self, n_floats):
        """fp32 scratch for the conv epilogue's statistics rows, shared by all units of the slot (stream-ordered reuse)."""
        sl = self.slot
        if sl._stat_buf is None or sl._stat_buf.numel() < n_floats:
            sl._stat_buf = torch.empty(n_floats, dtype=torch.float32, device=self.device)
        return sl._stat_buf

    def zeros_f32(self, c):
        t = self.slot._zeros.get(c)
        if t is None:
            t = self.slot._zeros[c] = torch.zeros(c, dtype=torch.float32, device=self.device)
        return t

    def ones_f32(self, c):
        t = self.slot._ones.get(c)
        if t is None:
            t = self.slot._ones[c] = torch.ones(c, dtype=torch.float32, device=self.device)
        return t

    def scratch_like(self, v: View) -> View:
        return View(torch.empty(v.n * v.h * v.w * v.c, dtype=self.dtype, device=self.device), v.n, v.h, v.w, v.c, v.c, 0)

    def add_into(self, src: View, dst_act: "Act"):
        """grad(dst) += src (or = src when nothing has been written yet) through the scale/shift kernel (scale 1, shift 0)."""
        dst = dst_act.grad()
        st, dt = src.y3(), dst.y3()
        check(
            _lib.lib().y3_bn_act_fwd(C.byref(st), self.ones_f32(src.c).data_ptr(), self.zeros_f32(src.c).data_ptr(), C.byref(dt) if dst_act.is_ready() else None, C.byref(dt),
                                     ops.dtype_code(self.dtype), _lib.Y3_ACT_NONE, ops.stream_ptr()),
            "add_into",
        )
        dst_act.mark_ready()

    # -- execution -------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor):
        global _FORWARD_TICK
        _FORWARD_TICK += 1
        if self._overflow or self.slot_generation != self.slot.generation:
            # a plan constructed directly (no TrainPlan.build: its views only counted bytes) or left behind when its slot's arena was re-allocated without it
            raise RuntimeError("TrainPlan.forward: the plan is not bound to its slot's activation arena -- build plans with TrainPlan.build(model, n, h, w, dtype, device, slot)")
        self.last_forward = self.slot.last_forward = _FORWARD_TICK   # recency across slots (the slot choice in run_model_train)
        self.generation += 1
        self.x_nchw = x if x.dtype in (torch.float32, torch.float16, torch.bfloat16, torch.uint8) else None
        self.x_version = x._version
        u0 = self.units[0] if self.units else None
        if not (isinstance(u0, ConvUnit) and u0.fused_stem_bwd() and self.epilogue_stats):
            ops.nchw_to_nhwc(x, self.x_in.view, 1.0)   # layer 0 through the generic kernels (or its generic filter gradient) reads the NHWC copy
        with torch.no_grad():
            try:
                if self.pack_jobs is not None:
                    self.pack_jobs.run()
                    self.banks_fresh = True
                for u in self.units:
                    u.fwd()
                if self._bn_counters:   # nn.BatchNorm2d.num_batches_tracked += 1, one launch for all 72 counters
                    torch._foreach_add_(self._bn_counters, 1)
                return [hd.fwd() for hd in self.heads]
            finally:
                self.banks_fresh = False   # a unit driven outside TrainPlan.forward (tools) packs for itself

    def backward(self, graws):
        sync = getattr(self.model, "grad_sync", None)  # parallel.GradBuckets: overlapped gradient all-reduce
        grads = _GradSink(sync, sum(p.numel() * 4 for p in self.params))
        self._arena, self._arena_off = None, 0          # a new arena per backward (see grad_alloc)
        self._bwd_stream = torch.cuda.current_stream() if self.wgrad_stream is not None else None
        for a in self.acts:
            a.drop_grad()
        with torch.no_grad():
            for hd, g in zip(self.heads, graws):
                if g is None:
                    g = torch.zeros_like(hd.raw)
                hd.bwd_from(g, grads)
            for u in reversed(self.units):
                u.bwd(grads)
            if self.wgrad_stream is not None:
                torch.cuda.current_stream().wait_stream(self.wgrad_stream)   # every filter gradient has landed
        for a in self.acts:
            a.drop_grad()
        grads = grads.result()
        out = []
        for p in self.params:
            g = grads.get(p)
            out.append(None if g is None else g.to(p.dtype).reshape(p.shape))
        return out


class _GradSink(dict):
    """dict of finished parameter gradients; with a GradBuckets object every gradient is handed to the overlapped
    all-reduce the moment its kernels have been issued (reverse layer order = the order backward produces them)."""

    TAIL_BYTES = 16 << 20   # what may still be pending when the backward ends (the last bucket's collective is the exposed one)

    def __init__(self, sync=None, total_bytes=0):
        super().__init__()
        self.sync = sync
        self.total_bytes, self.seen_bytes, self.tail_flushed = total_bytes, 0, False

    def __setitem__(self, key, value):
        if self.sync is not None:
            self.sync.add(key, value)
            # the gradients of the last layers of the backward (= the first layers of the net: a few MB of parameters behind the longest,
            # HBM-bound kernels of the step) would otherwise wait in a half-filled bucket with ~47 MB of earlier ones until finish():
            # send what has accumulated once only TAIL_BYTES of parameters are still to come
            self.seen_bytes += value.numel() * value.element_size()
            if not self.tail_flushed and self.total_bytes and self.seen_bytes >= self.total_bytes - self.TAIL_BYTES:
                self.tail_flushed = True
                self.sync.flush()
        else:
            super().__setitem__(key, value)

    def result(self):
        return self.sync.finish() if self.sync is not None else self


class _SlotToken:
    """lives in the autograd context of one training forward; its destruction (backward done and graph freed, or graph dropped without a
    backward) releases the plan slot that forward occupied, unless a later forward has taken the plan over since"""

    def __init__(self, plan, generation):
        self.plan, self.generation = weakref.ref(plan), generation

    def __del__(self):
        pl = self.plan()
        if pl is not None and pl.generation == self.generation:
            pl.outstanding = False


class _TrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plan, x, *params):
        ctx.plan = plan
        raws = plan.forward(x)
        ctx.generation = plan.generation
        # a forward whose graph is dropped without a backward (an exception in the loss, a dry run under grad mode) must not pin the plan's
        # slot forever: the token dies with the autograd node and marks the plan idle again (round-2 advisor finding)
        ctx._slot_token = _SlotToken(plan, plan.generation)
        raws = [r.detach() for r in raws]   # fresh tensor objects: the plan keeps its own (HeadUnit.raw) and must not hold the autograd node through them
        return tuple(raws)

    @staticmethod
    def backward(ctx, *graws):
        plan = ctx.plan
        if ctx.generation != plan.generation:
            # torch autograd would have kept this forward's saved tensors alive; the static plan keeps ONE set per plan
            raise RuntimeError(
                "yolov3_amd: backward of a training forward whose saved activations were overwritten by a later forward of the same "
                f"model (forward #{ctx.generation}, plan is at #{plan.generation}). More than Y3_MAX_TRAIN_PLANS (default 2) forwards "
                "were outstanding at once; call backward() before the next forward, or raise the limit.")
        grads = plan.backward(graws)
        plan.outstanding = False
        return (None, None, *grads)


def run_model_train(model, x: torch.Tensor):
    """DetectionModel.forward in training mode: list of raw (bs, na, ny, nx, no) tensors attached to autograd."""
    ops.require_gpu(x, "DetectionModel.forward")
    p0 = next(model.parameters())
    if p0.dtype != torch.float32:
        raise TypeError("training keeps fp32 master parameters (use torch.autocast for fp16/bf16 activations), as the reference does (train.py:402)")
    dtype = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else p0.dtype
    if dtype not in (torch.float16, torch.bfloat16, torch.float32):
        raise TypeError(f"unsupported training activation dtype {dtype}")
    n, c, h, w = x.shape
    from .engine import plan_cache

    pc = plan_cache(model)
    grad = torch.is_grad_enabled() and any(p.requires_grad for p in model.parameters())
    with pc.lock:
        # A slot (TrainSlot: activation arena + filter banks, shared by every shape) runs one forward at a time.  A forward whose backward is still outstanding keeps
        # its slot busy (two micro-batches whose losses are summed, a no_grad pass between forward and backward): take the first idle slot; when every slot is busy
        # the least recently used one is taken over and the stale backward raises (see _TrainFn.backward)
        slots = pc.train_slots(dtype, x.device, TrainSlot)
        slot = next((sl for sl in slots if not sl.busy()), None)
        if slot is None:
            slot = min(slots, key=lambda sl: sl.last_forward)
        ids = tuple(id(p) for p in model.parameters())
        if slot.param_ids != ids:
            # a Parameter OBJECT was replaced since the slot captured them (re-created head, pruning, `m.conv.weight = nn.Parameter(..)`): a plan's backward would
            # return None for it and the banks would be packed from the old tensor -- start over (in-place updates keep the ids: banks are re-packed per forward)
            had = slot.param_ids is not None
            slot.reset(ids)
            if had:
                pc.drop_train_slot(slot)
        key = ("train", n, h, w, dtype, x.device.index, slot.index)
        plan = pc.get(key)
        if plan is not None and (plan.slot_generation != slot.generation or plan.param_ids != ids):
            del pc.plans[key]   # built on an arena that has been replaced since / for other Parameter objects
            plan = None
        if plan is None:
            plan = TrainPlan.build(model, n, h, w, dtype, x.device, slot, siblings=[p for k, p in pc.plans.items() if k[0] == "train"])   # (a larger shape grows the arena
            pc.put(key, plan)                                                                                                      #  and re-points the slot's other plans)
        slot.take_over(plan)
        plan.outstanding = grad
    if not grad:
        with torch.no_grad():
            return plan.forward(x)   # nothing to save for: no autograd node, the plan stays idle
    return list(_TrainFn.apply(plan, x, *plan.params))
