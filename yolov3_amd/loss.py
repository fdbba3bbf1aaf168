"""Host-side mirror of reference utils/loss.py ``ComputeLoss`` (:98-244): same constructor (model, autobalance) and
call signature ``compute_loss(p, targets) -> (loss[1] with grad, loss_items[3] detached)``.  Target building, CIoU,
objectness / class BCE and their gradients run as HIP kernels (csrc/loss.hip) behind a torch.autograd.Function;
nothing is computed with PyTorch ops and CPU tensors are rejected."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib, ops
from ._lib import Y3LossParams, check


def smooth_bce(eps=0.1):
    """(positive, negative) BCE targets (upstream ultralytics.utils.metrics.smooth_bce; reference utils/loss.py:114)."""
    return 1.0 - 0.5 * eps, 0.5 * eps


def de_parallel(model):
    """Unwrap DP/DDP (reference utils/torch_utils.py:182)."""
    return model.module if type(model) in (torch.nn.parallel.DataParallel, torch.nn.parallel.DistributedDataParallel) else model


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, crit, targets, *preds):
        L = _lib.lib()
        dev = preds[0].device
        dtype = preds[0].dtype
        preds = tuple(t.contiguous() for t in preds)
        tg = targets.detach().to(dev, torch.float32).contiguous()
        nt = tg.shape[0]
        P = crit._params(preds)
        need = int(L.y3_loss_workspace_bytes(C.byref(P), nt))
        if need == 0:
            check(-1, "y3_loss_workspace_bytes")
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        out4 = torch.empty(4, dtype=torch.float32, device=dev)
        ptrs = (C.c_void_p * len(preds))(*[t.data_ptr() for t in preds])
        check(L.y3_loss_fwd(C.byref(P), ops.dtype_code(dtype), ptrs, tg.data_ptr() if nt else None, nt, out4.data_ptr(), ws.data_ptr(), need, ops.stream_ptr()), "y3_loss_fwd")
        ctx.crit, ctx.P, ctx.ws, ctx.tg, ctx.need = crit, P, ws, tg, need
        if crit.autobalance:   # ComputeLoss(autobalance=True) reads the per-level objectness losses out of the workspace right after this call and drops it
            crit._last_fwd = (P, ops.dtype_code(dtype), nt, ws, need)
        ctx.save_for_backward(*preds)
        ctx.mark_non_differentiable(out4)
        return out4[0:1].clone(), out4

    @staticmethod
    def backward(ctx, grad_loss, _grad_items):
        L = _lib.lib()
        preds = ctx.saved_tensors
        dtype = preds[0].dtype
        grads = [torch.empty_like(t) for t in preds]
        go = grad_loss.detach().to(torch.float32).contiguous().view(-1)
        ptrs = (C.c_void_p * len(preds))(*[t.data_ptr() for t in preds])
        gptrs = (C.c_void_p * len(preds))(*[t.data_ptr() for t in grads])
        nt = ctx.tg.shape[0]
        check(
            L.y3_loss_bwd(C.byref(ctx.P), ops.dtype_code(dtype), ptrs, ctx.tg.data_ptr() if nt else None, nt, go.data_ptr(), gptrs, ctx.ws.data_ptr(), ctx.need, ops.stream_ptr()),
            "y3_loss_bwd",
        )
        return (None, None, *grads)


class ComputeLoss:
    """reference utils/loss.py:98-181."""

    sort_obj_iou = False

    def __init__(self, model, autobalance=False):
        self.device = next(model.parameters()).device
        h = model.hyp
        self.cp, self.cn = smooth_bce(eps=h.get("label_smoothing", 0.0))
        m = de_parallel(model).model[-1]
        self.balance = {3: [4.0, 1.0, 0.4]}.get(m.nl, [4.0, 1.0, 0.25, 0.06, 0.02])
        self.ssi = [int(v) for v in m.stride.tolist()].index(16) if autobalance else 0   # stride-16 level (utils/loss.py:121)
        self.gr, self.hyp, self.autobalance = 1.0, h, autobalance
        self._last_fwd = None
        self.na, self.nc, self.nl = m.na, m.nc, m.nl
        self.anchors = m.anchors
        self._anchors_host = None

    def _params(self, preds) -> Y3LossParams:
        ver = (id(self.anchors), self.anchors._version)   # autoanchor rewrites m.anchors[:] in place (utils/autoanchor.py): identity alone would go stale
        if self._anchors_host is None or self._anchors_host[0] != ver:
            self._anchors_host = (ver, self.anchors.detach().float().cpu().reshape(-1).tolist())
        h = self.hyp
        P = Y3LossParams()
        P.nl, P.na, P.nc, P.bs = self.nl, self.na, self.nc, preds[0].shape[0]
        for i, t in enumerate(preds):
            bs, na, ny, nx, no = t.shape
            if na != self.na or no != self.nc + 5 or bs != P.bs:
                raise ValueError(f"prediction level {i} has shape {tuple(t.shape)}, expected (bs, {self.na}, ny, nx, {self.nc + 5})")
            P.ny[i], P.nx[i] = ny, nx
            P.balance[i] = self.balance[i]
        for k, v in enumerate(self._anchors_host[1]):
            P.anchors[k] = v
        P.anchor_t = float(h["anchor_t"])
        P.box_gain, P.obj_gain, P.cls_gain = float(h["box"]), float(h["obj"]), float(h["cls"])
        P.cls_pw, P.obj_pw = float(h["cls_pw"]), float(h["obj_pw"])
        P.cp, P.cn = float(self.cp), float(self.cn)
        P.fl_gamma = float(h.get("fl_gamma", 0.0))
        P.sort_obj_iou = int(bool(self.sort_obj_iou))   # (utils/loss.py:101: a class attribute the caller may set on the instance)
        return P

    def __call__(self, p, targets):
        if len(p) != self.nl:
            raise ValueError(f"expected {self.nl} prediction levels, got {len(p)}")
        for t in p:
            ops.require_gpu(t, "ComputeLoss")
        loss, out4 = _LossFn.apply(self, targets, *p)
        if self.autobalance:
            # utils/loss.py:171-175: every level's weight moves towards 1 / its objectness loss (after this call's loss was formed with the old
            # weights), then all are normalised by the stride-16 level's.  The reference pays one host sync per level (`obji.item()`); here one
            # read-back of nl floats per call
            (P, dcode, nt, ws, need), self._last_fwd = self._last_fwd, None   # (the criterion does not pin the workspace beyond this read)
            obj = torch.empty(self.nl, dtype=torch.float32, device=ws.device)
            check(_lib.lib().y3_loss_level_obj(C.byref(P), dcode, nt, ws.data_ptr(), need, obj.data_ptr(), ops.stream_ptr()), "y3_loss_level_obj")
            for i, oi in enumerate(obj.tolist()):
                self.balance[i] = self.balance[i] * 0.9999 + 0.0001 / oi
            self.balance = [x / self.balance[self.ssi] for x in self.balance]
        return loss, out4[1:4].detach()
