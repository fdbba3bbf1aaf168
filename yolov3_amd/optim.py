"""Fused SGD(Nesterov) + GradScaler unscale/inf-check + clip_grad_norm_ + ModelEMA on MI355X (csrc/optim.hip).

Mirror of the reference's optimizer step, train.py:414-422:
    scaler.unscale_(optimizer); clip_grad_norm_(model.parameters(), max_norm=10.0); scaler.step(optimizer); ema.update(model)
with the parameter groups of utils/torch_utils.py:207-237 (`smart_optimizer`: biases / BN weights without decay, other
weights with decay, SGD momentum + nesterov).  `param_groups` keeps torch's layout so LR schedulers that write
`group["lr"]` keep working.  No host synchronisation: a step whose gradients contain inf/nan is skipped on the device.
"""
from __future__ import annotations

import math
import struct

import torch
from torch import nn

from . import _lib, ops

CHUNK = 16384


def smart_param_groups(model: nn.Module, lr: float, weight_decay: float):
    """Three groups as reference utils/torch_utils.py:207-237: [weights (decay), norm weights (no decay), biases (no decay)]."""
    bn = tuple(v for k, v in nn.__dict__.items() if "Norm" in k)
    g = [], [], []
    for m in model.modules():
        for name, p in m.named_parameters(recurse=False):
            if name == "bias":
                g[2].append(p)
            elif name == "weight" and isinstance(m, bn):
                g[1].append(p)
            else:
                g[0].append(p)
    return [
        {"params": g[2], "lr": lr, "weight_decay": 0.0},
        {"params": g[0], "lr": lr, "weight_decay": weight_decay},
        {"params": g[1], "lr": lr, "weight_decay": 0.0},
    ]


class ModelEMA:
    """Exponential moving average of the parameters (upstream ultralytics ModelEMA; reference train.py:252,421):
    d = decay * (1 - exp(-updates / tau)); ema = d * ema + (1 - d) * p.  Buffers (BN running stats) are copied like the
    float entries of the state dict are lerped upstream -- here they are lerped by a torch foreach on the (small) buffers."""

    def __init__(self, model: nn.Module, decay=0.9999, tau=2000, updates=0):
        self.model = model
        self.shadow = {p: p.detach().clone() for p in model.parameters()}
        self.buffers = {b: b.detach().clone() for b in model.buffers() if b.dtype.is_floating_point}
        self.decay, self.tau, self.updates = decay, tau, updates

    def next_decay(self) -> float:
        self.updates += 1
        return self.decay * (1 - math.exp(-self.updates / self.tau))

    def update_buffers(self, d: float):
        if self.buffers:
            src = list(self.buffers.keys())
            dst = list(self.buffers.values())
            torch._foreach_mul_(dst, d)
            torch._foreach_add_(dst, src, alpha=1.0 - d)


class FusedSGD:
    def __init__(self, params, lr=0.01, momentum=0.937, nesterov=True, weight_decay=0.0):
        groups = list(params)
        if groups and not isinstance(groups[0], dict):
            groups = [{"params": groups}]
        self.param_groups = []
        for g in groups:
            g = dict(g)
            g.setdefault("lr", lr)
            g.setdefault("weight_decay", weight_decay)
            g["params"] = [p for p in g["params"] if p.requires_grad]
            self.param_groups.append(g)
        self.momentum, self.nesterov = momentum, nesterov
        self.state: dict = {}
        self._steps = 0
        self._dev_bufs = None
        self.last_norm = None

    def zero_grad(self, set_to_none=True):
        for g in self.param_groups:
            for p in g["params"]:
                p.grad = None if set_to_none else (p.grad.zero_() if p.grad is not None else None)

    @torch.no_grad()
    def step(self, grad_scale=1.0, max_norm: float = 0.0, ema: ModelEMA | None = None):
        """One fused update.  grad_scale: the loss scale the gradients still carry -- a Python float, or a 1-element DEVICE fp32
        tensor (GradScaler below: dynamic scale, read by the kernels); max_norm: clip_grad_norm_ threshold (0 = off; the
        reference uses 10.0); ema: ModelEMA to update in the same pass."""
        recs, n_chunks = [], 0
        dev = None
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is None:
                    continue
                ops.require_gpu(p, "FusedSGD.step")
                if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or not p.is_contiguous():
                    raise TypeError("FusedSGD expects contiguous fp32 master parameters and fp32 gradients")
                dev = p.device
                buf = self.state.get(p)
                if buf is None:
                    buf = self.state[p] = torch.zeros_like(p)
                grad = p.grad.contiguous()
                e = ema.shadow[p] if ema is not None else None
                recs.append((p.data_ptr(), grad.data_ptr(), buf.data_ptr(), e.data_ptr() if e is not None else 0, p.numel(), float(g["lr"]), float(g["weight_decay"]), n_chunks, grad))
                n_chunks += (p.numel() + CHUNK - 1) // CHUNK
        if not recs:
            return
        L = _lib.lib()
        assert L.y3_sgd_tensor_record_bytes() == 56
        table = b"".join(struct.pack("<QQQQqffii", a, b, c, d, n, lr, wd, fc, 0) for a, b, c, d, n, lr, wd, fc, _ in recs)
        host = torch.frombuffer(bytearray(table), dtype=torch.uint8)
        if self._dev_bufs is None or self._dev_bufs[0].numel() < host.numel() or self._dev_bufs[1].numel() < n_chunks + 2:
            self._dev_bufs = (torch.empty(host.numel(), dtype=torch.uint8, device=dev), torch.empty(n_chunks + 2, dtype=torch.float32, device=dev),
                              torch.zeros(1, dtype=torch.int32, device=dev))
        tab, scratch, found = self._dev_bufs
        tab[: host.numel()].copy_(host, non_blocking=True)
        d = ema.next_decay() if ema is not None else 0.0
        if isinstance(grad_scale, torch.Tensor):
            if grad_scale.dtype != torch.float32 or grad_scale.numel() != 1 or grad_scale.device != dev:
                raise TypeError("a dynamic loss scale must be a 1-element fp32 tensor on the parameters' device")
            _lib.check(
                L.y3_sgd_step_dynamic(tab.data_ptr(), len(recs), n_chunks, grad_scale.data_ptr(), float(max_norm), float(self.momentum), int(self.nesterov),
                                      int(self._steps == 0), float(d), scratch.data_ptr(), found.data_ptr(), ops.stream_ptr()),
                "y3_sgd_step_dynamic",
            )
        else:
            _lib.check(
                L.y3_sgd_step(tab.data_ptr(), len(recs), n_chunks, 1.0 / float(grad_scale), float(max_norm), float(self.momentum), int(self.nesterov), int(self._steps == 0),
                              float(d), scratch.data_ptr(), found.data_ptr(), ops.stream_ptr()),
                "y3_sgd_step",
            )
        if ema is not None:
            ema.update_buffers(d)
        self._steps += 1
        self.last_norm, self.found_inf = scratch[0:1], found  # device tensors; reading them is the caller's (optional) sync


class GradScaler:
    """torch.cuda.amp.GradScaler for the fused optimizer (reference train.py:345 `scaler = torch.cuda.amp.GradScaler(enabled=amp)`,
    :411 `scaler.scale(loss).backward()`, :414-418 `unscale_ / clip_grad_norm_ / step / update`).  The scale, the growth counter and
    the found-inf flag live on the device: `step` hands the scale tensor to the fused kernels (unscale + inf check + clip + SGD +
    EMA in one pass), `update` is one tiny launch -- the loop never synchronises with the host.  Defaults are torch's."""

    def __init__(self, init_scale=2.0**16, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, enabled=True, device=None):
        self.enabled = enabled
        self.growth_factor, self.backoff_factor, self.growth_interval = float(growth_factor), float(backoff_factor), int(growth_interval)
        self._init_scale, self._device = float(init_scale), device
        self._scale = self._tracker = None
        self._found = None

    def _lazy(self, device):
        if self._scale is None:
            self._scale = torch.full((1,), self._init_scale, dtype=torch.float32, device=device)
            self._tracker = torch.zeros(1, dtype=torch.int32, device=device)

    def scale(self, loss: torch.Tensor) -> torch.Tensor:
        if not self.enabled:
            return loss
        self._lazy(loss.device)
        return loss * self._scale.to(loss.dtype)

    def unscale_(self, optimizer):
        """no separate pass: FusedSGD.step unscales, checks for inf/nan and clips in the same kernels (kept for API parity)"""

    def step(self, optimizer: FusedSGD, max_norm: float = 0.0, ema: ModelEMA | None = None):
        if not self.enabled:
            return optimizer.step(1.0, max_norm, ema)
        p0 = next(p for g in optimizer.param_groups for p in g["params"])
        self._lazy(p0.device)
        optimizer.step(self._scale, max_norm, ema)
        self._found = optimizer.found_inf

    def update(self):
        if not self.enabled or self._found is None:
            return
        _lib.check(_lib.lib().y3_loss_scale_update(self._scale.data_ptr(), self._tracker.data_ptr(), self._found.data_ptr(), self.growth_factor, self.backoff_factor,
                                                   self.growth_interval, ops.stream_ptr()), "y3_loss_scale_update")
        self._found = None

    def get_scale(self) -> float:
        """host read (synchronises): for logging / checkpoints only"""
        return float(self._scale.item()) if self._scale is not None else self._init_scale

    def state_dict(self):
        return {"scale": self.get_scale(), "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor, "growth_interval": self.growth_interval,
                "_growth_tracker": int(self._tracker.item()) if self._tracker is not None else 0}

    def load_state_dict(self, sd):
        self._init_scale = float(sd["scale"])
        self.growth_factor, self.backoff_factor, self.growth_interval = float(sd["growth_factor"]), float(sd["backoff_factor"]), int(sd["growth_interval"])
        if self._scale is not None:
            self._scale.fill_(self._init_scale)
            self._tracker.fill_(int(sd.get("_growth_tracker", 0)))
