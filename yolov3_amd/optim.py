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
    def step(self, grad_scale: float = 1.0, max_norm: float = 0.0, ema: ModelEMA | None = None):
        """One fused update.  grad_scale: the loss scale the gradients still carry (GradScaler); max_norm: clip_grad_norm_
        threshold (0 = off; the reference uses 10.0); ema: ModelEMA to update in the same pass."""
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
        _lib.check(
            L.y3_sgd_step(tab.data_ptr(), len(recs), n_chunks, 1.0 / float(grad_scale), float(max_norm), float(self.momentum), int(self.nesterov), int(self._steps == 0),
                          float(d), scratch.data_ptr(), found.data_ptr(), ops.stream_ptr()),
            "y3_sgd_step",
        )
        if ema is not None:
            ema.update_buffers(d)
        self._steps += 1
        self.last_norm, self.found_inf = scratch[0:1], found  # device tensors; reading them is the caller's (optional) sync
