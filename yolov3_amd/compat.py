"""Checkpoint ingest for reference ``.pt`` files.

The reference saves PICKLED ``nn.Module`` objects (train.py:470-488: ``{"model": deepcopy(model).half(), "ema": ...}``)
and loads them with ``torch.load`` (models/experimental.py:88-136 ``attempt_load``), which resolves classes by their
qualified names ``models.yolo.DetectionModel``, ``models.yolo.Detect``, ``models.common.Conv`` ...  ``install_aliases()``
registers lightweight ``models`` / ``models.yolo`` / ``models.common`` / ``models.experimental`` modules that point those
names at the yolov3_amd classes, so a reference checkpoint unpickles straight into MI355X-backed modules (the attribute
names -- ``conv/bn/act``, ``cv1/cv2/add``, ``m/anchors/stride`` -- are the same).  torch's own ``nn.Upsample`` /
``nn.MaxPool2d`` / ``nn.ZeroPad2d`` instances inside the pickle are swapped for the parameter-free stand-ins the
engine plans with.  If a real reference checkout is importable, its modules are left alone.
"""
from __future__ import annotations

import sys
import types

import torch
from torch import nn

from . import common, yolo


def install_aliases(force: bool = False) -> bool:
    """Make ``models.yolo`` / ``models.common`` importable names for unpickling.  Returns True if installed."""
    if not force and "models.yolo" in sys.modules and not getattr(sys.modules["models.yolo"], "_yolov3_amd_alias", False):
        return False  # a real reference tree is loaded: do not shadow it
    pkg = types.ModuleType("models")
    pkg.__path__ = []
    m_yolo = types.ModuleType("models.yolo")
    m_common = types.ModuleType("models.common")
    m_exp = types.ModuleType("models.experimental")
    for name in ("Detect", "DetectionModel", "Model", "BaseModel", "parse_model"):
        setattr(m_yolo, name, getattr(yolo, name))
    for name in ("Conv", "Bottleneck", "SPP", "Concat", "autopad"):
        setattr(m_common, name, getattr(common, name))
    m_exp.attempt_load = attempt_load
    for m in (pkg, m_yolo, m_common, m_exp):
        m._yolov3_amd_alias = True
    pkg.yolo, pkg.common, pkg.experimental = m_yolo, m_common, m_exp
    sys.modules.update({"models": pkg, "models.yolo": m_yolo, "models.common": m_common, "models.experimental": m_exp})
    return True


def uninstall_aliases():
    """remove the alias modules again (a process that later imports a real reference checkout must not find them: the live-reference
    tests share a pytest session with the checkpoint-fixture tests)"""
    for name in ("models.experimental", "models.common", "models.yolo", "models"):
        if getattr(sys.modules.get(name), "_yolov3_amd_alias", False):
            del sys.modules[name]


def _adopt(model: nn.Module) -> nn.Module:
    """Normalise an unpickled reference model: swap torch's parameter-free layers for the engine's stand-ins and
    add the attributes newer code expects (mirrors the compatibility loop of models/experimental.py:115-124)."""
    seq = model.model
    for i, m in enumerate(seq):
        new = None
        if isinstance(m, nn.Upsample):
            new = common.Upsample(None, int(m.scale_factor), m.mode)
        elif isinstance(m, nn.MaxPool2d):
            new = common.MaxPool2d(m.kernel_size, m.stride, m.padding)
        elif isinstance(m, nn.ZeroPad2d):
            new = common.ZeroPad2d(list(m.padding))
        if new is not None:
            for a in ("i", "f", "type", "np"):
                if hasattr(m, a):
                    setattr(new, a, getattr(m, a))
            seq[i] = new
    for m in model.modules():
        if isinstance(m, common.SPP):
            ks = tuple(p.kernel_size for p in getattr(m, "m", [])) or (5, 9, 13)
            if ks != (5, 9, 13):
                raise NotImplementedError(f"SPP kernel sizes {ks}: the HIP pyramid kernel implements (5, 9, 13)")
            m.k = ks
        if isinstance(m, nn.SiLU):
            m.inplace = True
    model.__dict__.pop("_plans", None)   # checkpoints written before the plan cache moved out of the module
    return model


def attempt_load(weights, device=None, inplace=True, fuse=True):
    """reference models/experimental.py:88-136 for a single ``.pt``: unpickle, take ``ema`` or ``model``, fp32, fuse, eval."""
    install_aliases()
    if isinstance(weights, (list, tuple)):
        if len(weights) != 1:
            raise NotImplementedError("model ensembles are outside the accelerated hot path")
        weights = weights[0]
    ckpt = torch.load(str(weights), map_location="cpu", weights_only=False)
    model = ckpt.get("ema") or ckpt["model"] if isinstance(ckpt, dict) else ckpt
    model = _adopt(model).to(device).float()
    if not hasattr(model, "stride"):
        model.stride = torch.tensor([32.0])
    if hasattr(model, "names") and isinstance(model.names, (list, tuple)):
        model.names = dict(enumerate(model.names))
    model = model.fuse().eval() if fuse and hasattr(model, "fuse") else model.eval()
    for m in model.modules():
        if isinstance(m, yolo.Detect):
            m.inplace = inplace
    return model
