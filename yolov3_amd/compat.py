"""Checkpoint ingest for reference ``.pt`` files.

The reference saves PICKLED ``nn.Module`` objects (train.py:470-488: ``{"model": deepcopy(model).half(), "ema": ...}``)
and loads them with ``torch.load`` (models/experimental.py:88-136 ``attempt_load``), which resolves classes by their
qualified names ``models.yolo.DetectionModel``, ``models.yolo.Detect``, ``models.common.Conv`` ...

``attempt_load`` resolves those names through a SCOPED unpickler (``torch.load(pickle_module=...)``): ``find_class`` maps
the reference's qualified names to the yolov3_amd classes for that one load and ``sys.modules`` is never touched, so it
behaves the same in a bare process (the GPU box) and inside a live reference process that has the real ``models.yolo`` /
``models.common`` imported (train.py:50-51, val.py:39, utils/general.py:432).  Whatever the pickle resolved to --
including the reference's own classes when somebody else's ``torch.load`` produced the object -- goes through ``adopt``,
which re-classes the module tree (the attribute names -- ``conv/bn/act``, ``cv1/cv2/add``, ``m/anchors/stride`` -- are
the same) and swaps torch's parameter-free layers for the stand-ins the engine plans with.

``install_aliases()`` remains for callers that use a plain ``torch.load`` in a process WITHOUT a reference checkout: it
registers ``models`` / ``models.yolo`` / ``models.common`` / ``models.experimental`` alias modules, and only then: a real
``models`` package that is imported or importable is never shadowed or overwritten.
"""
from __future__ import annotations

import importlib.util
import pickle
import sys
import types

import torch
from torch import nn

from . import common, yolo

_YOLO_NAMES = ("Detect", "DetectionModel", "Model", "BaseModel", "parse_model")
_COMMON_NAMES = ("Conv", "Bottleneck", "SPP", "Concat", "autopad")
_ALIAS_MODULES = ("models", "models.yolo", "models.common", "models.experimental")


def _class_map() -> dict:
    """qualified reference name -> the class of this package that stands in for it"""
    table = {("models.yolo", n): getattr(yolo, n) for n in _YOLO_NAMES}
    table.update({("models.common", n): getattr(common, n) for n in _COMMON_NAMES})
    return table


class _Unpickler(pickle.Unpickler):
    """``find_class`` with the reference's module paths mapped to this package; everything else as usual.  A ``models.*`` class that has
    no MI355X implementation (the unused layer zoo of models/common.py) is refused by name instead of importing whatever
    ``models`` package the process happens to see."""

    def find_class(self, module, name):
        hit = _class_map().get((module, name))
        if hit is not None:
            return hit
        if module in ("models.yolo", "models.common", "models.experimental") or module.startswith("models."):
            raise NotImplementedError(f"checkpoint refers to {module}.{name}, which no yolov3*.yaml uses and which has no MI355X implementation")
        return super().find_class(module, name)


def _pickle_module() -> types.ModuleType:
    """a stand-in for the ``pickle`` module whose Unpickler is the scoped one (what ``torch.load(pickle_module=...)`` expects)"""
    mod = types.ModuleType("yolov3_amd._scoped_pickle")
    mod.__dict__.update({k: v for k, v in pickle.__dict__.items() if not k.startswith("__")})
    mod.Unpickler = _Unpickler

    def load(file, **kw):
        return _Unpickler(file, **kw).load()

    def loads(data, **kw):
        import io

        return _Unpickler(io.BytesIO(data), **kw).load()

    mod.load, mod.loads = load, loads
    return mod


def _real_reference_visible() -> bool:
    """a ``models`` package that is not one of our aliases is imported, or would be found by an import"""
    for name in _ALIAS_MODULES:
        m = sys.modules.get(name)
        if m is not None and not getattr(m, "_yolov3_amd_alias", False):
            return True
    if "models" in sys.modules:          # our alias package
        return False
    try:
        return importlib.util.find_spec("models") is not None
    except (ImportError, ValueError):
        return False


def install_aliases(force: bool = False) -> bool:
    """Make ``models.yolo`` / ``models.common`` importable names for a plain ``torch.load`` in a process without a reference checkout.
    Returns True if the aliases are (now) installed.  Never replaces or shadows a real ``models`` package (``force`` is accepted for
    backward compatibility and re-installs our own aliases only); ``attempt_load`` does not need this."""
    if _real_reference_visible():
        return False
    if not force and all(getattr(sys.modules.get(n), "_yolov3_amd_alias", False) for n in _ALIAS_MODULES):
        return True
    pkg = types.ModuleType("models")
    pkg.__path__ = []
    m_yolo = types.ModuleType("models.yolo")
    m_common = types.ModuleType("models.common")
    m_exp = types.ModuleType("models.experimental")
    for name in _YOLO_NAMES:
        setattr(m_yolo, name, getattr(yolo, name))
    for name in _COMMON_NAMES:
        setattr(m_common, name, getattr(common, name))
    from .autoshape import AutoShape
    from .backend import DetectMultiBackend

    m_common.AutoShape, m_common.DetectMultiBackend = AutoShape, DetectMultiBackend   # utils/general.py:432 imports both
    m_exp.attempt_load = attempt_load
    for m in (pkg, m_yolo, m_common, m_exp):
        m._yolov3_amd_alias = True
    pkg.yolo, pkg.common, pkg.experimental = m_yolo, m_common, m_exp
    sys.modules.update({"models": pkg, "models.yolo": m_yolo, "models.common": m_common, "models.experimental": m_exp})
    return True


def uninstall_aliases():
    """remove the alias modules again (only ours: a real reference package is left alone)"""
    for name in reversed(_ALIAS_MODULES):
        if getattr(sys.modules.get(name), "_yolov3_amd_alias", False):
            del sys.modules[name]


def _reclass(model: nn.Module):
    """objects of the reference's own classes (a model unpickled by somebody else's ``torch.load`` inside a reference process, or built by
    the reference's ``DetectionModel(cfg)``) become objects of this package's classes: same attributes, same parameters, our ``forward``"""
    table = _class_map()
    ours = set(table.values())
    for m in model.modules():
        cls = type(m)
        if cls in ours or not cls.__module__.startswith("models."):
            continue
        hit = table.get((cls.__module__, cls.__name__))
        if hit is None or not isinstance(hit, type):
            raise NotImplementedError(f"{cls.__module__}.{cls.__name__} is not used by any yolov3*.yaml and has no MI355X implementation")
        m.__class__ = hit


def adopt(model: nn.Module) -> nn.Module:
    """Normalise an unpickled reference model: re-class reference objects, swap torch's parameter-free layers for the engine's stand-ins
    and add the attributes newer code expects (mirrors the compatibility loop of models/experimental.py:115-124)."""
    _reclass(model)
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
        if isinstance(m, common.Conv):
            c = m.conv
            if c.groups != 1 or c.dilation != (1, 1) or c.kernel_size not in ((1, 1), (3, 3)) or c.stride not in ((1, 1), (2, 2)):
                raise NotImplementedError(f"Conv {c}: the HIP path implements k in (1,3), s in (1,2), g = d = 1 (all yolov3*.yaml shapes)")
            if not isinstance(m.act, (nn.SiLU, nn.Identity)):
                raise NotImplementedError("only SiLU / Identity activations have a fused HIP epilogue")
        if isinstance(m, common.SPP):
            ks = tuple(p.kernel_size for p in getattr(m, "m", [])) or (5, 9, 13)
            if ks != (5, 9, 13):
                raise NotImplementedError(f"SPP kernel sizes {ks}: the HIP pyramid kernel implements (5, 9, 13)")
            m.k = ks
        if isinstance(m, common.Concat) and m.d != 1:
            raise NotImplementedError("Concat along channels only")
        if isinstance(m, nn.SiLU):
            m.inplace = True
    model.__dict__.pop("_plans", None)   # checkpoints written before the plan cache moved out of the module
    return model


_adopt = adopt   # earlier name


def load_checkpoint(weights):
    """``torch.load`` of a reference ``.pt`` with the reference's class paths resolved to this package for this one call"""
    return torch.load(str(weights), map_location="cpu", weights_only=False, pickle_module=_pickle_module())


def attempt_load(weights, device=None, inplace=True, fuse=True):
    """reference models/experimental.py:88-136 for a single ``.pt``: unpickle, take ``ema`` or ``model``, fp32, fuse, eval."""
    if isinstance(weights, (list, tuple)):
        if len(weights) != 1:
            raise NotImplementedError("model ensembles are outside the accelerated hot path")
        weights = weights[0]
    ckpt = load_checkpoint(weights)
    model = ckpt.get("ema") or ckpt["model"] if isinstance(ckpt, dict) else ckpt
    model = adopt(model).to(device).float()
    if not hasattr(model, "stride"):
        model.stride = torch.tensor([32.0])
    if hasattr(model, "names") and isinstance(model.names, (list, tuple)):
        model.names = dict(enumerate(model.names))
    model = model.fuse().eval() if fuse and hasattr(model, "fuse") else model.eval()
    for m in model.modules():
        if isinstance(m, yolo.Detect):
            m.inplace = inplace
    return model
