"""Import the UNMODIFIED reference (/root/reference) on a box without ultralytics /
torchvision / cv2 / seaborn -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Only usable where /root/reference exists (the build container).  It is used by
tests/golden/make_golden.py to generate golden vectors and by the ``-m "not gpu"`` tests
that pin oracle/yolo_oracle.py against the reference itself.  Nothing here runs on the GPU
box, and nothing here is imported by the product package.

Mechanism: register permissive stub modules for the missing
third-party packages in ``sys.modules``; the arithmetic the hot path needs from them is
bound to the restatements in oracle/upstream.py.
"""
from __future__ import annotations

import contextlib
import logging
import sys
import types
from pathlib import Path

import torch

from . import upstream

REFERENCE_ROOT = Path("/root/reference")


def available() -> bool:
    return (REFERENCE_ROOT / "models" / "yolo.py").exists()


class _Anything:
    """Callable / iterable / context-manager / decorator dummy for symbols nobody computes with."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]  # used as a bare decorator
        return _Anything()

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Anything()

    def __iter__(self):
        return iter(())

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def __bool__(self):
        return False


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Anything()


def _mod(name: str, **attrs) -> types.ModuleType:
    m = _StubModule(name)
    m.__path__ = []  # behave as a package so submodule imports resolve through sys.modules
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], child, m)
    return m


class _TryExcept(contextlib.ContextDecorator):
    """Usable as ``@TryExcept()`` / ``@TryExcept("msg")`` and ``with TryExcept():``."""

    def __init__(self, msg="", verbose=True):
        self.msg = msg

    def __enter__(self):
        return self

    def __exit__(self, exc_type, value, tb):
        return True


class _WorkingDirectory(contextlib.ContextDecorator):
    def __init__(self, new_dir=None):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


class _Profile(contextlib.ContextDecorator):
    def __init__(self, t=0.0, device=None):
        self.t, self.dt = t, 0.0

    def __enter__(self):
        import time

        self._s = time.perf_counter()
        return self

    def __exit__(self, *exc):
        import time

        self.dt = time.perf_counter() - self._s
        self.t += self.dt
        return False


def _threaded(fn):
    return fn


def install() -> None:
    """Register the stubs and put the reference on ``sys.path`` (idempotent)."""
    if getattr(install, "_done", False):
        return
    if not available():
        raise RuntimeError("reference tree /root/reference not present (GPU box?) -- ref_shim unusable")

    logger = logging.getLogger("ref_shim")
    logger.setLevel(logging.ERROR)

    _mod("cv2", setNumThreads=lambda n: None, INTER_LINEAR=1, INTER_AREA=3, INTER_NEAREST=0, INTER_CUBIC=2,
         INTER_LANCZOS4=4, IMREAD_COLOR=1, BORDER_CONSTANT=0)
    _mod("seaborn")
    _mod("torchvision", __version__="0.0.0+oracle")
    _mod("torchvision.ops", nms=upstream.nms)
    _mod("ultralytics", __version__="8.4.110+oracle")
    _mod(
        "ultralytics.utils",
        LOGGER=logger,
        TQDM=_Anything(),
        colorstr=lambda *a: str(a[-1]) if a else "",
        get_default_args=lambda f: {},
        TryExcept=_TryExcept,
        emojis=lambda s="": s,
        threaded=_threaded,
    )
    _mod(
        "ultralytics.utils.checks",
        check_version=lambda *a, **k: True,
        check_requirements=lambda *a, **k: True,
        is_ascii=lambda s="": all(ord(c) < 128 for c in str(s)),
        print_args=lambda *a, **k: None,
    )
    _mod("ultralytics.utils.files", WorkingDirectory=_WorkingDirectory)
    _mod("ultralytics.utils.git")
    _mod("ultralytics.utils.patches", torch_load=torch.load)
    _mod(
        "ultralytics.utils.ops",
        Profile=_Profile,
        clip_boxes=upstream.clip_boxes,
        make_divisible=upstream.make_divisible,
        xywh2xyxy=upstream.xywh2xyxy,
        xyxy2xywh=upstream.xyxy2xywh,
    )
    _mod(
        "ultralytics.utils.torch_utils",
        fuse_conv_and_bn=upstream.fuse_conv_and_bn,
        initialize_weights=upstream.initialize_weights,
        intersect_dicts=upstream.intersect_dicts,
        one_cycle=upstream.one_cycle,
        scale_img=upstream.scale_img,
        model_info=lambda *a, **k: None,
        time_sync=lambda: __import__("time").perf_counter(),
        autocast=lambda *a, **k: contextlib.nullcontext(),
        TORCH_2_4=True,
        copy_attr=lambda a, b, include=(), exclude=(): None,
        smart_inference_mode=lambda *a, **k: (lambda fn: fn),
    )
    _mod(
        "ultralytics.utils.metrics",
        bbox_iou=upstream.bbox_iou,
        box_iou=upstream.box_iou,
        smooth_bce=upstream.smooth_bce,
        smooth=upstream.smooth,
        plot_mc_curve=lambda *a, **k: None,
        plot_pr_curve=lambda *a, **k: None,
    )
    _mod("ultralytics.utils.plotting")
    _mod("ultralytics.data")
    _mod("ultralytics.data.converter")
    _mod("ultralytics.data.build")
    _mod("ultralytics.data.utils")

    if str(REFERENCE_ROOT) not in sys.path:
        sys.path.insert(0, str(REFERENCE_ROOT))
    install._done = True


def load():
    """Return a namespace with the reference's hot-path symbols (unmodified code)."""
    install()
    from models.yolo import Detect, DetectionModel  # type: ignore  # noqa: E402
    from utils.general import non_max_suppression  # type: ignore
    from utils.loss import ComputeLoss  # type: ignore

    ns = types.SimpleNamespace(
        DetectionModel=DetectionModel,
        Detect=Detect,
        non_max_suppression=non_max_suppression,
        ComputeLoss=ComputeLoss,
        root=REFERENCE_ROOT,
    )
    torch.set_printoptions(profile="default")
    return ns
