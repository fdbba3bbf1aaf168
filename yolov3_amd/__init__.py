"""yolov3_amd -- MI355X-native (gfx950) implementation of the ultralytics/yolov3 detection hot path.

Public surface mirrors the reference's own Python signatures:
    DetectionModel / Model / Detect      (reference models/yolo.py)
    Conv / Bottleneck / SPP / Concat     (reference models/common.py)
    non_max_suppression, scale_boxes     (reference utils/general.py; + batched forms)
    process_batch                        (reference val.py:147-188; + batched form)
    ap_per_class / compute_ap / fitness  (reference utils/metrics.py:15-118; host NumPy, as in the reference)
    ComputeLoss                          (reference utils/loss.py)
    FusedSGD / GradScaler / ModelEMA     (reference train.py:345,411-422: scaler.scale / unscale_ / clip / step / update / ema.update)
    DetectMultiBackend (.pt branch), attempt_load, AutoShape   (reference models/common.py, models/experimental.py)
Everything executes through libyolov3_hip.so (include/yolov3_hip.h); there is no CPU/PyTorch fallback.
"""
from .common import SPP, Bottleneck, Concat, Conv  # noqa: F401
from .general import non_max_suppression, non_max_suppression_batched, scale_boxes, scale_boxes_batched, xywh2xyxy, clip_boxes  # noqa: F401
from .val import detect_batches, process_batch, process_batch_batched  # noqa: F401
from .metrics import ap_per_class, compute_ap, fitness  # noqa: F401
from .backend import DetectMultiBackend  # noqa: F401
from .autoshape import AutoShape, Detections, letterbox_batch  # noqa: F401
from .compat import attempt_load  # noqa: F401
from .loss import ComputeLoss  # noqa: F401
from .optim import FusedSGD, GradScaler, ModelEMA, smart_param_groups  # noqa: F401
from .yolo import Detect, DetectionModel, Model, parse_model  # noqa: F401

__version__ = "0.1.0"
