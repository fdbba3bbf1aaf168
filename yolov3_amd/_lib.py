"""ctypes binding of libyolov3_hip.so (include/yolov3_hip.h).  There is NO fallback: if the library is
missing or a symbol is absent this module raises -- the product path never routes around the HIP code."""
from __future__ import annotations

import ctypes as C
import os
import threading
from pathlib import Path

_PKG = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ["Y3_LIB"]) if os.environ.get("Y3_LIB") else _PKG / "lib" / "libyolov3_hip.so"  # Y3_LIB: instrumented debug builds (tools/timeline.py)

Y3_F16, Y3_BF16, Y3_F32, Y3_U8 = 0, 1, 2, 3
Y3_ACT_NONE, Y3_ACT_SILU = 0, 1
ABI_VERSION = 5   # include/yolov3_hip.h::Y3_ABI_VERSION (2: round-3 export set -- tune / SyncBN / TTA / wgrad_plan added, y3_bn_act_bwd_apply takes sums + 2C; 3: y3_loss_params.sort_obj_iou;
                  # 4: y3_conv_workspace_error / _reset removed (no-ops since the stream-K kernel went), the loss workspace grew by one int per slot;
                  # 5: layer 0 by recomputation (y3_stem_conv_stats_only / _fwd_bn / y3_stem_bn_bwd_wgrad_recompute), y3_shard_mean, knob wgrad_patch)
Y3_ALGO_AUTO, Y3_ALGO_MFMA, Y3_ALGO_DIRECT = 0, 1, 2


class Y3Tensor(C.Structure):
    _fields_ = [("data", C.c_void_p), ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("c", C.c_int32), ("pitch", C.c_int32)]


class Y3ConvDesc(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32),
        ("ksize", C.c_int32),
        ("stride", C.c_int32),
        ("act", C.c_int32),
        ("upsample2x", C.c_int32),
        ("algo", C.c_int32),
        ("cin", C.c_int32),
        ("cout", C.c_int32),
        ("in_dilation", C.c_int32),
        ("filter_elems", C.c_int64),   # elements the packed bank holds (0 = unchecked); ctypes pads to offset 40 like the C compiler
    ]


class Y3NmsParams(C.Structure):
    _fields_ = [
        ("iou_thres", C.c_double),
        ("conf_thres", C.c_float),
        ("multi_label", C.c_int32),
        ("agnostic", C.c_int32),
        ("max_det", C.c_int32),
        ("max_nms", C.c_int32),
        ("max_wh", C.c_float),
        ("n_classes_filter", C.c_int32),
    ]


class Y3LossParams(C.Structure):
    _fields_ = [
        ("nl", C.c_int32),
        ("na", C.c_int32),
        ("nc", C.c_int32),
        ("bs", C.c_int32),
        ("ny", C.c_int32 * 5),
        ("nx", C.c_int32 * 5),
        ("anchors", C.c_float * 50),
        ("balance", C.c_float * 5),
        ("anchor_t", C.c_float),
        ("box_gain", C.c_float),
        ("obj_gain", C.c_float),
        ("cls_gain", C.c_float),
        ("cls_pw", C.c_float),
        ("obj_pw", C.c_float),
        ("cp", C.c_float),
        ("cn", C.c_float),
        ("fl_gamma", C.c_float),
        ("sort_obj_iou", C.c_int32),
    ]


_P = C.POINTER
# symbol -> (restype, argtypes).  Every symbol include/yolov3_hip.h declares is listed here and is REQUIRED.
_SIGNATURES = {
    "y3_abi_version": (C.c_int, []),
    "y3_last_error": (C.c_char_p, []),
    "y3_tune_set": (C.c_int, [C.c_char_p, C.c_int64]),
    "y3_tune_get": (C.c_int64, [C.c_char_p]),
    "y3_tune_reset": (None, []),
    "y3_packed_filter_elems": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "y3_pack_filter": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "y3_conv2d_fwd": (C.c_int, [_P(Y3ConvDesc), _P(Y3Tensor), C.c_void_p, C.c_void_p, _P(Y3Tensor), _P(Y3Tensor), C.c_void_p]),
    "y3_conv_workspace_bytes": (C.c_size_t, []),
    "y3_conv2d_fwd_bnin_rows": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]),
    "y3_conv2d_fwd_bnin_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                          C.c_void_p, C.c_void_p]),
    "y3_conv2d_fwd_ws": (C.c_int, [_P(Y3ConvDesc), _P(Y3Tensor), C.c_void_p, C.c_void_p, _P(Y3Tensor), _P(Y3Tensor), C.c_void_p, C.c_size_t, C.c_void_p]),
    "y3_conv_last_variant": (C.c_int, [C.c_char_p, C.c_size_t]),
    "y3_conv2d_fwd_variant": (C.c_int, [_P(Y3ConvDesc), _P(Y3Tensor), _P(Y3Tensor), C.c_int32, C.c_size_t, C.c_char_p, C.c_size_t]),
    "y3_conv_v10_tiles": (C.c_int, [_P(Y3ConvDesc), _P(Y3Tensor), _P(Y3Tensor), C.c_size_t, _P(C.c_int32), C.c_int64, _P(C.c_int64), _P(C.c_int32), _P(C.c_int32)]),
    "y3_nchw_to_nhwc": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, _P(Y3Tensor), C.c_void_p]),
    "y3_nhwc_to_nchw": (C.c_int, [_P(Y3Tensor), C.c_int32, C.c_void_p, C.c_void_p]),
    "y3_maxpool2d": (C.c_int, [_P(Y3Tensor), _P(Y3Tensor), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "y3_spp_pyramid": (C.c_int, [_P(Y3Tensor), _P(Y3Tensor), C.c_int32, C.c_void_p]),
    "y3_upsample2x": (C.c_int, [_P(Y3Tensor), _P(Y3Tensor), C.c_int32, C.c_void_p]),
    "y3_copy_slice": (C.c_int, [_P(Y3Tensor), _P(Y3Tensor), C.c_int32, C.c_void_p]),
    "y3_detect_decode": (C.c_int, [_P(Y3Tensor), C.c_int32, C.c_int32, C.c_int32, _P(C.c_float), C.c_float, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "y3_nms_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, _P(Y3NmsParams), C.c_int64]),
    "y3_nms": (
        C.c_int,
        [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P(Y3NmsParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p],
    ),
    "y3_bneck_pair_fwd": (C.c_int, [_P(Y3Tensor), C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, _P(Y3Tensor), C.c_void_p]),
    "y3_stem_pair_fwd": (
        C.c_int,
        [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, _P(Y3Tensor),
         C.c_void_p],
    ),
    "y3_letterbox_u8": (
        C.c_int,
        [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p],
    ),
    "y3_scale_boxes": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "y3_match_detections": (
        C.c_int,
        [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p],
    ),
    "y3_loss_workspace_bytes": (C.c_size_t, [_P(Y3LossParams), C.c_int32]),
    "y3_loss_fwd": (C.c_int, [_P(Y3LossParams), C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "y3_loss_bwd": (C.c_int, [_P(Y3LossParams), C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "y3_bn_stats": (C.c_int, [_P(Y3Tensor), C.c_int32, C.c_void_p, C.c_void_p]),
    "y3_bn_finalize": (
        C.c_int,
        [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "y3_scale_img": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]),
    "y3_descale_pred": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_float, C.c_float, C.c_void_p, C.c_int32, C.c_int32,
                                  C.c_void_p]),
    "y3_loss_level_obj": (C.c_int, [_P(Y3LossParams), C.c_int32, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "y3_conv2d_fwd_stats_rows": (C.c_int64, [_P(Y3ConvDesc), _P(Y3Tensor), _P(Y3Tensor)]),
    "y3_conv2d_fwd_stats": (C.c_int, [_P(Y3ConvDesc), _P(Y3Tensor), C.c_void_p, C.c_void_p, _P(Y3Tensor), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "y3_conv2d_fwd_stats_rows_ws": (C.c_int64, [_P(Y3ConvDesc), _P(Y3Tensor), _P(Y3Tensor), C.c_size_t]),
    "y3_conv2d_fwd_stats_ws": (C.c_int, [_P(Y3ConvDesc), _P(Y3Tensor), C.c_void_p, C.c_void_p, _P(Y3Tensor), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "y3_bn_finalize_rows": (
        C.c_int,
        [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
         C.c_void_p],
    ),
    "y3_bn_stats_finalize": (
        C.c_int,
        [_P(Y3Tensor), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "y3_bn_act_fwd": (C.c_int, [_P(Y3Tensor), C.c_void_p, C.c_void_p, _P(Y3Tensor), _P(Y3Tensor), C.c_int32, C.c_int32, C.c_void_p]),
    "y3_bn_sum_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "y3_bn_finalize_devcount": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "y3_bn_act_bwd_reduce": (
        C.c_int,
        [_P(Y3Tensor), _P(Y3Tensor), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "y3_bn_act_bwd_apply": (
        C.c_int,
        [_P(Y3Tensor), _P(Y3Tensor), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, _P(Y3Tensor), _P(Y3Tensor), C.c_int32, C.c_void_p],
    ),
    "y3_bn_act_bwd": (
        C.c_int,
        [_P(Y3Tensor), _P(Y3Tensor), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, _P(Y3Tensor), C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "y3_bn_act_bwd_res": (
        C.c_int,
        [_P(Y3Tensor), _P(Y3Tensor), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, _P(Y3Tensor), C.c_void_p, C.c_void_p, _P(Y3Tensor), C.c_int32,
         C.c_void_p],
    ),
    "y3_pack_filter_dgrad": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "y3_pack_filter_pair": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "y3_pack_job_blocks": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "y3_pack_filter_jobs": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p]),
    "y3_packed_filter_dgrad_s2_elems": (C.c_size_t, [C.c_int32, C.c_int32]),
    "y3_pack_filter_dgrad_s2": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "y3_conv2d_dgrad_s2": (C.c_int, [C.c_int32, _P(Y3Tensor), C.c_void_p, _P(Y3Tensor), _P(Y3Tensor), C.c_void_p]),
    "y3_conv2d_wgrad_workspace_bytes": (C.c_size_t, [_P(Y3ConvDesc), _P(Y3Tensor)]),
    "y3_conv2d_wgrad_plan": (C.c_int, [_P(Y3ConvDesc), _P(Y3Tensor), C.c_void_p, C.c_void_p, C.c_void_p]),
    "y3_conv2d_wgrad": (C.c_int, [_P(Y3ConvDesc), _P(Y3Tensor), _P(Y3Tensor), C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "y3_upsample2x_bwd": (C.c_int, [_P(Y3Tensor), _P(Y3Tensor), C.c_int32, C.c_int32, C.c_void_p]),
    "y3_maxpool2d_bwd": (C.c_int, [_P(Y3Tensor), _P(Y3Tensor), _P(Y3Tensor), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "y3_maxpool2d_bwd_workspace_bytes": (C.c_size_t, [_P(Y3Tensor), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "y3_maxpool2d_bwd_ws": (C.c_int, [_P(Y3Tensor), _P(Y3Tensor), _P(Y3Tensor), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t,
                                      C.c_void_p]),
    "y3_detect_raw_bwd": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P(Y3Tensor), C.c_void_p]),
    "y3_packed_filter_stem_elems": (C.c_size_t, [C.c_int32]),
    "y3_pack_filter_stem": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "y3_stem_conv_fwd": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, _P(Y3Tensor), C.c_void_p]),
    "y3_stem_bn_bwd_wgrad_workspace_bytes": (C.c_size_t, []),
    "y3_stem_bn_bwd_wgrad": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, _P(Y3Tensor), _P(Y3Tensor), C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "y3_stem_bn_bwd_wgrad_recompute": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, _P(Y3Tensor), C.c_void_p, C.c_void_p,
                                                 C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "y3_stem_conv_stats_rows": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "y3_stem_conv_stats_only": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_int32, _P(Y3Tensor), C.c_void_p, C.c_int64,
                                          C.c_void_p, C.c_void_p]),
    "y3_stem_conv_fwd_bn": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                      _P(Y3Tensor), C.c_void_p]),
    "y3_stem_conv_fwd_stats": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, _P(Y3Tensor),
                                         C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "y3_sgd_tensor_record_bytes": (C.c_size_t, []),
    "y3_sgd_step_dynamic": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_float, C.c_float, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "y3_loss_scale_update": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int32, C.c_void_p]),
    "y3_shard_mean": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_void_p, C.c_void_p]),
    "y3_sgd_step": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
}

_lib = None


class Y3Error(RuntimeError):
    pass


def exported_symbols():
    return list(_SIGNATURES)


def lib() -> C.CDLL:
    """Load libyolov3_hip.so (once).  Import torch first so the HIP runtime torch ships is the one the library
    binds to (same SONAME libamdhip64.so.7): streams and device pointers then belong to one runtime."""
    global _lib
    if _lib is not None:
        t = _call_timer
        return _lib if t is None else t
    import torch  # noqa: F401  (must precede the dlopen, see docstring)

    if not LIB_PATH.exists():
        raise Y3Error(
            f"{LIB_PATH} is missing: the MI355X HIP library has not been built. Run `python -m yolov3_amd.build` "
            "(needs hipcc, gfx950). There is no CPU or PyTorch fallback for this path."
        )
    handle = C.CDLL(str(LIB_PATH))
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as e:
            raise Y3Error(f"{LIB_PATH} does not export {name}; rebuild with `python -m yolov3_amd.build --force`") from e
        fn.restype, fn.argtypes = res, args
    if handle.y3_abi_version() != ABI_VERSION:
        raise Y3Error(f"ABI version mismatch: library reports {handle.y3_abi_version()}, bindings expect {ABI_VERSION}")
    _lib = handle
    return handle


_call_timer = None
# exports that launch nothing (sizes, plans, names, knobs): not timed
_QUERIES = frozenset((
    "y3_abi_version", "y3_last_error", "y3_tune_set", "y3_tune_get", "y3_tune_reset", "y3_packed_filter_elems", "y3_conv_workspace_bytes", "y3_conv_last_variant",
    "y3_conv2d_fwd_variant", "y3_nms_workspace_bytes", "y3_loss_workspace_bytes", "y3_conv2d_fwd_stats_rows", "y3_conv2d_fwd_stats_rows_ws", "y3_conv2d_fwd_bnin_rows", "y3_pack_job_blocks",
    "y3_packed_filter_dgrad_s2_elems", "y3_conv2d_wgrad_workspace_bytes", "y3_conv2d_wgrad_plan", "y3_packed_filter_stem_elems", "y3_stem_bn_bwd_wgrad_workspace_bytes",
    "y3_stem_conv_stats_rows", "y3_conv_v10_tiles", "y3_sgd_tensor_record_bytes", "y3_maxpool2d_bwd_workspace_bytes"))


class CallTimer:
    """HIP-event timing of every C-ABI call made while the timer is installed (`with CallTimer() as t: step()`), on the stream the kernels are launched on (torch's
    current stream: what ops.stream_ptr() hands to the library).  bench.py uses it to split ONE training step into kernel families in the run that reports it,
    instead of quoting a profile taken elsewhere.  `by_function()` -> {C function: [milliseconds, calls]} once the stream has drained.  Bench-only: while it
    is installed EVERY thread's calls are timed (autograd runs the backward on its own thread), each on the stream current in that thread; code that cached the real handle
    before bypasses it; it stops recording at MAX_RECORDS."""

    MAX_RECORDS = 100_000   # bench-only tool: a timer left installed around a long loop stops recording instead of growing without bound

    def __init__(self):
        self.records = []   # (function name, start event, end event)
        self._fns = {}
        self._lock = threading.Lock()         # the backward of a step runs on autograd's own thread: every thread of the process is timed, appends are serialised
        self.dropped = 0

    def __getattr__(self, name):   # stands in for the ctypes handle: lib().y3_xxx(...)
        fn = self._fns.get(name)
        if fn is None:
            import torch

            raw = getattr(_lib, name)
            if name in _QUERIES:
                fn = raw          # queries: nothing is launched
            else:
                def fn(*a, raw_=raw, name_=name):
                    if len(self.records) >= self.MAX_RECORDS:
                        self.dropped += 1
                        return raw_(*a)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    r = raw_(*a)
                    e1.record()
                    with self._lock:
                        self.records.append((name_, e0, e1))
                    return r
            self._fns[name] = fn
        return fn

    def __enter__(self):
        global _call_timer
        lib()   # (loaded before the proxy stands in for it)
        _call_timer = self
        return self

    def __exit__(self, *exc):
        global _call_timer
        _call_timer = None
        return False

    def by_function(self):
        import torch

        torch.cuda.synchronize()
        out = {}
        for name, e0, e1 in self.records:
            a = out.setdefault(name, [0.0, 0])
            a[0] += e0.elapsed_time(e1)
            a[1] += 1
        return out


def check(status: int, what: str = ""):
    if status != 0:
        msg = lib().y3_last_error().decode(errors="replace")
        raise Y3Error(f"{what}: {msg}" if what else msg)
