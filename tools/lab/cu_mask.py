"""CU-masked HIP streams (hipExtStreamCreateWithCUMask) as torch streams -- a LAB tool (tools/cu_mask_probe.py, tools/wgrad_overlap_ab.py).
Measured in round 4 (profiles/r04_wgrad_overlap_probe.txt): confining the filter gradients to a CU subset works per kernel and fails per hand-over
(~0.2 ms per cross-stream event on a masked queue), so the product (yolov3_amd.train_engine) does not use it."""
import ctypes as C

import torch

_MASKED_STREAMS = {}


def masked_stream(device, n_cus: int, first: int = 0):
    """stream whose kernels may only occupy CUs [first, first + n_cus) of the CU-mask bit vector; one per (device, n_cus, first) and process (never destroyed:
    a lab process is short-lived)"""
    total = torch.cuda.get_device_properties(device).multi_processor_count
    n_cus, first = int(n_cus), int(first)
    if not (0 < n_cus and 0 <= first and first + n_cus <= total):
        raise ValueError(f"CU mask [{first}, {first + n_cus}) does not fit the device's {total} CUs")
    key = (torch.device(device).index or 0, n_cus, first)
    st = _MASKED_STREAMS.get(key)
    if st is None:
        hip = C.CDLL("libamdhip64.so")
        words = max(1, (total + 31) // 32)
        mask = (C.c_uint32 * words)()
        for b in range(first, first + n_cus):
            mask[b // 32] |= 1 << (b % 32)
        handle = C.c_void_p()
        with torch.cuda.device(device):
            rc = hip.hipExtStreamCreateWithCUMask(C.byref(handle), C.c_uint32(words), mask)
        if rc != 0 or not handle.value:
            raise RuntimeError(f"hipExtStreamCreateWithCUMask({n_cus} of {total} CUs) failed: hipError {rc}")
        st = _MASKED_STREAMS[key] = torch.cuda.ExternalStream(handle.value, device=device)
    return st
