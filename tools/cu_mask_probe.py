"""What a CU-masked HIP stream (masked_stream) gives one kernel when nothing else runs: one MFMA-bound launch (the 512 -> 1024 3x3 filter gradient at batch 64)
and one HBM-bound pass (y3_bn_act_fwd over a 420 MB tensor) per mask size.  Usage: python tools/cu_mask_probe.py"""
import ctypes as C, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from yolov3_amd import _lib, ops
from yolov3_amd.ops import View
sys.path.insert(0, str(Path(__file__).resolve().parent / "lab"))
from cu_mask import masked_stream

dev = torch.device("cuda:0")
dt = torch.float16
x = View.alloc(64, 20, 20, 512, dt, dev); x.buf.normal_()
du = View.alloc(64, 20, 20, 1024, dt, dev); du.buf.normal_()
u = View.alloc(64, 80, 80, 256, dt, dev); u.buf.normal_()
y = View.alloc(64, 80, 80, 256, dt, dev)
sc, sh = torch.ones(256, device=dev), torch.zeros(256, device=dev)


def wgrad():
    ops.conv2d_wgrad(x, du, 3, 1, 1024, 512)


def bn():
    ut, yt = u.y3(), y.y3()
    ops.check(_lib.lib().y3_bn_act_fwd(C.byref(ut), sc.data_ptr(), sh.data_ptr(), None, C.byref(yt), ops.dtype_code(dt), _lib.Y3_ACT_SILU, ops.stream_ptr()), "bn")


def timed(fn, st, reps=20):
    with torch.cuda.stream(st):
        for _ in range(3):
            fn()
        st.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        st.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


print(f"{'stream':>12s} {'wgrad us':>10s} {'bn pass us':>11s} {'GB/s':>8s}")
for name, st in [("plain", torch.cuda.Stream(device=dev))] + [(f"mask {n}", masked_stream(dev, n)) for n in (256, 248, 224, 192, 160, 128, 96, 64, 32)] + \
        [("mask 64@192", masked_stream(dev, 64, 192)), ("mask 32@0 x", masked_stream(dev, 32, 32))]:
    tw, tb = timed(wgrad, st), timed(bn, st)
    print(f"{name:>12s} {tw:10.1f} {tb:11.1f} {2 * u.buf.numel() * 2 / tb / 1e3:8.0f}", flush=True)
