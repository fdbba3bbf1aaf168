"""Where does a conv launch spend its time?  Builds an instrumented copy of the library (-DY3_TIMELINE: thread 0 of every
block stamps wall_clock64() at entry / first DMA issued / first tile landed / K-loop done / stores issued) and prints, per
layer shape, the launch span, the dispatch ramp and the per-block phase medians.  GPU box only:

    python tools/timeline.py            # builds yolov3_amd/lib/libyolov3_hip_tl.so, then re-executes itself with Y3_LIB set
"""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
TL = ROOT / "yolov3_amd" / "lib" / "libyolov3_hip_tl.so"


def build_tl():
    from yolov3_amd import build as B
    cc = B.hipcc()
    objs = []
    for src, extra in B.SOURCES:
        s = B.CSRC / src
        o = B.OBJ_DIR / (s.stem + ("_tl.o" if src == "conv.hip" else ".o"))
        if src == "conv.hip":
            subprocess.check_call([cc, *B.COMMON, *extra, "-DY3_TIMELINE", "-x", "hip", "-c", str(s), "-o", str(o)])
        objs.append(str(o))
    subprocess.check_call([cc, "-shared", "-fPIC", f"--offload-arch={B.ARCH}", *objs, "-o", str(TL)])


SHAPES = [  # name, cin, cout, k, stride, H(in), batch
    ("L16   1x1  512->256  @20", 512, 256, 1, 1, 20, 32),
    ("L10c1 1x1 1024->512  @20", 1024, 512, 1, 1, 20, 32),
    ("L8c1  1x1  512->256  @40", 512, 256, 1, 1, 40, 32),
    ("L6c1  1x1  256->128  @80", 256, 128, 1, 1, 80, 32),
    ("L4c1  1x1  128->64  @160", 128, 64, 1, 1, 160, 32),
    ("L6c2  3x3  128->256  @80", 128, 256, 3, 1, 80, 32),
    ("L6c2 +residual", 128, 256, 3, 1, 80, 32),
    ("L4c2  3x3 64->128 @160 +residual", 64, 128, 3, 1, 160, 32),
    ("L8c2  3x3  256->512  @40", 256, 512, 3, 1, 40, 32),
    ("L10c2 3x3  512->1024 @20", 512, 1024, 3, 1, 20, 32),
    ("L3    3x3s2 64->128 @320", 64, 128, 3, 2, 320, 32),
]


def main():
    import ctypes as C
    import numpy as np
    import torch
    from yolov3_amd import _lib, ops

    L = _lib.lib()
    L.y3_debug_timeline.argtypes = [C.c_void_p]
    L.y3_debug_timeline.restype = None
    dev = torch.device("cuda:0")
    print(f"{'layer':34s} {'blocks':>6s} {'event us':>9s} {'span us':>8s} | ramp p50/p90/max | prologue  1st-tile  k-loop  epilogue (us, median / p90)")
    for name, cin, cout, k, s, H, n in SHAPES:
        Ho = (H + 2 * (k // 2) - k) // s + 1
        x = ops.View.alloc(n, H, H, cin, torch.float16, dev)
        x.buf.normal_()
        y = ops.View.alloc(n, Ho, Ho, cout, torch.float16, dev)
        w = torch.randn(cout, cin, k, k, device=dev) * 0.05
        filt = ops.pack_filter(w, cout, cin, torch.float16)
        bias = torch.zeros(cout, device=dev)
        res = None
        if "residual" in name:
            res = ops.View.alloc(n, Ho, Ho, cout, torch.float16, dev)
            res.buf.normal_()
        tl = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
        L.y3_debug_timeline(None)
        for _ in range(3):
            ops.conv2d(x, filt, bias, y, k, s, True, residual=res)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            ops.conv2d(x, filt, bias, y, k, s, True, residual=res)
        e1.record()
        torch.cuda.synchronize()
        ev_us = e0.elapsed_time(e1) * 100.0
        L.y3_debug_timeline(tl.data_ptr())
        ops.conv2d(x, filt, bias, y, k, s, True, residual=res)
        torch.cuda.synchronize()
        L.y3_debug_timeline(None)
        t = tl.cpu().numpy().reshape(-1, 8)
        t = t[t[:, 0] != 0][:, :5].astype(np.float64) / 100.0  # 100 MHz -> us
        t0 = t[:, 0].min()
        ramp = t[:, 0] - t0
        ph = np.diff(t, axis=1)
        q = lambda a, p: np.percentile(a, p)
        print(f"{name:34s} {len(t):6d} {ev_us:9.1f} {t[:, 4].max() - t0:8.1f} | {q(ramp, 50):5.1f}/{q(ramp, 90):5.1f}/{ramp.max():5.1f} | "
              + "  ".join(f"{q(ph[:, i], 50):5.1f}/{q(ph[:, i], 90):5.1f}" for i in range(4)), flush=True)


if __name__ == "__main__":
    if os.environ.get("Y3_LIB"):
        main()
    else:
        if not TL.exists() or TL.stat().st_mtime < (ROOT / "yolov3_amd" / "csrc" / "conv.hip").stat().st_mtime:
            build_tl()
        os.execve(sys.executable, [sys.executable, __file__], dict(os.environ, Y3_LIB=str(TL)))
