"""Where does a tile of conv_v10.h spend its time?  Instrumented copy of the library (-DY3_TIMELINE: thread 0 of every block stamps the 100 MHz wall clock at block
start and, per tile, at K-loop start / K-loop end / epilogue end); prints per layer shape and form the block span and the per-tile phase medians.  GPU box only:

    python tools/v10_timeline.py --build   (here)      python tools/v10_timeline.py   (GPU box; delete the lab library afterwards)
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
    B.build(verbose=False)
    cc = B.hipcc()
    objs = []
    for src, extra in B.SOURCES:
        s = B.CSRC / src
        o = B.OBJ_DIR / (s.stem + ("_tl.o" if src == "conv.hip" else ".o"))
        if src == "conv.hip":
            subprocess.check_call([cc, *B.COMMON, *extra, "-DY3_TIMELINE", "-x", "hip", "-c", str(s), "-o", str(o)])
        objs.append(str(o))
    subprocess.check_call([cc, "-shared", "-fPIC", f"--offload-arch={B.ARCH}", *objs, "-o", str(TL)])
    print(TL)


SHAPES = [("L6  128->256 @80", 128, 256, 80), ("L8  256->512 @40", 256, 512, 40), ("L10 512->1024 @20", 512, 1024, 20)]


def main():
    import ctypes as C
    import numpy as np
    import torch
    from yolov3_amd import _lib, ops

    L = _lib.lib()
    L.y3_debug_timeline.argtypes = [C.c_void_p]
    L.y3_debug_timeline.restype = None
    dev = torch.device("cuda:0")
    n = int(os.environ.get("TL_BATCH", "32"))
    for name, cin, cout, H in SHAPES:
        x = ops.View.alloc(n, H, H, cin, torch.float16, dev)
        x.buf.normal_()
        y = ops.View.alloc(n, H, H, cout, torch.float16, dev)
        res = ops.View.alloc(n, H, H, cout, torch.float16, dev)
        res.buf.normal_()
        filt = ops.pack_filter(torch.randn(cout, cin, 3, 3, device=dev) * 0.05, cout, cin, torch.float16)
        bias = torch.zeros(cout, device=dev)
        for half in (0, 1):
            ops.tune_set("v10_half", half)
            tl = torch.zeros(1 << 18, dtype=torch.int64, device=dev)
            L.y3_debug_timeline(None)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(5):
                ops.conv2d(x, filt, bias, y, 3, 1, True, residual=res)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(20):
                ops.conv2d(x, filt, bias, y, 3, 1, True, residual=res)
            e1.record()
            torch.cuda.synchronize()
            ev_us = e0.elapsed_time(e1) * 50.0
            L.y3_debug_timeline(tl.data_ptr())
            ops.conv2d(x, filt, bias, y, 3, 1, True, residual=res)
            torch.cuda.synchronize()
            L.y3_debug_timeline(None)
            t = tl.cpu().numpy().reshape(-1, 64).astype(np.float64) / 100.0   # 100 MHz -> us
            t = t[t[:, 0] != 0]
            t0 = t[:, 0].min()
            ends = np.where(t > 0, t, 0).max(axis=1)
            nt = ((t[:, 1:] > 0).sum(axis=1) // 3)
            med = lambda a: float(np.median(a)) if len(a) else float("nan")
            setup, kloop, epi = [], [], []
            for b in range(len(t)):
                prev = t[b, 0]
                for k in range(int(nt[b])):
                    s0, s1, s2 = t[b, 1 + 3 * k], t[b, 2 + 3 * k], t[b, 3 + 3 * k]
                    setup.append(s0 - prev); kloop.append(s1 - s0); epi.append(s2 - s1)
                    prev = s2
            print(f"{name} batch {n} {'half' if half else 'full'}: event {ev_us:6.1f} us, {len(t)} blocks, span {ends.max() - t0:6.1f} us, start ramp p90 {np.percentile(t[:, 0] - t0, 90):4.1f} us, "
                  f"tiles/block {nt.min()}-{nt.max()} | per tile (median us): set-up {med(setup):5.2f}  K loop {med(kloop):6.2f}  epilogue {med(epi):5.2f} | block busy median {med(ends - t[:, 0]):6.1f} us", flush=True)
        ops.tune_reset()


if __name__ == "__main__":
    if "--build" in sys.argv:
        build_tl()
        sys.exit(0)
    if os.environ.get("Y3_LIB") != str(TL):
        assert TL.exists(), "build first: python tools/v10_timeline.py --build"
        os.execve(sys.executable, [sys.executable, __file__], dict(os.environ, Y3_LIB=str(TL)))
    main()
