"""Per-wave cycle sums of the four intervals of a v7 K-step (instrumented build, -DY3_TIMELINE): MEM issue, vmcnt wait,
barrier after MEM, MMA issue, barrier after MMA.  GPU box only:  python tools/v7_probe.py"""
import math
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


def main():
    import ctypes as C
    import torch
    from yolov3_amd import _lib, ops

    L = _lib.lib()
    L.y3_debug_timeline.argtypes = [C.c_void_p]
    L.y3_debug_timeline.restype = None
    dev = torch.device("cuda:0")
    ws = ops.conv_workspace(dev)
    g = torch.Generator().manual_seed(0)
    for name, n, h, w, cin, cout in [("L10 512->1024 @20", 32, 20, 20, 512, 1024), ("L8 256->512 @40", 32, 40, 40, 256, 512), ("L6 128->256 @80", 32, 80, 80, 128, 256)]:
        xv = ops.View.alloc(n, h, w, cin, torch.float16, dev)
        ops.nchw_to_nhwc(torch.randn(n, cin, h, w, generator=g).to(dev), xv)
        filt = ops.pack_filter((torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)).to(dev), cout, cin, torch.float16)
        bias = torch.randn(cout, generator=g).to(dev)
        yv = ops.View.alloc(n, h, w, cout, torch.float16, dev)
        for sched in ("0",):
            ops.tune_set("v7_grid", -1); ops.tune_set("conv_v10", 0)
            tl = torch.zeros(64 * 8 * 8, dtype=torch.int64, device=dev)
            for _ in range(3):
                ops.conv2d(xv, filt, bias, yv, 3, 1, True, None, workspace=ws)
            torch.cuda.synchronize()
            L.y3_debug_timeline(tl.data_ptr())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.conv2d(xv, filt, bias, yv, 3, 1, True, None, workspace=ws)
            e1.record()
            torch.cuda.synchronize()
            L.y3_debug_timeline(None)
            t = tl.view(64, 8, 8).cpu().double()
            tiles = math.ceil(n * h * w / 256) * (cout // 256)
            per_block_steps = 9 * (cin // 32) * max(1, round(tiles / min(tiles, 256)))
            print(f"{name} SCHED {sched}: launch {e0.elapsed_time(e1) * 1e3:.1f} us, ~{per_block_steps} K-steps per block; cycles per K-step (100 MHz ticks x clk ratio unknown: raw s_memtime ticks)")
            for half, wvs in (("leading waves 0-3", [0, 1, 2, 3]), ("trailing waves 4-7", [4, 5, 6, 7])):
                m = t[:32, wvs, :5].mean(dim=(0, 1)) / per_block_steps
                print(f"    {half}: MEM issue {m[0]:.0f}  vmcnt wait {m[1]:.0f}  barrier(MEM) {m[2]:.0f}  MMA issue {m[3]:.0f}  barrier(MMA) {m[4]:.0f}  sum {m.sum():.0f}")
        ops.tune_reset()


if __name__ == "__main__":
    if os.environ.get("Y3_LIB") != str(TL):
        if not TL.exists() or TL.stat().st_mtime < max(f.stat().st_mtime for f in (ROOT / "yolov3_amd" / "csrc").iterdir()):
            build_tl()
        os.environ["Y3_LIB"] = str(TL)
        os.execv(sys.executable, [sys.executable, *sys.argv])
    main()
