"""Per-wave tick sums of the phases of a stem_pair tile (instrumented build, -DY3_TIMELINE).  GPU box only:  python tools/stem_probe.py"""
import math
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
TL = ROOT / "yolov3_amd" / "lib" / "libyolov3_hip_stl.so"
NAMES = ["stash (waits for the prefetched image)", "barrier", "layer 0 (MFMA + SiLU -> LDS)", "barrier", "prefetch issue + layer 1 MFMAs", "SiLU + transpose + stores", "barrier"]


def build_tl():
    from yolov3_amd import build as B
    cc = B.hipcc()
    objs = []
    for src, extra in B.SOURCES:
        s = B.CSRC / src
        o = B.OBJ_DIR / (s.stem + ("_stl.o" if src == "stem.hip" else ".o"))
        if src == "stem.hip":
            subprocess.check_call([cc, *B.COMMON, *extra, "-DY3_TIMELINE", "-x", "hip", "-c", str(s), "-o", str(o)])
        objs.append(str(o))
    subprocess.check_call([cc, "-shared", "-fPIC", f"--offload-arch={B.ARCH}", *objs, "-o", str(TL)])


def main():
    import ctypes as C
    import torch
    from yolov3_amd import _lib, ops

    L = _lib.lib()
    L.y3_debug_pair_timeline.argtypes = [C.c_void_p]
    L.y3_debug_pair_timeline.restype = None
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    dt, bs, size = torch.float16, 32, 640
    x = torch.rand(bs, 3, size, size, generator=g).to(dev).to(dt)
    f0 = ops.pack_filter_stem((torch.randn(32, 3, 3, 3, generator=g) / math.sqrt(27)).to(dev), 32, dt)
    f1 = ops.pack_filter((torch.randn(64, 32, 3, 3, generator=g) / math.sqrt(288)).to(dev), 64, 32, dt)
    b0, b1 = torch.zeros(32, device=dev), torch.zeros(64, device=dev)
    yv = ops.View.alloc(bs, size // 2, size // 2, 64, dt, dev)
    tiles = bs * (size // 2 // 4) * (size // 2 // 32)
    for occ in ("2",):
        tl = torch.zeros(64 * 4 * 8, dtype=torch.int64, device=dev)
        for _ in range(5):
            ops.stem_pair(x, f0, b0, True, f1, b1, True, yv, 1.0)
        torch.cuda.synchronize()
        L.y3_debug_pair_timeline(tl.data_ptr())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.stem_pair(x, f0, b0, True, f1, b1, True, yv, 1.0)
        e1.record()
        torch.cuda.synchronize()
        L.y3_debug_pair_timeline(None)
        us = e0.elapsed_time(e1) * 1e3
        per_block = tiles / (256 * int(occ))
        t = tl.view(64, 4, 8).cpu().double()[:, :, :7].mean(dim=(0, 1)) / per_block
        tot = t.sum().item()
        print(f"launch {us:.1f} us, {per_block:.1f} tiles per block, {us / per_block:.2f} us per tile and block; ticks per tile {tot:.0f}")
        for n, v in zip(NAMES, t.tolist()):
            print(f"    {n:42s} {v:8.0f}  {100 * v / tot:5.1f} %")


if __name__ == "__main__":
    if os.environ.get("Y3_LIB") != str(TL):
        if not TL.exists() or TL.stat().st_mtime < max(f.stat().st_mtime for f in (ROOT / "yolov3_amd" / "csrc").iterdir()):
            build_tl()
        os.environ["Y3_LIB"] = str(TL)
        os.execv(sys.executable, [sys.executable, *sys.argv])
    main()
