"""What a K-step of the one-wave-per-SIMD 3x3 kernel (conv_v9.h) costs without one of its parts: launch time of the ablated instantiations
(-DY3_ABLATE build, Y3_V9_ABL=<n> read per launch) on the BASELINE 20x20 / 40x40 layers, interleaved rounds on one box.  Results of the
ablated arms are garbage by construction; only the time means something.  Build here (python tools/v9_ablate.py --build), run on the GPU box."""
import math
import os
import statistics
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
ABL = ROOT / "yolov3_amd" / "lib" / "libyolov3_hip_abl.so"
ARMS = [(0, "full kernel"), (1, "no filter requests"), (2, "no patch requests"), (3, "no pixel-fragment reads"), (4, "no filter-fragment reads"),
        (5, "no fragment reads"), (6, "no MFMAs"), (7, "MFMAs only"), (8, "no epilogue")]
# (arms 9-13 of profiles/r03_v9_ablation.txt -- register-load filters, other interleavings -- came from lab hooks that were removed with the round-3 clean-up)


def build():
    from yolov3_amd import build as B
    B.build(verbose=False)
    cc = B.hipcc()
    objs = []
    for src, extra in B.SOURCES:
        s = B.CSRC / src
        o = B.OBJ_DIR / (s.stem + ("_abl.o" if src == "conv.hip" else ".o"))
        if src == "conv.hip":
            subprocess.check_call([cc, *B.COMMON, *extra, "-DY3_ABLATE", "-x", "hip", "-c", str(s), "-o", str(o)])
        objs.append(str(o))
    subprocess.check_call([cc, "-shared", "-fPIC", f"--offload-arch={B.ARCH}", *objs, "-o", str(ABL)])
    print(ABL)


def main():
    import torch
    from yolov3_amd import ops

    dev = torch.device("cuda:0")
    ws = ops.conv_workspace(dev)
    ops.tune_set("conv_v9", 2)
    g = torch.Generator().manual_seed(0)
    rounds, reps = 5, 10
    for name, n, h, w, cin, cout in [("L10 512->1024 @20x20", 32, 20, 20, 512, 1024), ("L8 256->512 @40x40", 32, 40, 40, 256, 512)]:
        xv = ops.View.alloc(n, h, w, cin, torch.float16, dev)
        ops.nchw_to_nhwc(torch.randn(n, cin, h, w, generator=g).to(dev), xv)
        filt = ops.pack_filter((torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)).to(dev), cout, cin, torch.float16)
        bias = torch.randn(cout, generator=g).to(dev)
        yv = ops.View.alloc(n, h, w, cout, torch.float16, dev)
        times = {a: [] for a, _ in ARMS}
        for rnd in range(rounds + 1):
            for a, _ in ARMS:
                os.environ["Y3_V9_ABL"] = str(a)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    ops.conv2d(xv, filt, bias, yv, 3, 1, True, None, workspace=ws)
                e1.record()
                torch.cuda.synchronize()
                if rnd:
                    times[a].append(e0.elapsed_time(e1) * 1e3 / reps)
        tiles = math.ceil(n * h * w / 200) * (cout // 256)
        steps = 9 * cin // 32 * math.ceil(tiles / 256)
        print(f"{name}: {tiles} tiles, {steps} K-steps per block")
        base = statistics.median(times[0])
        for a, label in ARMS:
            med = statistics.median(times[a])
            print(f"    ABL {a} {label:26s} {med:8.1f} us  ({med - base:+7.1f} us, {(med - base) * 1e3 / steps:+7.0f} ns per K-step)")


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
        sys.exit(0)
    if os.environ.get("Y3_LIB") != str(ABL):
        assert ABL.exists(), "build first: python tools/v9_ablate.py --build"
        os.environ["Y3_LIB"] = str(ABL)
        os.execv(sys.executable, [sys.executable, *sys.argv])
    main()
