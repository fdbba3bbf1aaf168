"""What the persistent one-wave-per-SIMD 3x3 kernel (conv_v10.h) costs without one of its parts: launch time of the ablated instantiations
(-DY3_ABLATE build, Y3_V10_ABL=<n> read per launch) on the BASELINE 80x80 / 40x40 / 20x20 layers, interleaved rounds on one box.  Results of the
ablated arms are garbage by construction; only the time means something.  Build here (python tools/v10_ablate.py --build; f16 only would do but the
lab object is the whole conv.hip), run on the GPU box, delete the lab library afterwards (it must not ship)."""
import math
import time
import os
import statistics
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
ABL = ROOT / "yolov3_amd" / "lib" / "libyolov3_hip_abl.so"
ARMS = [(0, "full kernel"), (1, "no epilogue"), (2, "epilogue, stores + residual dropped"), (4, "no MFMAs"), (5, "MFMAs + epilogue only"),
        (6, "no filter loads"), (7, "no pixel-fragment reads"), (8, "no patch requests")]


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
    g = torch.Generator().manual_seed(0)
    rounds, reps = 2, 50
    batch = int(os.environ.get("ABL_BATCH", "32"))
    for name, n, h, w, cin, cout in [("L6 128->256 @80x80", batch, 80, 80, 128, 256), ("L8 256->512 @40x40", batch, 40, 40, 256, 512), ("L10 512->1024 @20x20", batch, 20, 20, 512, 1024)]:
        xv = ops.View.alloc(n, h, w, cin, torch.float16, dev)
        ops.nchw_to_nhwc(torch.randn(n, cin, h, w, generator=g).to(dev), xv)
        filt = ops.pack_filter((torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)).to(dev), cout, cin, torch.float16)
        bias = torch.randn(cout, generator=g).to(dev)
        yv = ops.View.alloc(n, h, w, cout, torch.float16, dev)
        rv = ops.View.alloc(n, h, w, cout, torch.float16, dev)
        rv.buf.normal_()
        times = {a: [] for a, _ in ARMS}
        for rnd in range(rounds + 1):
            for a, _ in ARMS:
                os.environ["Y3_V10_ABL"] = str(a)
                torch.cuda.synchronize()
                time.sleep(0.05)
                for _ in range(30):   # (clocks settle to the arm's own power level before the timed launches: an arm timed right after a denser one inherits its throttled clock)
                    ops.conv2d(xv, filt, bias, yv, 3, 1, True, rv, workspace=ws)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    ops.conv2d(xv, filt, bias, yv, 3, 1, True, rv, workspace=ws)
                e1.record()
                torch.cuda.synchronize()
                if rnd:
                    times[a].append(e0.elapsed_time(e1) * 1e3 / reps)
        cb = math.ceil(n * h * w / 32)
        per = cb / (256 // (cout // 256))
        print(f"{name} batch {n}: {per:.2f} column blocks per block, {9 * cin // 32} K-steps per tile")
        base = statistics.median(times[0])
        for a, label in ARMS:
            med = statistics.median(times[a])
            print(f"    ABL {a} {label:38s} {med:8.1f} us  ({med - base:+7.1f} us)")
        sys.stdout.flush()


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
        sys.exit(0)
    if os.environ.get("Y3_LIB") != str(ABL):
        assert ABL.exists(), "build first: python tools/v10_ablate.py --build"
        os.environ["Y3_LIB"] = str(ABL)
        os.execv(sys.executable, [sys.executable, *sys.argv])
    main()
