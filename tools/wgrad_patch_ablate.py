"""What the padded-position filter-gradient kernel (csrc/wgrad_patch.h) costs without one of its parts: launch time of the ablated instantiations (-DY3_ABLATE build of
train.hip, Y3_WP_ABL=<n> read per launch) on the batch-64 80x80 / 40x40 / 20x20 layers, interleaved rounds on one box.  Results of the ablated arms are garbage by
construction; only the time means something.  Build here (python tools/wgrad_patch_ablate.py --build), run on the GPU box with Y3_LIB pointing at the lab library,
delete it afterwards (it must not ship)."""
import os
import statistics
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
ABL = ROOT / "yolov3_amd" / "lib" / "libyolov3_hip_wpabl.so"
ARMS = [(0, "full kernel"), (1, "no requests in the loop"), (2, "no fragment reads in the loop"), (3, "no barrier / counted wait"), (4, "no request sources, no requests"),
        (5, "MFMAs only (2 + 3 + 4)"), (6, "no slab stores"), (7, "no MFMAs")]


def build():
    from yolov3_amd import build as B
    B.build(verbose=False)
    cc = B.hipcc()
    objs = []
    for src, extra in B.SOURCES:
        s = B.CSRC / src
        o = B.OBJ_DIR / (s.stem + ("_wpabl.o" if src == "train.hip" else ".o"))
        if src == "train.hip":
            subprocess.check_call([cc, *B.COMMON, *extra, "-DY3_ABLATE", "-x", "hip", "-c", str(s), "-o", str(o)])
        objs.append(str(o))
    subprocess.check_call([cc, "-shared", "-fPIC", f"--offload-arch={B.ARCH}", *objs, "-o", str(ABL)])
    print(ABL)


def main():
    import torch
    from yolov3_amd import ops

    dev = torch.device("cuda:0")
    rounds, reps = 3, 20
    for name, n, h, w, cin, cout in [("L6 128->256 @80x80", 64, 80, 80, 128, 256), ("L8 256->512 @40x40", 64, 40, 40, 256, 512), ("L10 512->1024 @20x20", 64, 20, 20, 512, 1024)]:
        g = torch.Generator(device=dev).manual_seed(3)
        xv = ops.View.alloc(n, h, w, cin, torch.float16, dev)
        xv.buf.normal_(generator=g)
        gv = ops.View.alloc(n, h, w, cout, torch.float16, dev)
        gv.buf.normal_(generator=g)
        times = {a: [] for a, _ in ARMS}
        for rnd in range(rounds + 1):
            for a, _ in ARMS:
                os.environ["Y3_WP_ABL"] = str(a)
                torch.cuda.synchronize()
                time.sleep(0.05)
                for _ in range(10):   # (clocks settle to the arm's own power level before the timed launches)
                    ops.conv2d_wgrad(xv, gv, 3, 1, cout, cin)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    ops.conv2d_wgrad(xv, gv, 3, 1, cout, cin)
                e1.record()
                torch.cuda.synchronize()
                if rnd:
                    times[a].append(e0.elapsed_time(e1) * 1e3 / reps)
        print(f"{name} batch {n} (kernel + slice sum per launch)")
        base = statistics.median(times[0])
        for a, label in ARMS:
            med = statistics.median(times[a])
            print(f"    ABL {a} {label:36s} {med:8.1f} us  ({med - base:+7.1f} us)")
        sys.stdout.flush()


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
    else:
        main()
