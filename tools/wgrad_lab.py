"""Filter-gradient launches of the batch-64 train step, timed per shape with knob arms interleaved (same process, same box).

    python tools/wgrad_lab.py [--arms "wgrad_strip=0;wgrad_strip=1"] [--shapes L1,L2cv2,...]

Prints median / min microseconds per arm (y3_conv2d_wgrad = the tile kernel + the slice sum), the algorithmic HBM rate (x + du read once) and TFLOP/s."""
import argparse
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from yolov3_amd import ops  # noqa: E402

SHAPES = {
    # name: (n, h, w, cin, cout, k, s)
    "L1": (64, 640, 640, 32, 64, 3, 2),
    "L2cv1": (64, 320, 320, 64, 32, 1, 1),
    "L2cv2": (64, 320, 320, 32, 64, 3, 1),
    "L3": (64, 320, 320, 64, 128, 3, 2),
    "L4cv1": (64, 160, 160, 128, 64, 1, 1),
    "L4cv2": (64, 160, 160, 64, 128, 3, 1),
    "L5": (64, 160, 160, 128, 256, 3, 2),
    "L6cv1": (64, 80, 80, 256, 128, 1, 1),
    "L6cv2": (64, 80, 80, 128, 256, 3, 1),
    "L8cv2": (64, 40, 40, 256, 512, 3, 1),
    "L10cv2": (64, 20, 20, 512, 1024, 3, 1),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arms", default="wgrad_strip=0;wgrad_strip=1")
    ap.add_argument("--shapes", default="L1,L2cv1,L2cv2,L4cv1,L3,L4cv2")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=6)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    arms = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in arm.split(",") if kv) for arm in a.arms.split(";")]
    print(f"{'shape':8s} {'arm':24s} {'tile':>4s} {'slices':>6s} {'med us':>9s} {'min us':>9s} {'TB/s':>6s} {'TF/s':>7s}")
    for name in a.shapes.split(","):
        n, h, w, cin, cout, k, s = SHAPES[name]
        ho, wo = (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1
        g = torch.Generator(device=dev).manual_seed(3)
        xv = ops.View.alloc(n, h, w, cin, torch.float16, dev)
        xv.buf.normal_(generator=g)
        gv = ops.View.alloc(n, ho, wo, cout, torch.float16, dev)
        gv.buf.normal_(generator=g)
        times = [[] for _ in arms]
        outs = []
        for rnd in range(a.rounds + 1):
            for i, arm in enumerate(arms):
                ops.tune_reset()
                for kk, vv in arm.items():
                    ops.tune_set(kk, vv)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                dw, _ = ops.conv2d_wgrad(xv, gv, k, s, cout, cin)
                e0.record()
                for _ in range(a.reps):
                    dw, _ = ops.conv2d_wgrad(xv, gv, k, s, cout, cin)
                e1.record()
                torch.cuda.synchronize()
                if rnd:
                    times[i].append(e0.elapsed_time(e1) * 1e3 / a.reps)
                elif len(outs) < len(arms):
                    outs.append(dw.clone())
        tile, slices, _ = ops.conv2d_wgrad_plan(xv, cout, k, s)
        byt = (n * h * w * cin + n * ho * wo * cout) * 2
        flop = 2.0 * n * ho * wo * cout * cin * k * k
        for i, arm in enumerate(arms):
            med, mn = statistics.median(times[i]), min(times[i])
            d = (outs[i] - outs[0]).abs().max().item()
            print(f"{name:8s} {str(arm):24s} {tile:4d} {slices:6d} {med:9.1f} {mn:9.1f} {byt / med / 1e6:6.2f} {flop / med / 1e6:7.1f}   max|d vs arm 0| {d:.3g}")
        ops.tune_reset()
        del xv, gv


if __name__ == "__main__":
    main()
