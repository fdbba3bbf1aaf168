"""Stride-2 data-gradient launches of the batch-64 train step (y3_conv2d_dgrad_s2, filter packing included), knob arms interleaved on one box.

    python tools/dgrad_lab.py [--arms "conv_strip=0;conv_strip=1"]"""
import argparse
import math
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from yolov3_amd import ops  # noqa: E402

SHAPES = {"L1 32->64 s2 @640": (64, 640, 640, 32, 64), "L3 64->128 s2 @320": (64, 320, 320, 64, 128)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arms", default="conv_strip=0;conv_strip=1")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=6)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    arms = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in arm.split(",") if kv) for arm in a.arms.split(";")]
    print(f"{'shape':22s} {'arm':22s} {'variant':12s} {'med us':>9s} {'min us':>9s} {'TB/s':>6s} {'TF/s':>7s}")
    for name, (n, h, w, cin, cout) in SHAPES.items():
        g = torch.Generator(device=dev).manual_seed(3)
        gv = ops.View.alloc(n, h // 2, w // 2, cout, torch.float16, dev)
        gv.buf.normal_(generator=g)
        gx = ops.View.alloc(n, h, w, cin, torch.float16, dev)
        wt = torch.randn(cout, cin, 3, 3, device=dev, generator=g) / math.sqrt(cin * 9)
        times, var, outs = [[] for _ in arms], [""] * len(arms), []
        for rnd in range(a.rounds + 1):
            for i, arm in enumerate(arms):
                ops.tune_reset()
                for kk, vv in arm.items():
                    ops.tune_set(kk, vv)
                ops.conv2d_dgrad_s2(wt, gv, gx, accumulate=False)
                var[i] = ops.last_conv_variant()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.reps):
                    ops.conv2d_dgrad_s2(wt, gv, gx, accumulate=False)
                e1.record()
                torch.cuda.synchronize()
                if rnd:
                    times[i].append(e0.elapsed_time(e1) * 1e3 / a.reps)
                elif len(outs) < len(arms):
                    outs.append(gx.as_nhwc().clone())
        byt = (n * h * w * cin + n * (h // 2) * (w // 2) * cout) * 2
        flop = 2.0 * n * (h // 2) * (w // 2) * cout * cin * 9
        for i, arm in enumerate(arms):
            med, mn = statistics.median(times[i]), min(times[i])
            same = torch.equal(outs[i], outs[0])
            print(f"{name:22s} {str(arm):22s} {var[i]:12s} {med:9.1f} {mn:9.1f} {byt / med / 1e6:6.2f} {flop / med / 1e6:7.1f}   bit-identical to arm 0: {same}")
        ops.tune_reset()
        del gv, gx


if __name__ == "__main__":
    main()
