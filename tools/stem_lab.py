"""Time y3_stem_pair_fwd alone (yolov3 layers 0 + 1, 640x640 batch 32) for both builds: python tools/stem_lab.py [--batch 32] [--size 640]"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolov3_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    dt = torch.float16
    x = torch.rand(a.batch, 3, a.size, a.size, generator=g).to(dev).to(dt)
    w0 = torch.randn(32, 3, 3, 3, generator=g) / math.sqrt(27)
    w1 = torch.randn(64, 32, 3, 3, generator=g) / math.sqrt(288)
    b0, b1 = torch.zeros(32, device=dev), torch.zeros(64, device=dev)
    f0 = ops.pack_filter_stem(w0.to(dev), 32, dt)
    f1 = ops.pack_filter(w1.to(dev), 64, 32, dt)
    ho = (a.size - 1) // 2 + 1
    yv = ops.View.alloc(a.batch, ho, ho, 64, dt, dev)
    for _ in range(200):   # clocks settle
        ops.stem_pair(x, f0, b0, True, f1, b1, True, yv, 1.0)
    torch.cuda.synchronize()
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            ops.stem_pair(x, f0, b0, True, f1, b1, True, yv, 1.0)
        e1.record()
        torch.cuda.synchronize()
        print(f"stem_pair: {e0.elapsed_time(e1) / a.reps * 1e3:.1f} us", flush=True)


if __name__ == "__main__":
    main()
