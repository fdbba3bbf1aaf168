"""Does splitting a batch over HIP streams fill the CUs that one launch of 200 / 400 / 800 tiles leaves idle?

    python tools/stream_lab.py [--chain 12] [--rounds 5]

For each BASELINE 3x3 layer shape: the batch of 32 as ONE launch chain on one stream vs the same images as S sub-batches
(S = 2, 4, 8), each sub-batch a chain of `--chain` back-to-back launches on its own stream (the launches of one chain are
dependent through the stream, chains are independent: what a forward of S sub-batches looks like to the GPU).  Reports the
wall time per equivalent full-batch launch and TFLOP/s.  Arms: round-1 kernels (no workspace) and v7 with whole tiles."""
import argparse
import math
import os
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

SHAPES = [
    ("L6.cv2 128->256 @80", 80, 80, 128, 256, 3, 1),
    ("L8.cv2 256->512 @40", 40, 40, 256, 512, 3, 1),
    ("L10.cv2 512->1024 @20", 20, 20, 512, 1024, 3, 1),
    ("L8.cv1 512->256 @40 1x1", 40, 40, 512, 256, 1, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chain", type=int, default=12)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32)
    args = ap.parse_args()
    from yolov3_amd import ops

    dev = torch.device("cuda:0")
    dtype = torch.float16
    g = torch.Generator().manual_seed(0)
    streams = [torch.cuda.Stream(device=dev) for _ in range(8)]
    wss = [ops.conv_workspace(dev) for _ in range(8)]
    print(f"{'shape':26s} {'arm':8s} {'streams':>7s} {'us / full-batch launch':>24s} {'TF/s':>8s}")
    for name, h, w, cin, cout, k, s in SHAPES:
        n = args.batch
        x = torch.randn(n, cin, h, w, generator=g)
        wt = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
        filt = ops.pack_filter(wt.to(dev), cout, cin, dtype)
        bias = torch.randn(cout, generator=g).to(dev)
        flops = 2.0 * n * h * w * cout * cin * k * k / (s * s)
        for arm in ("nows", "v7whole"):
            if arm == "v7whole":
                ops.tune_set("v7_grid", -1); ops.tune_set("conv_v10", 0)
            for S in (1, 2, 4, 8):
                nb = n // S
                xs, ys = [], []
                for i in range(S):
                    xv = ops.View.alloc(nb, h, w, cin, dtype, dev)
                    ops.nchw_to_nhwc(x[i * nb:(i + 1) * nb].to(dev), xv)
                    xs.append(xv)
                    ys.append(ops.View.alloc(nb, h // s, w // s, cout, dtype, dev))
                torch.cuda.synchronize()
                ts = []
                for rnd in range(args.rounds + 1):
                    e0 = torch.cuda.Event(enable_timing=True)
                    e1 = torch.cuda.Event(enable_timing=True)
                    e0.record()
                    ends = []
                    for i in range(S):
                        st = streams[i]
                        st.wait_event(e0)
                        with torch.cuda.stream(st):
                            for _ in range(args.chain):
                                ops.conv2d(xs[i], filt, bias, ys[i], k, s, True, None, workspace=wss[i] if arm == "v7whole" else None)
                            ev = torch.cuda.Event()
                            ev.record()
                            ends.append(ev)
                    for ev in ends:
                        torch.cuda.current_stream().wait_event(ev)
                    e1.record()
                    torch.cuda.synchronize()
                    if rnd:
                        ts.append(e0.elapsed_time(e1) * 1e3 / args.chain)
                med = statistics.median(ts)
                print(f"{name:26s} {arm:8s} {S:7d} {med:24.1f} {flops / med / 1e6:8.1f}")
                sys.stdout.flush()
            ops.tune_reset()


if __name__ == "__main__":
    main()
