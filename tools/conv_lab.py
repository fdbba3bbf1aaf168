"""A/B timing of conv kernel variants on the BASELINE layer shapes (one MI355X, one process, interleaved rounds).

    python tools/conv_lab.py [--rounds 7] [--reps 20] [--batch 32] [--only 3x3]

Arms per shape (run-time knobs, y3_tune_set): "auto" = the dispatcher's choice with a workspace (v10 where eligible), "no v10" = knob
conv_v10 = 0 (v6 / v3), "nows" = y3_conv2d_fwd without a workspace and without v10 (round-1 kernels).
Prints median / min microseconds per launch and TFLOP/s.  Inputs are random (DVFS: never time on zeros)."""
import argparse
import math
import os
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

SHAPES = [
    # name, h, w, cin, cout, k, s, residual
    ("L4.cv2 64->128 @160", 160, 160, 64, 128, 3, 1, True),
    ("L6.cv2 128->256 @80", 80, 80, 128, 256, 3, 1, True),
    ("L8.cv2 256->512 @40", 40, 40, 256, 512, 3, 1, True),
    ("L10.cv2 512->1024 @20", 20, 20, 512, 1024, 3, 1, True),
    ("L13 512->1024 @20 nores", 20, 20, 512, 1024, 3, 1, False),
    ("L7 256->512 s2 @80", 80, 80, 256, 512, 3, 2, False),
    ("L9 512->1024 s2 @40", 40, 40, 512, 1024, 3, 2, False),
    ("L6.cv1 256->128 @80", 80, 80, 256, 128, 1, 1, False),
    ("L8.cv1 512->256 @40", 40, 40, 512, 256, 1, 1, False),
    ("L10.cv1 1024->512 @20", 20, 20, 1024, 512, 1, 1, False),
    ("L12 1024->512 @20 (head)", 20, 20, 1024, 512, 1, 1, False),
    ("L16 512->256 @20", 20, 20, 512, 256, 1, 1, False),
    ("L19.cv1 768->256 @40", 40, 40, 768, 256, 1, 1, False),
    ("L26.cv1 384->128 @80", 80, 80, 384, 128, 1, 1, False),
    ("L4.cv1 128->64 @160", 160, 160, 128, 64, 1, 1, False),
    ("L2.cv1 64->32 @320", 320, 320, 64, 32, 1, 1, False),
    ("L2.cv2 32->64 @320", 320, 320, 32, 64, 3, 1, True),
    ("L3 64->128 s2 @320", 320, 320, 64, 128, 3, 2, False),
    ("L5 128->256 s2 @160", 160, 160, 128, 256, 3, 2, False),
    # training-mode launches of the small-channel 3x3 layers (no residual in the conv; --noact): forward and data gradient (= the conv with the channels swapped)
    ("T L2.cv2 32->64 @320", 320, 320, 32, 64, 3, 1, False),
    ("T L2.cv2 dgrad 64->32 @320", 320, 320, 64, 32, 3, 1, False),
    ("T L4.cv2 64->128 @160", 160, 160, 64, 128, 3, 1, False),
    ("T L4.cv2 dgrad 128->64 @160", 160, 160, 128, 64, 3, 1, False),
    ("T L3 64->128 s2 @320", 320, 320, 64, 128, 3, 2, False),
    # data gradients of the Bottleneck.cv1 layers (1x1, the shortcut's gradient as the residual operand)
    ("T L6.cv1 dgrad 128->256 @80", 80, 80, 128, 256, 1, 1, True),
    ("T L4.cv1 dgrad 64->128 @160", 160, 160, 64, 128, 1, 1, True),
    ("T L2.cv1 64->32 @320", 320, 320, 64, 32, 1, 1, False),
    ("T L2.cv1 dgrad 32->64 @320", 320, 320, 32, 64, 1, 1, True),
    ("head 256->256 @80", 80, 80, 256, 256, 1, 1, False),
    ("T L26.cv1 dgrad 128->384 @80", 80, 80, 128, 384, 1, 1, False),
    ("T L8.cv1 dgrad 256->512 @40", 40, 40, 256, 512, 1, 1, True),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--only", default="")
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--arms", default="", help='custom arms instead of the default three: "knob=value,knob=value;knob=value" (each with a workspace)')
    ap.add_argument("--noact", action="store_true", help="no activation (the training-mode launches)")
    ap.add_argument("--train-forms", action="store_true", help="the three forms a 3x3 layer runs in the train step, interleaved: plain (no activation), forward with the "
                    "BatchNorm statistics rows from the epilogue, data-gradient form (accumulating into the output through the residual port)")
    ap.add_argument("--sweep", action="store_true", help="also time the forced tile variants (knob conv = 4 / 5 / 6 / 15) and the forced v10 wave-tile widths")
    args = ap.parse_args()
    from yolov3_amd import ops

    dev = torch.device("cuda:0")
    dtype = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    ws = ops.conv_workspace(dev)
    g = torch.Generator().manual_seed(0)
    print(f"{'shape':28s} {'arm':10s} {'variant':18s} {'med us':>9s} {'min us':>9s} {'TF/s(med)':>10s} {'TF/s(min)':>10s}")
    for name, h, w, cin, cout, k, s, res in SHAPES:
        if args.only and not any(o in name for o in args.only.split(",")):
            continue
        n = args.batch
        ho, wo = (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1
        x = torch.randn(n, cin, h, w, generator=g)
        xv = ops.View.alloc(n, h, w, cin, dtype, dev)
        ops.nchw_to_nhwc(x.to(dev), xv)
        wt = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
        filt = ops.pack_filter(wt.to(dev), cout, cin, dtype)
        bias = torch.randn(cout, generator=g).to(dev)
        yv = ops.View.alloc(n, ho, wo, cout, dtype, dev)
        rv = None
        if res:
            rv = ops.View.alloc(n, ho, wo, cout, dtype, dev)
            rv.buf.copy_(torch.randn(rv.buf.numel(), generator=g).to(dev).to(dtype))
        flops = 2.0 * n * ho * wo * cout * cin * k * k
        arms = [("nows", None, {"conv_v10": 0}), ("no v10", ws, {"conv_v10": 0}), ("auto", ws, {})]
        if args.arms:
            arms = [(a, ws, dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split(",") if kv)) for a in args.arms.split(";")]
        if args.sweep:
            arms += [(f"conv={v}", None, {"conv": v, "conv_v10": 0}) for v in (4, 6, 15)]
            if k == 3 and s == 1 and cout % 256 == 0:
                arms += [(f"v10 mp{m}", ws, {"conv_v10": 2, "v10_mp": m}) for m in (6, 7, 8)]
        if args.train_forms:   # what the statistics rows and the accumulating residual cost a launch of the same shape (same kernel source, same box, interleaved)
            rows = ops.conv2d_stats_rows(xv, yv, k, s, workspace=ws)
            sbuf = torch.empty(rows * 2 * cout, dtype=torch.float32, device=dev)
            acc = ops.View.alloc(n, ho, wo, cout, dtype, dev)
            acc.buf.zero_()
            zb = torch.zeros(cout, device=dev)
            forms = [("plain", lambda: ops.conv2d(xv, filt, zb, yv, k, s, False, None, workspace=ws)),
                     ("fwd+stats", lambda: ops.conv2d_stats(xv, filt, zb, yv, k, s, sbuf, rows, workspace=ws)),
                     ("dgrad-acc", lambda: ops.conv2d(xv, filt, zb, acc, k, s, False, acc, workspace=ws))]
            tf = {f[0]: [] for f in forms}
            for rnd in range(args.rounds + 1):
                for fname, fn in forms:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    fn()
                    e0.record()
                    for _ in range(args.reps):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    if rnd:
                        tf[fname].append(e0.elapsed_time(e1) * 1e3 / args.reps)
            for fname, _ in forms:
                med, mn = statistics.median(tf[fname]), min(tf[fname])
                print(f"{name:28s} {fname:10s} {'':18s} {med:9.1f} {mn:9.1f} {flops / med / 1e6:10.1f} {flops / mn / 1e6:10.1f}")
            sys.stdout.flush()
            continue
        times = {a[0]: [] for a in arms}
        outs = {}
        for rnd in range(args.rounds + 1):
            for arm, wsp, env in arms:
                for kk, vv in env.items():
                    ops.tune_set(kk, vv)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    ops.conv2d(xv, filt, bias, yv, k, s, not args.noact, rv, workspace=wsp)
                e1.record()
                torch.cuda.synchronize()
                ops.tune_reset()
                if rnd:   # round 0 = warm-up
                    times[arm].append(e0.elapsed_time(e1) * 1e3 / args.reps)
                else:
                    outs[arm] = yv.as_nhwc().float().clone()
        for arm, wsp, env in arms:
            for kk, vv in env.items():
                ops.tune_set(kk, vv)
            var = ops.conv_variant(xv, yv, k, s, res, workspace_bytes=wsp.numel() if wsp is not None else 0)
            ops.tune_reset()
            med, mn = statistics.median(times[arm]), min(times[arm])
            diff = (outs[arm] - outs[arms[0][0]]).abs().max().item()
            print(f"{name:28s} {arm:10s} {var:18s} {med:9.1f} {mn:9.1f} {flops / med / 1e6:10.1f} {flops / mn / 1e6:10.1f}   max|d vs arm 0| {diff:.3g}")
        sys.stdout.flush()
    hdr = ws[:64].view(torch.int32).tolist()
    print("workspace ctl words (ticket, finished, error):", hdr[:3])


if __name__ == "__main__":
    main()
