"""Shader clock and socket power while one conv kernel runs back to back (rocm-smi sampled from a thread).  GPU box only:
python tools/clk_probe.py [--seconds 6]"""
import argparse
import math
import os
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolov3_amd import ops  # noqa: E402


def sample(stop, out):
    while not stop.is_set():
        try:
            txt = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
        except Exception as e:  # noqa: BLE001
            out.append(("err", str(e)))
            return
        sclk = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", txt)
        pw = re.search(r"Power \(W\): ([\d.]+)", txt)
        out.append((time.time(), int(sclk.group(1)) if sclk else None, float(pw.group(1)) if pw else None))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=6.0)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    ws = ops.conv_workspace(dev)
    cases = [("3x3 512->1024 @20x20 bs32 (v10)", 32, 20, 20, 512, 1024, 3), ("3x3 128->256 @80x80 bs32 (v10h)", 32, 80, 80, 128, 256, 3), ("1x1 256->128 @80x80 bs32", 32, 80, 80, 256, 128, 1)]
    for name, n, h, w, cin, cout, k in cases:
        xv = ops.View.alloc(n, h, w, cin, torch.float16, dev)
        ops.nchw_to_nhwc(torch.randn(n, cin, h, w, generator=g).to(dev), xv)
        filt = ops.pack_filter((torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)).to(dev), cout, cin, torch.float16)
        bias = torch.randn(cout, generator=g).to(dev)
        yv = ops.View.alloc(n, h, w, cout, torch.float16, dev)
        for _ in range(20):
            ops.conv2d(xv, filt, bias, yv, k, 1, True, None, workspace=ws)
        torch.cuda.synchronize()
        stop, out = threading.Event(), []
        th = threading.Thread(target=sample, args=(stop, out))
        th.start()
        t0 = time.time()
        launches = 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        while time.time() - t0 < a.seconds:
            for _ in range(200):
                ops.conv2d(xv, filt, bias, yv, k, 1, True, None, workspace=ws)
            launches += 200
            torch.cuda.synchronize()
        e1.record()
        torch.cuda.synchronize()
        stop.set()
        th.join()
        us = e0.elapsed_time(e1) * 1e3 / launches
        flops = 2.0 * n * h * w * cin * cout * k * k
        rows = [r for r in out if r[0] != "err" and r[1]]
        clk = sorted(r[1] for r in rows)
        pw = sorted(r[2] for r in rows if r[2])
        med = lambda v: v[len(v) // 2] if v else None  # noqa: E731
        print(f"{name}: {us:.1f} us per launch, {flops / us / 1e6:.0f} TFLOP/s; rocm-smi over {len(rows)} samples: sclk median {med(clk)} MHz (min {clk[0] if clk else None}, max {clk[-1] if clk else None}), "
              f"socket power median {med(pw)} W (max {pw[-1] if pw else None})", flush=True)


if __name__ == "__main__":
    main()
