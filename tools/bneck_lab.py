"""Time y3_bneck_pair_fwd against the two launches it replaces (yolov3 layer 2: Bottleneck(64, 64) on the 320x320 map, batch 32)."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolov3_amd import ops  # noqa: E402


def run(c, h, w):
    dev, dt = torch.device("cuda:0"), torch.float16
    n, cm = 32, c // 2
    g = torch.Generator().manual_seed(1)
    xv = ops.View.alloc(n, h, w, c, dt, dev)
    xv.buf.copy_(torch.randn(xv.buf.shape, generator=g).to(dt))
    f1 = ops.pack_filter((torch.randn(cm, c, 1, 1, generator=g) / math.sqrt(c)).to(dev), cm, c, dt)
    f2 = ops.pack_filter((torch.randn(c, cm, 3, 3, generator=g) / math.sqrt(9 * cm)).to(dev), c, cm, dt)
    b1, b2 = torch.zeros(cm, device=dev), torch.zeros(c, device=dev)
    tv, yv, y2 = ops.View.alloc(n, h, w, cm, dt, dev), ops.View.alloc(n, h, w, c, dt, dev), ops.View.alloc(n, h, w, c, dt, dev)

    def fused():
        ops.bneck_pair(xv, f1, b1, True, f2, b2, True, True, yv)

    def two():
        ops.conv2d(xv, f1, b1, tv, 1, 1, True)
        ops.conv2d(tv, f2, b2, y2, 3, 1, True, residual=xv)

    for _ in range(100):
        fused(); two()
    torch.cuda.synchronize()
    for name, fn in (("fused", fused), ("two launches", two), ("fused", fused), ("two launches", two)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"Bottleneck({c}) {h}x{w} batch 32, {name}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us", flush=True)
    print("max |fused - two| =", (yv.as_nhwc().float() - y2.as_nhwc().float()).abs().max().item())


if __name__ == "__main__":
    run(64, 320, 320)
    run(128, 160, 160)
