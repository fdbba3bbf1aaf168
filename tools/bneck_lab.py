"""Time y3_bneck_pair_fwd against the two launches it replaces (yolov3 layer 2: Bottleneck(64, 64) on the 320x320 map, batch 32)."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolov3_amd import ops  # noqa: E402


def main():
    dev, dt = torch.device("cuda:0"), torch.float16
    n, h, w = 32, 320, 320
    g = torch.Generator().manual_seed(1)
    xv = ops.View.alloc(n, h, w, 64, dt, dev)
    xv.buf.copy_(torch.randn(xv.buf.shape, generator=g).to(dt))
    f1 = ops.pack_filter((torch.randn(32, 64, 1, 1, generator=g) / 8).to(dev), 32, 64, dt)
    f2 = ops.pack_filter((torch.randn(64, 32, 3, 3, generator=g) / math.sqrt(288)).to(dev), 64, 32, dt)
    b1, b2 = torch.zeros(32, device=dev), torch.zeros(64, device=dev)
    tv, yv, y2 = ops.View.alloc(n, h, w, 32, dt, dev), ops.View.alloc(n, h, w, 64, dt, dev), ops.View.alloc(n, h, w, 64, dt, dev)

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
        print(f"{name}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us", flush=True)
    print("max |fused - two| =", (yv.as_nhwc().float() - y2.as_nhwc().float()).abs().max().item())


if __name__ == "__main__":
    main()
