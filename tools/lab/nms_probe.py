"""The kernels of one y3_nms call on the benchmark's load (bs 32, 25200 x 85 fp16, val settings): run under rocprofv3 --kernel-trace --stats (tools/lab/run_b.sh)."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from oracle import yolo_oracle as yo  # noqa: E402
from yolov3_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
pred = yo.synth_predictions(bs=32, n_rows=25200, nc=80, seed=2).half().to(dev)
for _ in range(3):
    ops.nms_raw(pred, 0.001, 0.6, None, False, True, 300)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    rows, counts = ops.nms_raw(pred, 0.001, 0.6, None, False, True, 300)
torch.cuda.synchronize()
print("ms per call", (time.perf_counter() - t0) / 20 * 1e3, "candidates per image", ops.nms_raw.last_candidates / 32)
