"""Does capturing the eval plan in a HIP graph pay?  (round-3 review item 5.)  Times the yolov3 640x640 batch-32 forward launched call by call
(75 conv launches + 3 decodes through ctypes) against a replay of the same launches captured once with torch.cuda.graph (stream capture: the library
launches on torch's current stream and allocates nothing).  GPU box only."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from yolov3_amd import DetectionModel  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = DetectionModel("yolov3.yaml").to(dev).half().eval()
x = torch.rand(32, 3, 640, 640, device=dev).half()


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


with torch.no_grad():
    eager = timed(lambda: model(x))
    pred0 = model(x)[0].clone()
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                model(x)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):   # (plans are keyed by stream: capture on the stream the warm-up compiled the plan for)
            out = model(x)
        graph = timed(g.replay)
        same = torch.equal(out[0], pred0)
        print(f"forward launched call by call {eager:.3f} ms; HIP graph replay {graph:.3f} ms ({graph - eager:+.3f} ms); outputs identical: {same}")
    except Exception as e:  # noqa: BLE001
        print(f"forward launched call by call {eager:.3f} ms; graph capture failed: {type(e).__name__}: {e}")
