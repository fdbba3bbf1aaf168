"""timing of the val / detect edges around the hot path (lab): AutoShape on raw uint8 images, test-time augmentation, scale_boxes + process_batch of a val-style batch"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from yolov3_amd import AutoShape, DetectionModel, non_max_suppression  # noqa: E402
from yolov3_amd import val as yval  # noqa: E402

dev = torch.device("cuda:0")
m = DetectionModel("yolov3.yaml").to(dev).half().eval()


def timed(fn, n=10, w=3):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


x = torch.rand(32, 3, 640, 640, device=dev).half()
print("forward bs32 640        %.2f ms" % timed(lambda: m(x)))
print("forward bs32 640 TTA    %.2f ms" % timed(lambda: m(x, augment=True)))
imgs_np = [torch.randint(0, 255, (1080, 1920, 3), dtype=torch.uint8).numpy() for _ in range(32)]
a = AutoShape(m)
print("AutoShape 32 x 1080p np %.2f ms" % timed(lambda: a(imgs_np, size=640), n=5, w=2))
print("AutoShape 8 x 1080p np  %.2f ms" % timed(lambda: a(imgs_np[:8], size=640), n=5, w=2))
pred = m(x)[0]
print("NMS bs32 (random head)  %.2f ms" % timed(lambda: non_max_suppression(pred, 0.001, 0.6, multi_label=True, max_det=300)))
