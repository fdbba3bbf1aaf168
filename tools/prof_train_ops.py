"""torch.profiler view of ONE batch-64 train step: call counts and device time per op / kernel (finds host-side glue such as the per-parameter
gradient copies the arena removed).  GPU box only: python tools/prof_train_ops.py"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from yolov3_amd import DetectionModel, ComputeLoss
from yolov3_amd.optim import FusedSGD, ModelEMA, smart_param_groups, GradScaler
from oracle import yolo_oracle as yo
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = DetectionModel("yolov3.yaml", nc=80).to(dev).train()
m.hyp = dict(box=0.05, cls=0.5, cls_pw=1.0, obj=1.0, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0, label_smoothing=0.0)
crit = ComputeLoss(m)
opt = FusedSGD(smart_param_groups(m, 0.01, 5e-4), lr=0.01)
ema = ModelEMA(m)
sc = GradScaler(init_scale=1024.0)
bs = 64
x = torch.rand(bs, 3, 640, 640, device=dev)
tg = yo.synth_targets(bs, 80, seed=1).to(dev)
def step():
    with torch.autocast("cuda", dtype=torch.float16):
        loss, _ = crit(m(x), tg)
    sc.scale(loss).backward()
    sc.step(opt, 10.0, ema); sc.update(); opt.zero_grad()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
rows = [(e.key, e.count, e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total) for e in prof.key_averages()]
rows.sort(key=lambda r: -r[1])
for k, c, t in rows[:40]:
    print(f"{c:6d} {t/1e3:9.3f} ms  {k[:100]}")
