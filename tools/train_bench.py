"""Phase timing of one training step (forward / loss / backward) on MI355X.  Not the driver's bench (bench.py keeps
BASELINE configs[1]); this is the measurement for the training half of the path (configs[2] per-GPU shape)."""
import argparse, json, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from oracle import yolo_oracle as yo
from yolov3_amd import ComputeLoss, DetectionModel

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--imgsz", type=int, default=640)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--model", default="yolov3")
ap.add_argument("--fused", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda:0")
m = DetectionModel(f"{args.model}.yaml").to(dev).train()
m.hyp = dict(box=0.05, cls=0.5, cls_pw=1.0, obj=1.0, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0, label_smoothing=0.0)
crit = ComputeLoss(m)
from yolov3_amd.optim import FusedSGD, ModelEMA, smart_param_groups
if args.fused:
    opt = FusedSGD(smart_param_groups(m, 0.01, 5e-4), momentum=0.937, nesterov=True)
    ema = ModelEMA(m)
else:
    opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.937, nesterov=True)
x = torch.rand(args.batch, 3, args.imgsz, args.imgsz, device=dev)
tg = yo.synth_targets(args.batch, 80, seed=1).to(dev)
def sync(): torch.cuda.synchronize(); return time.perf_counter()
res = []
for it in range(args.steps + 1):
    t0 = sync()
    with torch.autocast("cuda", dtype=torch.float16):
        raws = m(x)
        t1 = sync()
        loss, items = crit(raws, tg)
    t2 = sync()
    (loss * 1024.0).backward()
    t3 = sync()
    if args.fused:
        opt.step(grad_scale=1024.0, max_norm=10.0, ema=ema); opt.zero_grad()
    else:
        for p_ in m.parameters():
            p_.grad.div_(1024.0)  # GradScaler.unscale_
        opt.step(); opt.zero_grad(set_to_none=True)
    t4 = sync()
    if it:
        res.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
f, l, b, o = (sum(r[i] for r in res) / len(res) * 1e3 for i in range(4))
tot = f + l + b + o
print(json.dumps({"workload": f"{args.model} train step {args.imgsz}x{args.imgsz} batch={args.batch} autocast fp16 (fwd BN batch stats + ComputeLoss + bwd + torch SGD)",
                  "ms": {"forward": round(f, 2), "loss": round(l, 2), "backward": round(b, 2), "optimizer(fused sgd+clip+ema)" if args.fused else "optimizer(torch sgd)": round(o, 2), "total": round(tot, 2)},
                  "images_per_sec": round(args.batch / tot * 1e3, 1), "loss": float(loss)}))
