"""A/B of the batch-64 train step (bench.py's step) with the filter gradients on a second HIP stream -- unconfined, or confined to N CUs by a CU mask
(masked_stream) so that the CUs outside the mask stay free for the HBM-bound BatchNorm passes of the backward's critical chain.  One process, interleaved
rounds, the plan rebuilt per arm.  Usage: python tools/wgrad_overlap_ab.py [--batch 64] [--rounds 2] [--steps 6] [--arms base,s,s+m192,s+hp]"""
import argparse, json, os, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from oracle import yolo_oracle as yo   # seeded synthetic targets only
from yolov3_amd import ComputeLoss, DetectionModel
from yolov3_amd.engine import plan_cache
from yolov3_amd.optim import FusedSGD, GradScaler, ModelEMA, smart_param_groups
sys.path.insert(0, str(Path(__file__).resolve().parent / "lab"))
from cu_mask import masked_stream

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--imgsz", type=int, default=640)
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--arms", default="base,s,s+m192,s+m128,s+hp")
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = DetectionModel("yolov3.yaml").to(dev).train()
model.hyp = dict(box=0.05, cls=0.5, cls_pw=1.0, obj=1.0, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0, label_smoothing=0.0)
crit = ComputeLoss(model)
opt = FusedSGD(smart_param_groups(model, 0.01, 5e-4 * args.batch / 64), momentum=0.937, nesterov=True)
ema = ModelEMA(model)
scaler = GradScaler(init_scale=1024.0)
x = torch.rand(args.batch, 3, args.imgsz, args.imgsz, generator=torch.Generator().manual_seed(0)).to(dev)
tg = yo.synth_targets(args.batch, 80, seed=1).to(dev)
hp_stream = torch.cuda.Stream(device=dev, priority=-1)
MASK_CUS = 0


def step():
    with torch.autocast("cuda", dtype=torch.float16):
        loss, _ = crit(model(x), tg)
    scaler.scale(loss).backward()
    scaler.unscale_(opt)
    scaler.step(opt, max_norm=10.0, ema=ema)
    scaler.update()
    opt.zero_grad(set_to_none=True)
    return loss


def set_arm(arm):
    """arm = tokens joined by '+': base (one stream) | s (filter gradients on a side stream) | mN (that stream confined to N CUs) | hp (compute work on a high-priority stream)"""
    global MASK_CUS
    os.environ.pop("Y3_WGRAD_STREAM", None)
    hp, MASK_CUS = False, 0
    for tok in arm.split("+"):
        if tok == "base":
            continue
        os.environ["Y3_WGRAD_STREAM"] = "1"
        if tok == "s":
            pass
        elif tok == "hp":
            hp = True
        elif tok[0] == "m":
            MASK_CUS = int(tok[1:])
        else:
            raise SystemExit(f"unknown arm token {tok}")
    pc = plan_cache(model)
    for k in [k for k in pc.plans if k[0] == "train"]:
        del pc.plans[k]
    return hp


def run(arm):
    hp = set_arm(arm)
    ctx = torch.cuda.stream(hp_stream) if hp else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        step()   # builds the plan
        if MASK_CUS:
            for k, pl in plan_cache(model).plans.items():
                if k[0] == "train":
                    pl.wgrad_stream = masked_stream(dev, MASK_CUS)
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
    return dt * 1e3, float(loss.detach())


arms = args.arms.split(",")
res = {a: [] for a in arms}
for r in range(args.rounds):
    for a in arms:
        try:
            ms, loss = run(a)
            res[a].append(round(ms, 3))
            print(f"round {r} {a:16s} {ms:8.3f} ms/step  {args.batch / ms * 1e3:8.1f} img/s  loss {loss:.5f}", flush=True)
        except Exception as e:  # noqa: BLE001
            print(f"round {r} {a:16s} FAILED {type(e).__name__}: {e}"[:300], flush=True)
print(json.dumps({"workload": f"yolov3 train step {args.imgsz}x{args.imgsz} batch={args.batch} autocast fp16 (bench.py's step)", "ms_per_step": res,
                  "best": {a: min(v) for a, v in res.items() if v}}))
