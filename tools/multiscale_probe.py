"""What a change of input size costs the training step (reference train.py:394-399 --multi-scale: a new size from [0.5, 1.5] x imgsz in steps of 32 for EVERY batch).
yolov3, batch 16, autocast fp16, bench.py's step.  Pass 1 visits the 21 sizes of imgsz 640 in shuffled order (plans are built, the slot's arena grows to the largest);
pass 2 visits them again in another order: per-step wall time, plan builds and arena allocations -- against the same size run twice in a row (no change of shape).
    python tools/multiscale_probe.py [--batch 16]"""
import argparse, json, random, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from oracle import yolo_oracle as yo   # seeded synthetic targets only
from yolov3_amd import ComputeLoss, DetectionModel, train_engine
from yolov3_amd.engine import plan_cache
from yolov3_amd.optim import FusedSGD, GradScaler, ModelEMA, smart_param_groups

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = DetectionModel("yolov3.yaml").to(dev).train()
model.hyp = dict(box=0.05, cls=0.5, cls_pw=1.0, obj=1.0, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0, label_smoothing=0.0)
crit = ComputeLoss(model)
opt = FusedSGD(smart_param_groups(model, 0.01, 5e-4 * args.batch / 64), momentum=0.937, nesterov=True)
ema = ModelEMA(model)
scaler = GradScaler(init_scale=1024.0)
tg = yo.synth_targets(args.batch, 80, seed=1).to(dev)
sizes = list(range(320, 961, 32))
xs = {s: torch.rand(args.batch, 3, s, s, generator=torch.Generator().manual_seed(s)).to(dev) for s in sizes}


def step(s):
    with torch.autocast("cuda", dtype=torch.float16):
        loss, _ = crit(model(xs[s]), tg)
    scaler.scale(loss).backward()
    scaler.unscale_(opt)
    scaler.step(opt, max_norm=10.0, ema=ema)
    scaler.update()
    opt.zero_grad(set_to_none=True)


def timed(s):
    torch.cuda.synchronize()
    t = time.perf_counter()
    step(s)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3


order = sizes[:]
random.Random(1).shuffle(order)
first = {s: timed(s) for s in order}
slot = plan_cache(model).train_slots(torch.float16, dev, train_engine.TrainSlot)[0]
builds, allocs, reserved = train_engine.PLAN_BUILDS, slot.arena_allocations, torch.cuda.memory_reserved()
random.Random(2).shuffle(order)
second = {s: timed(s) for s in order}          # every step changes the shape
same = {s: (timed(s), timed(s))[1] for s in sizes}   # the size of the step before: no change of shape
print(f"# yolov3 train step, batch {args.batch}, autocast fp16; 21 sizes; ms per step (wall, synchronised)")
print(f"# pass 1 built {builds} plans, the arena was allocated {allocs} times ({slot.arena.numel() / 2**30:.2f} GiB = the 960 x 960 plan); pass 2: {train_engine.PLAN_BUILDS - builds} plan builds, "
      f"{slot.arena_allocations - allocs} arena allocations, torch reserved memory {reserved / 2**30:.2f} -> {torch.cuda.memory_reserved() / 2**30:.2f} GiB")
print(f"{'size':>6s} {'pass 1 (build)':>15s} {'pass 2 (shape changed)':>23s} {'same shape again':>17s}")
for s in sizes:
    print(f"{s:6d} {first[s]:15.2f} {second[s]:23.2f} {same[s]:17.2f}")
tot2, tots = sum(second.values()), sum(same.values())
print(json.dumps({"sum_ms_shape_changing": round(tot2, 2), "sum_ms_same_shape": round(tots, 2), "overhead_of_changing_shape_pct": round(100 * (tot2 / tots - 1), 2), "plan_builds_pass2": train_engine.PLAN_BUILDS - builds,
                  "arena_allocations_pass2": slot.arena_allocations - allocs}))
