"""Per-layer, per-phase GPU time of one training step (HIP events around every library call of the TrainPlan):

    python tools/train_layers.py [--batch 64] [--top 40]

Phases per conv unit: fwd.conv (conv + statistics), fwd.bn (finalize + BN/act apply), bwd.bn (reduce + apply), bwd.wgrad, bwd.dgrad."""
import argparse
import sys
from collections import defaultdict
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from oracle import yolo_oracle as yo  # noqa: E402
from yolov3_amd import ComputeLoss, DetectionModel, ops, train_engine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--top", type=int, default=45)
args = ap.parse_args()
dev = torch.device("cuda:0")
m = DetectionModel("yolov3.yaml").to(dev).train()
m.hyp = dict(box=0.05, cls=0.5, cls_pw=1.0, obj=1.0, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0, label_smoothing=0.0)
crit = ComputeLoss(m)
x = torch.rand(args.batch, 3, 640, 640, device=dev)
tg = yo.synth_targets(args.batch, 80, seed=1).to(dev)

EV = []          # (label, phase, e0, e1)
CUR = ["?"]


def timed(phase, fn):
    def w(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **k)
        e1.record()
        EV.append((CUR[0], phase, e0, e1))
        return r
    return w


def wrap_unit(cls):
    of, ob = cls.fwd, getattr(cls, "bwd", None)

    def fwd(self):
        CUR[0] = getattr(self, "label", type(self).__name__)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = of(self); e1.record()
        EV.append((CUR[0], "fwd.total", e0, e1))
        return r

    cls.fwd = fwd
    if ob is not None:
        def bwd(self, grads):
            CUR[0] = getattr(self, "label", type(self).__name__)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); r = ob(self, grads); e1.record()
            EV.append((CUR[0], "bwd.total", e0, e1))
            return r
        cls.bwd = bwd


for c in (train_engine.ConvUnit, train_engine.UpsampleUnit, train_engine.MaxPoolUnit, train_engine.SPPPoolUnit):
    wrap_unit(c)
train_engine.TrainPlan.wgrad = timed("bwd.wgrad", train_engine.TrainPlan.wgrad)
ops.conv2d = timed("conv (fwd conv / dgrad)", ops.conv2d)
ops.conv2d_stats = timed("fwd.conv+stats", ops.conv2d_stats)
ops.conv2d_dgrad_s2 = timed("bwd.dgrad_s2", ops.conv2d_dgrad_s2)
ops.stem_conv = timed("fwd.stem", ops.stem_conv)

for it in range(3):
    EV.clear()
    with torch.autocast("cuda", dtype=torch.float16):
        loss, _ = crit(m(x), tg)
    (loss * 1024.0).backward()
    m.zero_grad(set_to_none=True)
torch.cuda.synchronize()
rows = defaultdict(float)
for lab, ph, e0, e1 in EV:
    rows[(lab, ph)] += e0.elapsed_time(e1)
units = defaultdict(dict)
for (lab, ph), t in rows.items():
    units[lab][ph] = t
tot_f = sum(u.get("fwd.total", 0) for u in units.values())
tot_b = sum(u.get("bwd.total", 0) for u in units.values())
print(f"units: fwd {tot_f:.2f} ms, bwd {tot_b:.2f} ms (heads + loss + optimizer not included)")
print(f"{'unit':18s} {'fwd':>7s} {'conv':>7s} {'bn':>7s} | {'bwd':>7s} {'bn':>7s} {'wgrad':>7s} {'dgrad':>7s}")
order = sorted(units.items(), key=lambda kv: -(kv[1].get("fwd.total", 0) + kv[1].get("bwd.total", 0)))
for lab, u in order[: args.top]:
    f, b = u.get("fwd.total", 0), u.get("bwd.total", 0)
    fc = u.get("fwd.conv+stats", 0) + u.get("fwd.stem", 0)
    wg = u.get("bwd.wgrad", 0)
    dg = u.get("bwd.dgrad_s2", 0) + u.get("conv (fwd conv / dgrad)", 0)
    print(f"{lab:18s} {f:7.3f} {fc:7.3f} {f - fc:7.3f} | {b:7.3f} {b - wg - dg:7.3f} {wg:7.3f} {dg:7.3f}")
