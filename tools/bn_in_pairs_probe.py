"""Per pair: forward time of (producer unit + its 1x1 consumer unit) with the producer's BatchNorm applied inside the consumer's launch (Y3_BN_IN_CONSUMER=1, default) and as two
launches (=0), same process, interleaved: HIP events around the two units' fwd().  yolov3 640x640 batch 64 autocast fp16.   python tools/bn_in_pairs_probe.py"""
import os, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from yolov3_amd import DetectionModel, train_engine
from yolov3_amd.engine import plan_cache

dev = torch.device("cuda:0")
m = DetectionModel("yolov3.yaml").to(dev).train()
x = torch.rand(64, 3, 640, 640, device=dev)


def run(mode, reps=4):
    os.environ["Y3_BN_IN_CONSUMER"] = mode
    pc = plan_cache(m)
    pc.clear()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        m(x)
    plan = next(p for k, p in pc.plans.items() if k[0] == "train")
    units = plan.units
    pairs = [(a, b) for a, b in zip(units, units[1:]) if isinstance(a, train_engine.ConvUnit) and isinstance(b, train_engine.ConvUnit) and b.k == 1 and b.x is a.y]
    acc = {}
    orig = train_engine.ConvUnit.fwd
    evs = []

    def fwd(self):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = orig(self); e1.record()
        evs.append((self.label, e0, e1))
        return r

    train_engine.ConvUnit.fwd = fwd
    try:
        for _ in range(reps):
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
                m(x)
    finally:
        train_engine.ConvUnit.fwd = orig
    torch.cuda.synchronize()
    t = {}
    for lab, e0, e1 in evs:
        t[lab] = t.get(lab, 0.0) + e0.elapsed_time(e1) / reps
    return {(a.label, b.label): (t[a.label] + t[b.label], b.bn_in is not None) for a, b in pairs}


res = {"0": run("0"), "1": run("1")}
res2 = {"0": run("0"), "1": run("1")}
print(f"{'producer':12s} {'consumer':12s} {'two launches':>13s} {'one launch':>11s}   (ms, forward of both units; second round in brackets)")
tot0 = tot1 = 0.0
for key in res["0"]:
    a0, a1 = res["0"][key][0], res["1"][key][0]
    b0, b1 = res2["0"][key][0], res2["1"][key][0]
    fused = res["1"][key][1]
    print(f"{key[0]:12s} {key[1]:12s} {a0:13.3f} {a1:11.3f}   [{b0:.3f} {b1:.3f}] {'fused' if fused else 'not covered'}")
    if fused:
        tot0 += min(a0, b0); tot1 += min(a1, b1)
print(f"covered pairs: {tot0:.3f} -> {tot1:.3f} ms")
