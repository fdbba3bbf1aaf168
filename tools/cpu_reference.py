"""Time the UNMODIFIED reference (/root/reference through oracle/ref_shim.py) on this host's CPU cores, on the same bounded sample
bench.py's `cpu_baseline` leg gives the oracle port: SURVEY 8(d) "the oracle path itself (reference Python + stubs)".

    python tools/cpu_reference.py [--threads N] [--out profiles/r03_cpu_reference.json]

Runs in the BUILD container only (the GPU box has no /root/reference); bench.py prints the committed record next to the port's number
(`cpu_baseline.reference_recorded`), with the host it was measured on.  Legs:
  * inference: yolov3 (fused, eval, fp32) forward on 4 seeded 640x640 images + utils.general.non_max_suppression with val.py's
    settings on the seeded synthetic prediction tensor (4 images) -- the workload of BASELINE configs[1] per image;
  * train step: DetectionModel.train() forward + utils.loss.ComputeLoss + backward + SGD(nesterov) step on 4 images (configs[2] per image).
The oracle port is timed in the same process for the port-vs-reference ratio on identical cores."""
from __future__ import annotations

import argparse
import json
import os
import platform
import sys
import time
from pathlib import Path

import torch
import yaml

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from oracle import ref_shim, yolo_oracle as yo  # noqa: E402

HYP = dict(box=0.05, cls=0.5, cls_pw=1.0, obj=1.0, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0, label_smoothing=0.0)


def best_of(fn, n=2):
    best = 1e9
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=len(os.sched_getaffinity(0)))
    ap.add_argument("--bs", type=int, default=4)
    ap.add_argument("--out", default=str(ROOT / "profiles" / "r03_cpu_reference.json"))
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    ns = ref_shim.load()
    cfg = ROOT / "yolov3_amd" / "cfg" / "yolov3.yaml"
    d = yaml.safe_load(open(cfg))
    layers, save, anchors, nc = yo.parse_cfg(d)
    strides = yo.model_strides(layers)
    sd = yo.seeded_state_dict(layers, nc, anchors, strides, seed=0)
    bs = args.bs
    x = torch.rand(bs, 3, 640, 640, generator=torch.Generator().manual_seed(0))
    pred_s = yo.synth_predictions(bs=bs, n_rows=25200, nc=80, seed=2)
    tg = yo.synth_targets(bs, 80, seed=1)

    # ---- inference: reference vs port ----
    m = ns.DetectionModel(str(cfg), ch=3, nc=80)
    m.load_state_dict(sd, strict=True)
    m.eval().fuse()
    with torch.inference_mode():
        m(x[:1])
        t_fwd = best_of(lambda: m(x))
        t_nms = best_of(lambda: ns.non_max_suppression(pred_s, 0.001, 0.6, multi_label=True, max_det=300))
        sdf = yo.fuse_state_dict(sd)
        yo.forward(layers, save, sdf, x[:1], strides)
        p_fwd = best_of(lambda: yo.forward(layers, save, sdf, x, strides))
        p_nms = best_of(lambda: yo.non_max_suppression(pred_s, 0.001, 0.6, multi_label=True, max_det=300))

    # ---- train step: reference ----
    mt = ns.DetectionModel(str(cfg), ch=3, nc=80)
    mt.load_state_dict(sd, strict=True)
    mt.train()
    mt.hyp = dict(HYP)
    crit = ns.ComputeLoss(mt)
    opt = torch.optim.SGD(mt.parameters(), lr=0.01, momentum=0.937, nesterov=True)

    def step():
        loss, _ = crit(mt(x), tg)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(mt.parameters(), max_norm=10.0)
        opt.step()
        opt.zero_grad()

    step()
    t_train = best_of(step)

    rec = {
        "_doc": "unmodified /root/reference (oracle/ref_shim.py stubs for the un-installable third-party imports) timed on the build container's CPU; tools/cpu_reference.py",
        "kind": "reference",
        "host": {"machine": platform.machine(), "cpu": next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?"),
                 "usable_cores": len(os.sched_getaffinity(0)), "threads": torch.get_num_threads(), "torch": torch.__version__},
        "inference": {"value": round(bs / (t_fwd + t_nms), 3), "unit": "images/sec", "cores": torch.get_num_threads(),
                      "sample": f"{bs} images 640x640 fp32 fused-eval DetectionModel.forward ({t_fwd:.2f}s) + utils.general.non_max_suppression val settings ({t_nms:.2f}s), best of 2"},
        "inference_port_same_host": {"value": round(bs / (p_fwd + p_nms), 3), "unit": "images/sec", "cores": torch.get_num_threads(),
                                     "sample": f"oracle/yolo_oracle.py forward ({p_fwd:.2f}s) + NMS ({p_nms:.2f}s), same inputs, same process"},
        "train_step": {"value": round(bs / t_train, 3), "unit": "images/sec", "cores": torch.get_num_threads(),
                       "sample": f"{bs} images 640x640 fp32: train-mode forward + ComputeLoss + backward + clip_grad_norm_(10) + SGD(nesterov) step ({t_train:.2f}s), best of 2"},
    }
    Path(args.out).write_text(json.dumps(rec, indent=1) + "\n")
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
