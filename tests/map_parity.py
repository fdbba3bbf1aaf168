"""mAP parity of the MI355X path against the reference's CPU path on the same weights (BASELINE.json target: "mAP@0.5 within 0.1 of
reference").  coco128 and pretrained weights are not reachable offline, so the experiment is self-contained:

  1. synthetic detection scenes (colour-coded textured rectangles on a noisy background, labels in the reference's normalised
     (image, class, xc, yc, w, h) format) -- `make_scenes`, seeded, CPU generator;
  2. a yolov3-tiny is TRAINED on them with the product's own training path (DetectionModel.train() + ComputeLoss + GradScaler +
     FusedSGD under autocast: the HIP forward / backward / optimizer kernels) until it detects;
  3. the SAME weights are evaluated twice with val.py's procedure (conf 0.001, iou 0.6, multi_label, iouv 0.5:0.95,
     reference val.py:364-421): by the product (HIP forward + decode + non_max_suppression + process_batch, fp32 and fp16 engines) and by
     the CPU oracle (oracle/yolo_oracle.py: the torch-CPU restatement pinned to the unmodified reference by tests/golden);
     P / R / mAP@0.5 / mAP@0.5:0.95 come from yolov3_amd.metrics.ap_per_class, itself pinned to the reference's utils/metrics.py
     (tests/golden/metrics.pt).

Test infrastructure (imports the oracle): used by tests/test_gpu_parity.py::test_map_parity_on_synthetic_scenes and runnable as
`python tests/map_parity.py [--steps N] [--out file.json]` on a GPU box for the numbers in profiles/."""
from __future__ import annotations

import argparse
import json
import sys
import time
from pathlib import Path

import torch
import yaml

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from oracle import yolo_oracle as yo  # noqa: E402

CFG = ROOT / "yolov3_amd" / "cfg"
COLOURS = [[0.9, 0.15, 0.15], [0.15, 0.85, 0.2], [0.2, 0.25, 0.95], [0.9, 0.85, 0.1], [0.8, 0.2, 0.85], [0.1, 0.85, 0.85]]


def make_scenes(n, hw, nc, seed, max_obj=3):
    """images (n, 3, hw, hw) in [0, 1] and labels (m, 6) = (image, class, xc, yc, w, h), normalised (reference label format)"""
    g = torch.Generator().manual_seed(seed)
    imgs = torch.empty(n, 3, hw, hw)
    labels = []
    base = torch.tensor(COLOURS)[:nc]
    yy, xx = torch.meshgrid(torch.arange(hw).float(), torch.arange(hw).float(), indexing="ij")
    for i in range(n):
        a = torch.rand(3, 3, generator=g)
        bg = 0.35 + 0.15 * (a[:, 0, None, None] * torch.sin(xx / hw * 6.28 * (1 + a[0, 1])) + a[:, 1, None, None] * torch.cos(yy / hw * 6.28 * (1 + a[1, 2])))
        bg = bg + 0.05 * torch.randn(3, hw, hw, generator=g)
        for _ in range(int(torch.randint(1, max_obj + 1, (1,), generator=g))):
            c = int(torch.randint(0, nc, (1,), generator=g))
            w, h = float(0.12 + 0.3 * torch.rand(1, generator=g)), float(0.12 + 0.3 * torch.rand(1, generator=g))
            xc, yc = float(w / 2 + (1 - w) * torch.rand(1, generator=g)), float(h / 2 + (1 - h) * torch.rand(1, generator=g))
            x0, x1, y0, y1 = int((xc - w / 2) * hw), int((xc + w / 2) * hw), int((yc - h / 2) * hw), int((yc + h / 2) * hw)
            col = base[c] * (0.8 + 0.2 * torch.rand(1, generator=g))
            tex = 1.0 + 0.08 * torch.sin(xx[y0:y1, x0:x1] * 0.7) * torch.cos(yy[y0:y1, x0:x1] * 0.5)
            bg[:, y0:y1, x0:x1] = col[:, None, None] * tex
            labels.append([i, c, (x0 + x1) / 2 / hw, (y0 + y1) / 2 / hw, (x1 - x0) / hw, (y1 - y0) / hw])
        imgs[i] = bg.clamp(0, 1)
    return imgs, torch.tensor(labels, dtype=torch.float32)


def hyp_for(nc, hw, nl):
    """hyp.scratch-low with the scalings of reference train.py:236-241 (box / cls / obj by layers, classes and image size)"""
    return dict(box=0.05 * 3 / nl, cls=0.5 * nc / 80 * 3 / nl, cls_pw=1.0, obj=1.0 * (hw / 640) ** 2 * 3 / nl, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0,
                label_smoothing=0.0)


def batch_targets(labels, idx):
    sel = []
    for j, i in enumerate(idx.tolist()):
        lab = labels[labels[:, 0] == i].clone()
        lab[:, 0] = j
        sel.append(lab)
    return torch.cat(sel) if sel else torch.zeros(0, 6)


def train_on_gpu(name, nc, hw, steps, dev, train_x, train_l, bs=16, lr0=0.02, seed=1):
    """the product's training path (reference train.py:402-422 loop shape); returns the trained model (fp32 master weights)"""
    from yolov3_amd import ComputeLoss, DetectionModel, FusedSGD, GradScaler, smart_param_groups

    torch.manual_seed(seed)
    d = yaml.safe_load(open(CFG / f"{name}.yaml"))
    layers, _, anchors, nc_v = yo.parse_cfg(d, 3, nc)
    sd = yo.seeded_state_dict(layers, nc_v, anchors, yo.model_strides(layers), seed=seed)
    model = DetectionModel(f"{name}.yaml", nc=nc)
    model.load_state_dict(sd)
    model = model.to(dev).train()
    nl = model.model[-1].nl
    model.hyp = hyp_for(nc, hw, nl)
    crit = ComputeLoss(model)
    opt = FusedSGD(smart_param_groups(model, lr0, 5e-4), momentum=0.937, nesterov=True)
    scaler = GradScaler(init_scale=1024.0)
    xd = train_x.to(dev)
    losses = []
    for it in range(steps):
        idx = torch.randint(0, train_x.shape[0], (bs,), generator=torch.Generator().manual_seed(100 + it))
        x = xd[idx.to(dev)]
        tg = batch_targets(train_l, idx).to(dev)
        lr = lr0 * min(1.0, (it + 1) / 30)   # linear warm-up (train.py:383-392), then constant
        for g in opt.param_groups:
            g["lr"] = lr
        with torch.autocast("cuda", dtype=torch.float16):
            loss, _ = crit(model(x), tg)
        scaler.scale(loss).backward()
        scaler.unscale_(opt)
        scaler.step(opt, max_norm=10.0)
        scaler.update()
        opt.zero_grad(set_to_none=True)
        if it % 50 == 0 or it == steps - 1:
            losses.append((it, float(loss)))
    return model, losses


def labels_native(val_l, i, hw):
    lab = val_l[val_l[:, 0] == i][:, 1:].clone()
    b = lab[:, 1:5] * hw
    return torch.cat([lab[:, :1], torch.stack([b[:, 0] - b[:, 2] / 2, b[:, 1] - b[:, 3] / 2, b[:, 0] + b[:, 2] / 2, b[:, 1] + b[:, 3] / 2], 1)], 1)


def evaluate_product(model, val_x, val_l, hw, dev, dtype, bs=16):
    """val.py:364-421 with the product: forward + decode + NMS + process_batch on the device, ap_per_class on the host"""
    import copy

    from yolov3_amd import metrics, non_max_suppression, process_batch

    m = copy.deepcopy(model).to(dtype).eval()
    iouv = torch.linspace(0.5, 0.95, 10, device=dev)
    stats, n_det = [], 0
    with torch.no_grad():
        for b in range(0, val_x.shape[0], bs):
            pred = m(val_x[b:b + bs].to(dev).to(dtype))[0]
            dets = non_max_suppression(pred, 0.001, 0.6, multi_label=True, max_det=300)
            for si, det in enumerate(dets):
                labn = labels_native(val_l, b + si, hw).to(dev)
                correct = process_batch(det, labn, iouv) if det.shape[0] and labn.shape[0] else torch.zeros(det.shape[0], 10, dtype=torch.bool, device=dev)
                stats.append((correct.cpu().numpy(), det[:, 4].cpu().numpy(), det[:, 5].cpu().numpy(), labn[:, 0].cpu().numpy()))
                n_det += det.shape[0]
    return metrics.mean_results(stats), n_det


def evaluate_oracle(name, nc, state_dict, val_x, val_l, hw, bs=16):
    """the same procedure through the CPU oracle (the pinned restatement of the reference's fp32 CPU path)"""
    from yolov3_amd import metrics

    d = yaml.safe_load(open(CFG / f"{name}.yaml"))
    layers, save, _, _ = yo.parse_cfg(d, 3, nc)
    strides = yo.model_strides(layers)
    sd = {k: v.detach().float().cpu() for k, v in state_dict.items()}
    iouv = torch.linspace(0.5, 0.95, 10)
    stats, n_det = [], 0
    with torch.no_grad():
        for b in range(0, val_x.shape[0], bs):
            pred = yo.forward(layers, save, sd, val_x[b:b + bs], strides)[0]
            dets = yo.non_max_suppression(pred, 0.001, 0.6, multi_label=True, max_det=300)
            for si, det in enumerate(dets):
                labn = labels_native(val_l, b + si, hw)
                correct = yo.process_batch(det, labn, iouv) if det.shape[0] and labn.shape[0] else torch.zeros(det.shape[0], 10, dtype=torch.bool)
                stats.append((correct.numpy(), det[:, 4].numpy(), det[:, 5].numpy(), labn[:, 0].numpy()))
                n_det += det.shape[0]
    return metrics.mean_results(stats), n_det


def run(steps=1500, hw=256, nc=3, n_train=256, n_val=64, name="yolov3-tiny", dev=None):
    dev = dev or torch.device("cuda:0")
    train_x, train_l = make_scenes(n_train, hw, nc, seed=1)
    val_x, val_l = make_scenes(n_val, hw, nc, seed=2)
    t0 = time.perf_counter()
    model, losses = train_on_gpu(name, nc, hw, steps, dev, train_x, train_l)
    torch.cuda.synchronize()
    t_train = time.perf_counter() - t0
    res = {"model": name, "nc": nc, "imgsz": hw, "train_images": n_train, "val_images": n_val, "val_labels": int(val_l.shape[0]), "train_steps": steps,
           "train_seconds": round(t_train, 1), "loss_curve": losses}
    (p32, n32), (p16, n16) = evaluate_product(model, val_x, val_l, hw, dev, torch.float32), evaluate_product(model, val_x, val_l, hw, dev, torch.float16)
    ref, nref = evaluate_oracle(name, nc, model.state_dict(), val_x, val_l, hw)
    keys = ("P", "R", "mAP50", "mAP50-95")
    res["reference_cpu_fp32"] = dict(zip(keys, ref), detections=nref)
    res["hip_fp32"] = dict(zip(keys, p32), detections=n32)
    res["hip_fp16"] = dict(zip(keys, p16), detections=n16)
    res["abs_diff_fp32"] = {k: abs(a - b) for k, a, b in zip(keys, p32, ref)}
    res["abs_diff_fp16"] = {k: abs(a - b) for k, a, b in zip(keys, p16, ref)}
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--imgsz", type=int, default=256)
    ap.add_argument("--model", default="yolov3-tiny")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    out = run(steps=a.steps, hw=a.imgsz, name=a.model)
    s = json.dumps(out, indent=1)
    print(s)
    if a.out:
        Path(a.out).write_text(s + "\n")
