"""Generate golden vectors by running the UNMODIFIED reference (/root/reference, through
oracle/ref_shim.py) on seeded synthetic inputs.  Run in the build container only:

    python tests/golden/make_golden.py

The reference holds no tests / golden vectors of its own, so these files ARE
the pin: tests/test_oracle_golden.py checks oracle/yolo_oracle.py against them on CPU, and the
``-m gpu`` tests check the HIP path against them on the MI355X.  Inputs are regenerated from seeds
(torch CPU generator) by oracle.yolo_oracle helpers, so the fixtures only store reference OUTPUTS
(plus input checksums to detect RNG drift).
"""
from __future__ import annotations

import sys
from pathlib import Path

import torch
import yaml

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from oracle import ref_shim, upstream, yolo_oracle as yo  # noqa: E402

OUT = Path(__file__).resolve().parent
CFG = ROOT / "yolov3_amd" / "cfg"

HYP = dict(box=0.05, cls=0.5, cls_pw=1.0, obj=1.0, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0, label_smoothing=0.0)


def checksum(t: torch.Tensor) -> float:
    return float(t.double().abs().sum())


def build_ref_model(ns, name, nc, seed):
    d = yaml.safe_load(open(CFG / f"{name}.yaml"))
    layers, save, anchors, nc_v = yo.parse_cfg(d, 3, nc)
    strides = yo.model_strides(layers)
    sd = yo.seeded_state_dict(layers, nc_v, anchors, strides, seed=seed)
    m = ns.DetectionModel(str(CFG / f"{name}.yaml"), ch=3, nc=nc)
    missing = m.load_state_dict(sd, strict=True)
    assert [float(s) for s in m.stride] == [float(s) for s in strides], (m.stride, strides)
    return m, sd, layers, save, strides


def gen_model_goldens(ns):
    out = {}
    for name, nc, hw, bs in [("yolov3-tiny", 80, 64, 2), ("yolov3", 80, 64, 2), ("yolov3-spp", 80, 64, 1), ("yolov3", 7, 96, 1)]:
        m, sd, layers, save, strides = build_ref_model(ns, name, nc, seed=11)
        g = torch.Generator().manual_seed(5)
        x = torch.rand(bs, 3, hw, hw, generator=g)
        rec = {"x_sum": checksum(x), "w_sum": sum(checksum(v) for v in sd.values() if v.is_floating_point())}
        m.train()
        with torch.no_grad():
            tr = m(x)
        rec["train_raw"] = [t.clone() for t in tr]
        # train-mode forward updates running stats; reload to keep eval independent
        m.load_state_dict(sd)
        m.eval()
        with torch.no_grad():
            pred, raw = m(x)
        rec["eval_pred"], rec["eval_raw"] = pred.clone(), [t.clone() for t in raw]
        m.fuse()
        with torch.no_grad():
            predf, _ = m(x)
        rec["fused_pred"] = predf.clone()
        rec["fused_sd_sum"] = sum(checksum(v) for v in m.state_dict().values() if v.is_floating_point())
        out[f"{name}-nc{nc}-{hw}-bs{bs}"] = rec
        print(name, nc, hw, "pred", tuple(pred.shape), "max|fused-eval|", float((predf - pred).abs().max()))
    torch.save(out, OUT / "model_fwd.pt")


def gen_decode_goldens(ns):
    """Detect eval branch alone: identity 1x1 convs so the module's own decode code runs on seeded raw
    maps, fp32 and fp16."""
    out = {}
    anchors = [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]
    for nc, dtype, sizes in [(80, torch.float32, (12, 6, 3)), (80, torch.float16, (40, 20, 10)), (3, torch.float16, (9, 5, 2))]:
        no = nc + 5
        det = ns.Detect(nc, anchors, ch=(3 * no,) * 3)
        det.stride = torch.tensor([8.0, 16.0, 32.0])
        det.anchors /= det.stride.view(-1, 1, 1)
        for conv in det.m:
            conv.weight.data = torch.eye(3 * no).view(3 * no, 3 * no, 1, 1)
            conv.bias.data.zero_()
        det.eval()
        g = torch.Generator().manual_seed(21)
        xs = [torch.randn(2, 3 * no, s, s + 1, generator=g) * 2.0 for s in sizes]
        if dtype == torch.float16:
            # elementwise decode in half, as DetectMultiBackend(fp16=True) does (models/common.py:475);
            # keep the identity conv out of the half path by feeding pre-rounded maps through fp32 conv
            xs = [x.half().float() for x in xs]
            det_h = det
            with torch.no_grad():
                raws = [det_h.m[i](x).view(2, 3, no, x.shape[2], x.shape[3]).permute(0, 1, 3, 4, 2).contiguous().half() for i, x in enumerate(xs)]
            # run the module's decode lines on half tensors: swap convs for Identity, feed NCHW half
            det.m = torch.nn.ModuleList(torch.nn.Identity() for _ in range(3))
            det.half()
            det.stride = det.stride.half()
            with torch.no_grad():
                z, _ = det([x.half() for x in xs])
        else:
            with torch.no_grad():
                z, raws = det([x.clone() for x in xs])
        out[f"nc{nc}-{str(dtype).split('.')[-1]}"] = {
            "sizes": sizes,
            "in_sum": sum(checksum(x) for x in xs),
            "z": z.clone(),
            "anchors_grid": det.anchors.clone().float(),
        }
        print("decode", nc, dtype, tuple(z.shape))
    torch.save(out, OUT / "decode.pt")


def gen_nms_goldens(ns):
    out = {}
    # known-answer example (SURVEY 8a' item 8)
    p = torch.tensor(
        [[[50, 50, 20, 20, 0.9, 0.9, 0.5, 0.0], [200, 200, 30, 30, 0.8, 0.1, 0.2, 0.95], [52, 51, 20, 20, 0.7, 0.8, 0.6, 0.0], [400, 400, 10, 10, 0.0005, 0.9, 0.9, 0.9]]]
    )
    out["kat_val"] = ns.non_max_suppression(p, 0.001, 0.6, multi_label=True)
    out["kat_det"] = ns.non_max_suppression(p, 0.25, 0.45)
    out["kat_cls2"] = ns.non_max_suppression(p, 0.25, 0.45, classes=[2])
    out["kat_agn"] = ns.non_max_suppression(p, 0.25, 0.45, agnostic=True)

    cases = {
        # name: (gen kwargs, nms kwargs)
        "val_fp32": (dict(bs=3, n_rows=2400, nc=80, seed=2), dict(conf_thres=0.001, iou_thres=0.6, multi_label=True, max_det=300)),
        "det_fp32": (dict(bs=3, n_rows=2400, nc=80, seed=3), dict(conf_thres=0.25, iou_thres=0.45, max_det=1000)),
        "det_agnostic": (dict(bs=2, n_rows=2400, nc=80, seed=4), dict(conf_thres=0.25, iou_thres=0.45, agnostic=True)),
        "det_classes": (dict(bs=2, n_rows=2400, nc=8, seed=5, n_gt=40), dict(conf_thres=0.1, iou_thres=0.45, classes=[0, 3, 5])),
        "val_nc3_maxdet": (dict(bs=2, n_rows=3000, nc=3, seed=6, hits=0.3), dict(conf_thres=0.001, iou_thres=0.6, multi_label=True, max_det=50)),
        "single_class": (dict(bs=2, n_rows=1500, nc=1, seed=7, hits=0.2), dict(conf_thres=0.01, iou_thres=0.5, multi_label=True)),
        "det_fp16": (dict(bs=2, n_rows=1200, nc=80, seed=8, dtype=torch.float16), dict(conf_thres=0.25, iou_thres=0.45)),
        "all_filtered": (dict(bs=2, n_rows=500, nc=80, seed=9, hits=0.0), dict(conf_thres=0.9, iou_thres=0.45)),
    }
    def tie_free(pred, thr):
        sc = (pred[..., 5:].float() * pred[..., 4:5].float()).half()
        for b in range(pred.shape[0]):
            s = sc[b][pred[b, :, 4] > thr].max(1)[0]
            s = s[s > thr]
            if s.unique().numel() != s.numel():
                return False
        return True

    for name, (gk, nk) in cases.items():
        pred = yo.synth_predictions(**gk)
        if pred.dtype == torch.float16:
            # the reference's argsort is unstable -> golden only well-defined without exact score ties
            while not tie_free(pred, nk["conf_thres"]):
                gk["seed"] += 100
                pred = yo.synth_predictions(**gk)
        res = ns.non_max_suppression(pred.clone(), **nk)
        out[name] = {"gen": {k: (str(v) if isinstance(v, torch.dtype) else v) for k, v in gk.items()}, "nms": nk, "in_sum": checksum(pred), "out": [r.clone() for r in res]}
        print("nms", name, [tuple(r.shape) for r in res])
    # autolabel rows (labels=...) case
    pred = yo.synth_predictions(bs=2, n_rows=800, nc=80, seed=10)
    lb = [torch.tensor([[3.0, 100, 120, 40, 50], [7.0, 300, 310, 80, 60]]), torch.zeros(0, 5)]
    out["labels"] = {"in_sum": checksum(pred), "lb": lb, "out": ns.non_max_suppression(pred.clone(), 0.25, 0.45, labels=lb)}
    torch.save(out, OUT / "nms.pt")


def gen_loss_goldens(ns):
    out = {}
    # ("...-sorted": ComputeLoss.sort_obj_iou = True, utils/loss.py:101,156-158 -- the duplicate-cell case keeps another iou than the default order does)
    for name, nc, hw, bs, nt_mode in [("yolov3", 80, 128, 3, "synth"), ("yolov3-tiny", 80, 96, 2, "synth"), ("yolov3", 80, 64, 2, "empty"), ("yolov3", 5, 64, 2, "dups"), ("yolov3", 5, 64, 2, "edges"),
                                      ("yolov3", 5, 64, 2, "dups_sorted")]:
        sort_iou = nt_mode.endswith("_sorted")
        key_mode, nt_mode = nt_mode, nt_mode.replace("_sorted", "")
        m, sd, layers, save, strides = build_ref_model(ns, name, nc, seed=13)
        hyp = dict(HYP)
        nl = len(strides)
        hyp["box"] *= 3 / nl  # reference train.py:327-329
        hyp["cls"] *= nc / 80 * 3 / nl
        hyp["obj"] *= (hw / 640) ** 2 * 3 / nl
        m.hyp = hyp
        crit = ns.ComputeLoss(m)
        crit.sort_obj_iou = sort_iou
        p = [t.requires_grad_(True) for t in yo.synth_raw_predictions([(bs, 3, hw // s, hw // s, nc + 5) for s in strides], seed=31)]
        if nt_mode == "synth":
            tg = yo.synth_targets(bs, nc, seed=1)
        elif nt_mode == "empty":
            tg = torch.zeros(0, 6)
        elif nt_mode == "edges":  # centres on the image border / exact cell edges: clamp + offset corner cases
            tg = torch.tensor(
                [[0, 1, 1.0, 1.0, 0.3, 0.3], [0, 2, 0.0, 0.0, 0.2, 0.25], [1, 0, 0.5, 0.5, 0.4, 0.4], [1, 3, 1.0, 0.25, 0.1, 0.12], [0, 4, 0.125, 0.875, 0.05, 0.9]],
                dtype=torch.float32,
            )
        else:  # forced duplicate cells: identical centres, different classes / sizes
            tg = torch.tensor(
                [[0, 1, 0.51, 0.52, 0.2, 0.3], [0, 2, 0.51, 0.52, 0.21, 0.29], [0, 1, 0.515, 0.525, 0.2, 0.3], [1, 4, 0.26, 0.74, 0.5, 0.45], [1, 0, 0.26, 0.74, 0.5, 0.45]],
                dtype=torch.float32,
            )
        loss, items = crit(p, tg)
        loss.backward()
        out[f"{name}-nc{nc}-{hw}-{key_mode}"] = {
            "hyp": hyp,
            "p_sum": sum(checksum(t.detach()) for t in p),
            "bs": bs,
            "targets": tg,
            "loss": loss.detach().clone(),
            "items": items.clone(),
            "grads": [t.grad.clone() for t in p],
            "anchors_grid": m.model[-1].anchors.clone(),
        }
        print("loss", name, nc, hw, key_mode, float(loss), items.tolist())
    torch.save(out, OUT / "loss.pt")


VAL_MATCH_CASES = {
    # name: synth_val_case kwargs
    "coco_like": dict(seed=0, n_lab=12, n_det=60, nc=5),
    "crowded_dups": dict(seed=1, n_lab=30, n_det=300, nc=3, dup=0.3),
    "single_class": dict(seed=2, n_lab=8, n_det=40, nc=1, dup=1.0),
    "few_labels": dict(seed=3, n_lab=1, n_det=25, nc=4),
    "no_labels": dict(seed=4, n_lab=0, n_det=10, nc=4),
    "one_det": dict(seed=5, n_lab=6, n_det=1, nc=2),
    "many_classes": dict(seed=6, n_lab=40, n_det=200, nc=80),
}
VAL_SCALE_CASES = {
    # name: (img1_shape, img0_shape, ratio_pad)
    "landscape": ((640, 640), (480, 640), None),
    "portrait_hd": ((640, 640), (1280, 720), None),
    "rect_val": ((384, 640), (427, 640), ((0.9, 0.9), (0.0, 0.15))),       # val.py:397 passes shapes[si][1] = (ratio, pad)
    "upscaled": ((640, 640), (200, 333), ((1.92, 1.92), (0.3199999, 128.0))),
}


def gen_val_edge_goldens(ns):
    """Output edge after NMS: the unmodified reference's val.process_batch (val.py:147-188) and utils.general.scale_boxes
    (:613-626) on the seeded cases of oracle.yolo_oracle.synth_val_case / synth_predictions."""
    import importlib

    val = importlib.import_module("val")  # reference val.py (imports resolve through the shim)
    from utils.general import scale_boxes  # type: ignore

    out = {"match": {}, "scale": {}}
    iouv = torch.linspace(0.5, 0.95, 10)  # val.py:262
    for name, kw in VAL_MATCH_CASES.items():
        det, lab = yo.synth_val_case(**kw)
        if det.shape[0] == 0 or lab.shape[0] == 0:
            # the reference never calls process_batch without labels (val.py:386-395 short-cuts) -- zeros by its own convention
            correct = torch.zeros(det.shape[0], iouv.numel(), dtype=torch.bool)
        else:
            correct = val.process_batch(det.clone(), lab.clone(), iouv)
        out["match"][name] = {"gen": kw, "in_sum": checksum(det) + checksum(lab), "correct": correct.clone()}
        print("match", name, tuple(det.shape), tuple(lab.shape), correct.sum(0).tolist())
    for name, (s1, s0, rp) in VAL_SCALE_CASES.items():
        boxes = yo.synth_scale_case(s1)
        ref = scale_boxes(s1, boxes.clone()[:, :4], s0, rp)
        out["scale"][name] = {"img1": s1, "img0": s0, "ratio_pad": rp, "in_sum": checksum(boxes), "out": ref.clone()}
        print("scale", name, float(ref.min()), float(ref.max()))
    torch.save(out, OUT / "val_edge.pt")


def gen_metrics_goldens(ns):
    """utils/metrics.py ap_per_class / compute_ap of the unmodified reference on seeded statistics (oracle.yolo_oracle.synth_ap_stats)."""
    import importlib

    rm = importlib.import_module("utils.metrics")
    out = {}
    for key, kw in [("mixed", dict(seed=3)), ("single_iou", dict(seed=4, n_iou=1)), ("few", dict(seed=5, n_det=7, n_lab=5, nc=3, absent=0)),
                    ("one_class", dict(seed=6, n_det=120, n_lab=40, nc=2, absent=0)), ("big", dict(seed=7, n_det=20000, n_lab=3000, nc=80, absent=5))]:
        tp, conf, pc, tc = yo.synth_ap_stats(**kw)
        res = rm.ap_per_class(tp, conf, pc, tc, plot=False, names={})   # (the reference default names=() has no .items(): val.py always passes a dict)
        out[key] = {"kw": kw, "in_sum": float(tp.sum() + conf.sum() + pc.sum() + tc.sum()), "out": [torch.as_tensor(r.copy()) for r in res]}
    r = torch.linspace(0, 0.83, 57).numpy() ** 1.5
    p = (1.0 - 0.6 * torch.linspace(0, 1, 57).numpy() ** 2) * (1 + 0.05 * torch.sin(torch.arange(57.0)).numpy())
    ap, mpre, mrec = rm.compute_ap(r, p)
    out["compute_ap"] = {"ap": float(ap), "mpre": torch.as_tensor(mpre.copy()), "mrec": torch.as_tensor(mrec.copy())}
    torch.save(out, OUT / "metrics.pt")
    print("metrics.pt", {k: (float(v["out"][5].mean()) if "out" in v else v["ap"]) for k, v in out.items()})


def gen_big_goldens(ns):
    """The benchmarked resolutions (416x416 tiny = BASELINE configs[0], 640x640 yolov3 = configs[1]/[3]): eval output of the
    unmodified reference.  Fixtures stay small: every STEP-th prediction row + whole-tensor statistics (sum |.|, max |.|)."""
    out = {}
    for name, nc, hw, bs, step in [("yolov3-tiny", 80, 416, 2, 4), ("yolov3", 80, 640, 1, 16), ("yolov3-spp", 80, 640, 1, 16)]:
        m, sd, layers, save, strides = build_ref_model(ns, name, nc, seed=11)
        x = torch.rand(bs, 3, hw, hw, generator=torch.Generator().manual_seed(5))
        m.eval()
        with torch.no_grad():
            pred, raw = m(x)
        out[f"{name}-nc{nc}-{hw}-bs{bs}"] = {
            "x_sum": checksum(x), "step": step, "pred_rows": pred[:, ::step].clone(), "pred_sum": checksum(pred), "pred_max": float(pred.abs().max()),
            "raw_sum": [checksum(t) for t in raw], "raw_max": [float(t.abs().max()) for t in raw],
            "raw_rows": [t.reshape(t.shape[0], -1, t.shape[-1])[:, ::step].clone() for t in raw],
        }
        print("big", name, hw, tuple(pred.shape), out[f"{name}-nc{nc}-{hw}-bs{bs}"]["pred_sum"])
    torch.save(out, OUT / "model_fwd_big.pt")


def gen_ckpt_fixture(ns):
    """A checkpoint pickled by the UNMODIFIED reference exactly as train.py:470-488 saves it ({'model': deepcopy(model).half(), ...}),
    for a width-0.25 yolov3-tiny (0.56 M parameters -> ~1.2 MB), plus the reference's own eval output of that checkpoint loaded the way
    models/experimental.py:88-136 attempt_load does (float, fuse, eval) on a rectangular batch."""
    from copy import deepcopy

    d = yaml.safe_load(open(CFG / "yolov3-tiny.yaml"))
    d["width_multiple"] = 0.25
    torch.manual_seed(7)
    m = ns.DetectionModel(d, ch=3, nc=80)
    g = torch.Generator().manual_seed(8)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.weight.data = torch.rand(mod.weight.shape, generator=g) + 0.5
            mod.bias.data = torch.randn(mod.bias.shape, generator=g) * 0.1
            mod.running_mean = torch.randn(mod.running_mean.shape, generator=g) * 0.1
            mod.running_var = torch.rand(mod.running_var.shape, generator=g) + 0.5
    m.names = {i: f"c{i}" for i in range(80)}
    ckpt = {"epoch": 5, "best_fitness": None, "model": deepcopy(m).half(), "ema": None, "updates": None, "optimizer": None, "opt": {}, "git": None, "date": "2026-09-21"}
    torch.save(ckpt, OUT / "ref_tiny_w025_fp16.pt")
    loaded = torch.load(OUT / "ref_tiny_w025_fp16.pt", map_location="cpu", weights_only=False)["model"].float().fuse().eval()   # attempt_load's steps
    x = torch.rand(2, 3, 96, 160, generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        pred, raw = loaded(x)
    torch.save({"x_sum": checksum(x), "pred": pred.clone(), "raw": [t.clone() for t in raw], "n_params": sum(p.numel() for p in m.parameters()),
                "stride": [float(s) for s in m.stride]}, OUT / "ref_tiny_w025_eval.pt")
    print("ckpt", (OUT / "ref_tiny_w025_fp16.pt").stat().st_size, tuple(pred.shape))


def gen_tta_goldens(ns):
    """model(x, augment=True) of the UNMODIFIED reference (models/yolo.py:239-276) on top of the restated upstream scale_img (oracle/upstream.py;
    ultralytics is not vendored), plus the scaled / mirrored inputs themselves (F.interpolate on this host)."""
    out = {}
    for name, nc, h, w, bs in [("yolov3-tiny", 20, 96, 160, 2), ("yolov3", 7, 128, 96, 1)]:
        m, sd, layers, save, strides = build_ref_model(ns, name, nc, seed=13)
        m.eval()
        x = torch.rand(bs, 3, h, w, generator=torch.Generator().manual_seed(8))
        with torch.no_grad():
            pred, none = m(x, augment=True)
        assert none is None
        gs = int(m.stride.max())
        out[f"{name}-nc{nc}-{h}x{w}-bs{bs}"] = {
            "x_sum": checksum(x),
            "pred": pred.clone(),
            "x_083_flip": upstream.scale_img(x.flip(3), 0.83, gs=gs).clone(),
            "x_067": upstream.scale_img(x, 0.67, gs=gs).clone(),
        }
        print("tta", name, tuple(pred.shape))
    torch.save(out, OUT / "tta.pt")


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ns = ref_shim.load()
    which = set(sys.argv[1:]) or {"model", "decode", "nms", "loss", "val_edge", "big", "ckpt", "metrics", "tta"}  # python make_golden.py [model decode nms loss val_edge big ckpt metrics tta]
    if "model" in which:
        gen_model_goldens(ns)
    if "decode" in which:
        gen_decode_goldens(ns)
    if "nms" in which:
        gen_nms_goldens(ns)
    if "loss" in which:
        gen_loss_goldens(ns)
    if "val_edge" in which:
        gen_val_edge_goldens(ns)
    if "metrics" in which:
        gen_metrics_goldens(ns)
    if "big" in which:
        gen_big_goldens(ns)
    if "ckpt" in which:
        gen_ckpt_fixture(ns)
    if "tta" in which:
        gen_tta_goldens(ns)
    print("golden files:", [(p.name, p.stat().st_size) for p in OUT.glob("*.pt")])
