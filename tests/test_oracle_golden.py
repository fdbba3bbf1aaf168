"""Pin oracle/yolo_oracle.py (the CPU restatement that travels to the GPU box) against golden
vectors produced by the UNMODIFIED reference (tests/golden/make_golden.py).  CPU only."""
import torch
import yaml
import pytest

from oracle import yolo_oracle as yo
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
CFG = ROOT / "yolov3_amd" / "cfg"


def checksum(t):
    return float(t.double().abs().sum())


def build(name, nc, seed):
    d = yaml.safe_load(open(CFG / f"{name}.yaml"))
    layers, save, anchors, nc_v = yo.parse_cfg(d, 3, nc)
    strides = yo.model_strides(layers)
    sd = yo.seeded_state_dict(layers, nc_v, anchors, strides, seed=seed)
    return layers, save, sd, strides


@pytest.mark.parametrize("key", ["yolov3-tiny-nc80-64-bs2", "yolov3-nc80-64-bs2", "yolov3-spp-nc80-64-bs1", "yolov3-nc7-96-bs1"])
def test_model_forward_matches_reference(golden_dir, key):
    gold = torch.load(golden_dir / "model_fwd.pt")[key]
    name, nc, hw, bs = key.rsplit("-", 3)
    nc, hw, bs = int(nc[2:]), int(hw), int(bs[2:])
    layers, save, sd, strides = build(name, nc, 11)
    x = torch.rand(bs, 3, hw, hw, generator=torch.Generator().manual_seed(5))
    assert checksum(x) == gold["x_sum"]
    assert abs(sum(checksum(v) for v in sd.values() if v.is_floating_point()) - gold["w_sum"]) < 1e-6 * gold["w_sum"]
    with torch.no_grad():
        tr = yo.forward(layers, save, sd, x, strides, training=True)
        pred, raw = yo.forward(layers, save, sd, x, strides, training=False)
        predf, _ = yo.forward(layers, save, yo.fuse_state_dict(sd), x, strides, training=False)
    for a, b in zip(tr, gold["train_raw"]):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(pred, gold["eval_pred"], rtol=1e-5, atol=1e-5)
    for a, b in zip(raw, gold["eval_raw"]):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(predf, gold["fused_pred"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("key", ["yolov3-tiny-nc80-416-bs2", "yolov3-nc80-640-bs1", "yolov3-spp-nc80-640-bs1"])
def test_model_forward_matches_reference_at_benchmark_resolutions(golden_dir, key):
    """the oracle at 416x416 (BASELINE configs[0]) and 640x640 (configs[1] / [3]) against the unmodified reference: every STEP-th
    prediction / raw row within 1e-5 (+ rounding of the big box coordinates), whole-tensor |.| sums within 1e-6 relative"""
    gold = torch.load(golden_dir / "model_fwd_big.pt")[key]
    name, nc, hw, bs = key.rsplit("-", 3)
    nc, hw, bs = int(nc[2:]), int(hw), int(bs[2:])
    layers, save, sd, strides = build(name, nc, 11)
    x = torch.rand(bs, 3, hw, hw, generator=torch.Generator().manual_seed(5))
    assert checksum(x) == gold["x_sum"]
    with torch.no_grad():
        pred, raw = yo.forward(layers, save, sd, x, strides, training=False)
    st = gold["step"]
    torch.testing.assert_close(pred[:, ::st], gold["pred_rows"], rtol=1e-5, atol=1e-4)
    assert abs(checksum(pred) - gold["pred_sum"]) < 1e-6 * gold["pred_sum"]
    for a, rows, sm in zip(raw, gold["raw_rows"], gold["raw_sum"]):
        torch.testing.assert_close(a.reshape(a.shape[0], -1, a.shape[-1])[:, ::st], rows, rtol=1e-5, atol=1e-5)
        assert abs(checksum(a) - sm) < 1e-6 * sm


def test_reference_checkpoint_fixture_loads_without_the_reference(golden_dir):
    """tests/golden/ref_tiny_w025_fp16.pt was pickled by the UNMODIFIED reference the way train.py:470-488 saves checkpoints.  It must
    unpickle through yolov3_amd.compat alone (no /root/reference on the GPU box) into our module classes, and the oracle fed its
    state dict must reproduce the reference's own eval output stored next to it."""
    from yolov3_amd import DetectionModel, compat

    try:
        m = compat.attempt_load(golden_dir / "ref_tiny_w025_fp16.pt", device="cpu", fuse=False)
    finally:
        compat.uninstall_aliases()   # the live-reference tests of this session import the real `models` package
    gold = torch.load(golden_dir / "ref_tiny_w025_eval.pt")
    assert type(m) is DetectionModel and not m.training and next(m.parameters()).dtype == torch.float32
    assert sum(p.numel() for p in m.parameters()) == gold["n_params"] and [float(s) for s in m.stride] == gold["stride"]
    d = yaml.safe_load(open(CFG / "yolov3-tiny.yaml"))
    d["width_multiple"] = 0.25
    layers, save, anchors, nc_v = yo.parse_cfg(d, 3, 80)
    x = torch.rand(2, 3, 96, 160, generator=torch.Generator().manual_seed(9))
    assert checksum(x) == gold["x_sum"]
    with torch.no_grad():
        pred, raw = yo.forward(layers, save, m.state_dict(), x, yo.model_strides(layers), training=False)
    torch.testing.assert_close(pred, gold["pred"], rtol=1e-4, atol=2e-4)   # the reference output is of the FUSED model: BN folded in fp32


@pytest.mark.parametrize("key,dtype,nc", [("nc80-float32", torch.float32, 80), ("nc80-float16", torch.float16, 80), ("nc3-float16", torch.float16, 3)])
def test_decode_matches_reference(golden_dir, key, dtype, nc):
    gold = torch.load(golden_dir / "decode.pt")[key]
    no = nc + 5
    g = torch.Generator().manual_seed(21)
    xs = [torch.randn(2, 3 * no, s, s + 1, generator=g) * 2.0 for s in gold["sizes"]]
    if dtype == torch.float16:
        xs = [x.half() for x in xs]
    raw = [x.view(2, 3, no, x.shape[2], x.shape[3]).permute(0, 1, 3, 4, 2).contiguous() for x in xs]
    z = yo.detect_decode(raw, gold["anchors_grid"].to(dtype), torch.tensor([8.0, 16.0, 32.0]).to(dtype))
    assert z.dtype == gold["z"].dtype
    assert torch.equal(z, gold["z"])


def _canon(t):
    """rows ordered by (-score, then x1,y1,x2,y2,cls): removes the arbitrary order the reference's unstable argsort
    gives to EXACT score ties (apriori label rows all have conf 1.0)"""
    t = t.float().cpu()
    keys = torch.stack((-t[:, 4], t[:, 0], t[:, 1], t[:, 2], t[:, 3], t[:, 5]), 1).tolist()
    order = sorted(range(len(keys)), key=lambda i: keys[i])
    return t[order]


def _cmp_nms(res, gold):
    assert len(res) == len(gold)
    for a, b in zip(res, gold):
        assert a.shape == b.shape
        assert torch.equal(a.float(), b.float())


def test_nms_known_answer(golden_dir):
    gold = torch.load(golden_dir / "nms.pt")
    p = torch.tensor(
        [[[50, 50, 20, 20, 0.9, 0.9, 0.5, 0.0], [200, 200, 30, 30, 0.8, 0.1, 0.2, 0.95], [52, 51, 20, 20, 0.7, 0.8, 0.6, 0.0], [400, 400, 10, 10, 0.0005, 0.9, 0.9, 0.9]]]
    )
    _cmp_nms(yo.non_max_suppression(p, 0.001, 0.6, multi_label=True), gold["kat_val"])
    _cmp_nms(yo.non_max_suppression(p, 0.25, 0.45), gold["kat_det"])
    _cmp_nms(yo.non_max_suppression(p, 0.25, 0.45, classes=[2]), gold["kat_cls2"])
    _cmp_nms(yo.non_max_suppression(p, 0.25, 0.45, agnostic=True), gold["kat_agn"])
    assert gold["kat_val"][0].shape == (5, 6) and gold["kat_det"][0].shape == (2, 6)


NMS_CASES = ["val_fp32", "det_fp32", "det_agnostic", "det_classes", "val_nc3_maxdet", "single_class", "det_fp16", "all_filtered"]


def nms_case_input(rec):
    gk = dict(rec["gen"])
    if "dtype" in gk:
        gk["dtype"] = getattr(torch, gk["dtype"].split(".")[-1])
    pred = yo.synth_predictions(**gk)
    assert checksum(pred) == rec["in_sum"]
    return pred


@pytest.mark.parametrize("name", NMS_CASES)
def test_nms_matches_reference(golden_dir, name):
    rec = torch.load(golden_dir / "nms.pt")[name]
    pred = nms_case_input(rec)
    _cmp_nms(yo.non_max_suppression(pred, **rec["nms"]), rec["out"])


def test_nms_labels_matches_reference(golden_dir):
    rec = torch.load(golden_dir / "nms.pt")["labels"]
    pred = yo.synth_predictions(bs=2, n_rows=800, nc=80, seed=10)
    res = yo.non_max_suppression(pred, 0.25, 0.45, labels=rec["lb"])
    _cmp_nms([_canon(r) for r in res], [_canon(r) for r in rec["out"]])


def test_nms_c_equals_numpy():
    from oracle import upstream
    g = torch.Generator().manual_seed(0)
    xy = torch.rand(400, 2, generator=g) * 100
    wh = torch.rand(400, 2, generator=g) * 40
    boxes = torch.cat((xy, xy + wh), 1)
    scores = torch.rand(400, generator=g)
    scores[10:20] = scores[5]  # ties -> stable order
    a = upstream.nms(boxes, scores, 0.4)
    b = torch.from_numpy(upstream.nms_py(boxes.numpy(), scores.numpy(), 0.4))
    assert torch.equal(a, b)


LOSS_CASES = ["yolov3-nc80-128-synth", "yolov3-tiny-nc80-96-synth", "yolov3-nc80-64-empty", "yolov3-nc5-64-dups", "yolov3-nc5-64-edges", "yolov3-nc5-64-dups_sorted"]


@pytest.mark.parametrize("key", LOSS_CASES)
def test_loss_matches_reference(golden_dir, key):
    rec = torch.load(golden_dir / "loss.pt")[key]
    name, nc, hw, mode = key.rsplit("-", 3)
    nc, hw = int(nc[2:]), int(hw)
    layers, save, sd, strides = build(name, nc, 13)
    bs = rec["bs"]
    p = [t.requires_grad_(True) for t in yo.synth_raw_predictions([(bs, 3, hw // s, hw // s, nc + 5) for s in strides], seed=31)]
    assert sum(checksum(t.detach()) for t in p) == rec["p_sum"]
    loss, items, _ = yo.compute_loss(p, rec["targets"], rec["anchors_grid"], rec["hyp"], nc, sort_obj_iou=mode.endswith("_sorted"))   # (ComputeLoss.sort_obj_iou, utils/loss.py:156-158)
    loss.backward()
    if mode == "dups_sorted":   # the case is only worth something when the order matters
        other = torch.load(golden_dir / "loss.pt")["yolov3-nc5-64-dups"]
        assert abs(float(other["items"][1]) - float(rec["items"][1])) > 5e-6
        assert any(not torch.equal(a, b) for a, b in zip(other["grads"], rec["grads"]))
    torch.testing.assert_close(loss, rec["loss"], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(items, rec["items"], rtol=1e-6, atol=1e-6)
    for a, b in zip(p, rec["grads"]):
        torch.testing.assert_close(a.grad, b, rtol=1e-5, atol=1e-8)


# ------------------------------------------------------------------------------------------------ output edge after NMS
def test_process_batch_matches_reference(golden_dir):
    """oracle.process_batch (restated val.py:147-188) against the unmodified reference's output, all seeded cases."""
    gold = torch.load(golden_dir / "val_edge.pt")["match"]
    iouv = torch.linspace(0.5, 0.95, 10)
    for name, rec in gold.items():
        det, lab = yo.synth_val_case(**rec["gen"])
        assert checksum(det) + checksum(lab) == rec["in_sum"], name
        got = yo.process_batch(det, lab, iouv) if det.shape[0] and lab.shape[0] else torch.zeros(det.shape[0], 10, dtype=torch.bool)
        assert torch.equal(got, rec["correct"]), name


def test_scale_boxes_matches_reference(golden_dir):
    """oracle.scale_boxes (restated utils/general.py:613-626 + upstream clip_boxes) bit-exact against the reference."""
    gold = torch.load(golden_dir / "val_edge.pt")["scale"]
    for name, rec in gold.items():
        boxes = yo.synth_scale_case(rec["img1"])
        assert checksum(boxes) == rec["in_sum"], name
        got = yo.scale_boxes(rec["img1"], boxes.clone()[:, :4], rec["img0"], rec["ratio_pad"])
        assert torch.equal(got, rec["out"]), name


# ------------------------------------------------------------------------------------------------ validation metrics (host, NumPy)
def test_ap_per_class_matches_reference_golden(golden_dir):
    """yolov3_amd.metrics.ap_per_class / compute_ap (the host half of val.py:416-421) against the UNMODIFIED reference's utils/metrics.py on
    seeded statistics: every returned array equal to 1e-12 (same NumPy operations in the same order), incl. classes with labels but no
    predictions, predictions without labels, a single IoU threshold and 20 000 detections over 80 classes."""
    import numpy as np

    from yolov3_amd import metrics

    gold = torch.load(golden_dir / "metrics.pt")
    for key, rec in gold.items():
        if key == "compute_ap":
            continue
        tp, conf, pc, tc = yo.synth_ap_stats(**rec["kw"])
        assert float(tp.sum() + conf.sum() + pc.sum() + tc.sum()) == rec["in_sum"], "seeded statistics drifted"
        res = metrics.ap_per_class(tp, conf, pc, tc, names={})
        assert len(res) == len(rec["out"]) == 7
        for got, ref in zip(res, rec["out"]):
            np.testing.assert_allclose(np.asarray(got, dtype=np.float64), ref.double().numpy(), rtol=0, atol=1e-12, err_msg=key)
    r = torch.linspace(0, 0.83, 57).numpy() ** 1.5
    p = (1.0 - 0.6 * torch.linspace(0, 1, 57).numpy() ** 2) * (1 + 0.05 * torch.sin(torch.arange(57.0)).numpy())
    ap, mpre, mrec = metrics.compute_ap(r, p)
    c = gold["compute_ap"]
    assert abs(ap - c["ap"]) < 1e-12
    np.testing.assert_allclose(mpre, c["mpre"].numpy(), atol=1e-12)
    np.testing.assert_allclose(mrec, c["mrec"].numpy(), atol=1e-12)
    # val.py:416-421 glue
    tp, conf, pc, tc = yo.synth_ap_stats(seed=3)
    mp, mr, m50, m5095 = metrics.mean_results([(tp[:300], conf[:300], pc[:300], tc[:90]), (tp[300:], conf[300:], pc[300:], tc[90:])])
    ref = gold["mixed"]["out"]
    assert abs(m50 - float(ref[5][:, 0].mean())) < 1e-12 and abs(m5095 - float(ref[5].mean(1).mean())) < 1e-12
    assert abs(mp - float(ref[2].mean())) < 1e-12 and abs(mr - float(ref[3].mean())) < 1e-12


TTA_KEYS = ["yolov3-tiny-nc20-96x160-bs2", "yolov3-nc7-128x96-bs1"]


def tta_case(key):
    name, nc, hw, bs = key.rsplit("-", 3)
    h, w = (int(v) for v in hw.split("x"))
    return name, int(nc[2:]), h, w, int(bs[2:])


@pytest.mark.parametrize("key", TTA_KEYS)
def test_augmented_forward_matches_reference(golden_dir, key):
    """oracle.forward_augment against model(x, augment=True) of the unmodified reference (models/yolo.py:239-276) on non-square inputs"""
    gold = torch.load(golden_dir / "tta.pt")[key]
    name, nc, h, w, bs = tta_case(key)
    layers, save, sd, strides = build(name, nc, 13)
    x = torch.rand(bs, 3, h, w, generator=torch.Generator().manual_seed(8))
    assert checksum(x) == gold["x_sum"]
    with torch.no_grad():
        pred = yo.forward_augment(layers, save, sd, x, strides)
    assert pred.shape == gold["pred"].shape
    torch.testing.assert_close(pred, gold["pred"], rtol=1e-5, atol=1e-5)
