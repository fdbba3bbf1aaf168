"""Host-logic test (CPU, no GPU): compile the static plan for each model and INTERPRET its symbolic steps with
torch CPU ops over the planned NHWC buffers (recycled storage, channel slices, fused upsample / zero-pad).
The result must match the oracle's graph walk, which proves the planner's aliasing, lifetimes and
concat offsets.  The HIP kernels themselves are covered by the -m gpu tests."""
import pytest
import torch
import torch.nn.functional as F
import yaml

from oracle import yolo_oracle as yo
from yolov3_amd import DetectionModel, engine, ops
from pathlib import Path

CFG = Path(__file__).resolve().parents[1] / "yolov3_amd" / "cfg"


def interpret(plan, x_nchw):
    xin = plan.input_view.real().as_nhwc()
    xin.zero_()
    xin[..., : x_nchw.shape[1]] = x_nchw.permute(0, 2, 3, 1)
    for kind, kw in plan.trace:
        if kind == "conv":
            w = kw["w"]
            wf, bf = w.folded
            x = kw["x"].real().as_nhwc()[..., : wf.shape[1]].permute(0, 3, 1, 2)
            y = F.conv2d(x, wf, bf, stride=w.s, padding=w.k // 2)
            if w.act:
                y = F.silu(y)
            if kw["res"] is not None:
                y = y + kw["res"].real().as_nhwc().permute(0, 3, 1, 2)
            if kw["ups"]:
                y = F.interpolate(y, scale_factor=2.0, mode="nearest")
            dst = kw["y"].real().as_nhwc()
            dst.zero_()
            dst[..., : y.shape[1]] = y.permute(0, 2, 3, 1)
        elif kind == "maxpool":
            x = kw["x"].real().as_nhwc().permute(0, 3, 1, 2)
            if kw["zr"] or kw["zb"]:
                x = F.pad(x, [0, kw["zr"], 0, kw["zb"]])
            kw["y"].real().as_nhwc().copy_(F.max_pool2d(x, kw["k"], kw["s"], kw["p"]).permute(0, 2, 3, 1))
        elif kind == "spp":
            x = kw["x"].real().as_nhwc().permute(0, 3, 1, 2)
            y = torch.cat([F.max_pool2d(x, k, 1, k // 2) for k in (5, 9, 13)], 1)
            kw["y"].real().as_nhwc().copy_(y.permute(0, 2, 3, 1))
        elif kind == "upsample":
            x = kw["x"].real().as_nhwc().permute(0, 3, 1, 2)
            kw["y"].real().as_nhwc().copy_(F.interpolate(x, scale_factor=2.0, mode="nearest").permute(0, 2, 3, 1))
        elif kind == "copy":
            kw["y"].real().as_nhwc().copy_(kw["x"].real().as_nhwc())
        else:
            raise AssertionError(kind)
    det, heads = plan.detect
    raws = []
    for hv in heads:
        t = hv.real().as_nhwc()[..., : det.na * det.no]
        raws.append(t.reshape(t.shape[0], t.shape[1], t.shape[2], det.na, det.no).permute(0, 3, 1, 2, 4).contiguous())
    return raws


@pytest.mark.parametrize("name,hw,nc", [("yolov3-tiny", 96, 80), ("yolov3", 64, 80), ("yolov3-spp", 64, 5)])
def test_plan_interpreted_on_cpu_matches_oracle(monkeypatch, name, hw, nc):
    monkeypatch.setattr(ops, "pack_filter", lambda w, cout, cin, dtype: torch.zeros(1))
    monkeypatch.setattr(engine, "KEEP_FOLDED", True)
    d = yaml.safe_load(open(CFG / f"{name}.yaml"))
    layers, save, anchors, nc_v = yo.parse_cfg(d, 3, nc)
    strides = yo.model_strides(layers)
    sd = yo.seeded_state_dict(layers, nc_v, anchors, strides, seed=4)
    m = DetectionModel(f"{name}.yaml", nc=nc).eval()
    m.load_state_dict(sd)
    x = torch.rand(2, 3, hw, hw + 32, generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        plan = engine.compile_model(m, 2, hw, hw + 32, torch.float32, torch.device("cpu"))
        raws = interpret(plan, x)
        _, ref = yo.forward(layers, save, sd, x, strides, training=False)
    assert len(raws) == len(ref)
    for a, b in zip(raws, ref):
        torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-4)
    # buffers really are recycled (fewer allocations than symbolic buffers)
    storages = {b.tensor.data_ptr() for b in plan.bufs if b.tensor is not None}
    assert len(storages) < sum(b.tensor is not None for b in plan.bufs)
    n_conv = sum(k == "conv" for k, _ in plan.trace)
    assert n_conv == {"yolov3-tiny": 13, "yolov3": 75, "yolov3-spp": 76}[name]
    assert not any(k in ("upsample", "copy") for k, _ in plan.trace)  # both fused away


def test_fuse_matches_unfused_plan_weights(monkeypatch):
    monkeypatch.setattr(ops, "pack_filter", lambda w, cout, cin, dtype: torch.zeros(1))
    monkeypatch.setattr(engine, "KEEP_FOLDED", True)
    m = DetectionModel("yolov3-tiny.yaml").eval()
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.normal_(0, 0.1)
            mod.running_var.uniform_(0.5, 1.5)
    with torch.no_grad():
        p1 = engine.compile_model(m, 1, 64, 64, torch.float32, torch.device("cpu"))
        keys_before = set(m.state_dict())
        m.fuse()
        p2 = engine.compile_model(m, 1, 64, 64, torch.float32, torch.device("cpu"))
    assert any(".bn." in k for k in keys_before) and not any(".bn." in k for k in m.state_dict())
    for (k1, a), (k2, b) in zip(p1.trace, p2.trace):
        if k1 == "conv":
            torch.testing.assert_close(a["w"].folded[0], b["w"].folded[0], rtol=1e-6, atol=1e-7)
            torch.testing.assert_close(a["w"].folded[1], b["w"].folded[1], rtol=1e-6, atol=1e-6)


def test_stem_pair_eligibility(monkeypatch):
    """Layers 0 + 1 are fused only for the Conv(<=4, 32, 3, 1) -> Conv(32, 64, 3, 2) opening of yolov3 / yolov3-spp with layer 0
    consumed by layer 1 alone; yolov3-tiny (Conv(3, 16) + MaxPool) keeps the stem kernel; Y3_STEM_PAIR=0 switches it off."""
    from yolov3_amd import DetectionModel
    from yolov3_amd import engine as e

    def eligible(name):
        m = DetectionModel(f"{name}.yaml").eval()
        layers = list(m.model)
        src = [e._sources(i, l.f) for i, l in enumerate(layers)]
        consumers = {i: [] for i in range(-1, len(layers))}
        for i, s in enumerate(src):
            for j in s:
                consumers[j].append(i)
        return e._stem_pair_eligible(layers, src, consumers, {}, {})

    monkeypatch.delenv("Y3_STEM_PAIR", raising=False)
    assert eligible("yolov3") and eligible("yolov3-spp") and not eligible("yolov3-tiny")
    monkeypatch.setenv("Y3_STEM_PAIR", "0")
    assert not eligible("yolov3")


def test_bneck_pair_eligibility(monkeypatch):
    """Bottleneck(64, 64) and Bottleneck(128, 128) (layers 2 and 4 of yolov3 / yolov3-spp) on half-precision views take the one-kernel form;
    the wider bottlenecks, fp32 plans and Y3_BNECK_PAIR=0 keep the two generic launches (Y3_BNECK_PAIR=64: only C = 64)."""
    from types import SimpleNamespace

    from yolov3_amd import DetectionModel
    from yolov3_amd import engine as e

    m = DetectionModel("yolov3.yaml").eval()
    b2, b4, b6 = m.model[2], m.model[4][0], m.model[6][0]
    def view(c, n=32, h=320, w=320):
        return SimpleNamespace(c=c, pitch=c, n=n, h=h, w=w)

    v64, v128, v256 = view(64), view(128, h=160, w=160), view(256, h=80, w=80)
    monkeypatch.delenv("Y3_BNECK_PAIR", raising=False)
    assert e._bneck_pair_eligible(b2, v64, torch.float16) and e._bneck_pair_eligible(b2, v64, torch.bfloat16)
    assert e._bneck_pair_eligible(b4, v128, torch.float16)
    assert not e._bneck_pair_eligible(b2, v64, torch.float32)
    assert not e._bneck_pair_eligible(b6, v256, torch.float16)
    assert not e._bneck_pair_eligible(b4, v64, torch.float16)      # channel count of the view and of the module disagree
    # a tensor beyond the 2 GiB reach of one buffer descriptor stays on the generic (batch-chunking) launches (round-2 advisor finding)
    assert e._bneck_pair_eligible(b2, view(64, n=163), torch.float16) and not e._bneck_pair_eligible(b2, view(64, n=164), torch.float16)
    assert not e._bneck_pair_eligible(b4, view(128, n=82, h=320, w=320), torch.float16)
    monkeypatch.setenv("Y3_BNECK_PAIR", "64")
    assert e._bneck_pair_eligible(b2, v64, torch.float16) and not e._bneck_pair_eligible(b4, v128, torch.float16)
    monkeypatch.setenv("Y3_BNECK_PAIR", "0")
    assert not e._bneck_pair_eligible(b2, v64, torch.float16)


def test_train_plan_static_analysis_on_cpu():
    """TrainPlan's build-time analysis needs no GPU: the one-launch filter packing registers every non-stem conv unit and the three
    Detect heads, with a data-gradient bank for every unit whose data gradient runs through the forward kernels; fp32 plans pack per layer."""
    import torch

    from yolov3_amd import DetectionModel
    from yolov3_amd.train_engine import ConvUnit, TrainPlan, TrainSlot

    m = DetectionModel("yolov3.yaml", nc=80).train()
    cpu = torch.device("cpu")
    slot = TrainSlot(torch.float16, cpu)
    p = TrainPlan.build(m, 1, 64, 64, torch.float16, cpu, slot)
    convs = [u for u in p.units if isinstance(u, ConvUnit)]
    assert len(convs) == 72
    jobs = p.pack_jobs.jobs
    assert len(jobs) == 71 + 3 and sum(1 for j in jobs if j[2] is not None) == 66 + 3   # 5 stride-2 3x3 units keep their parity-class banks
    assert all(u.bank_fwd is not None for u in convs if not u.use_stem) and convs[0].use_stem and convs[0].bank_fwd is None
    r = TrainPlan.build(m, 1, 64, 64, torch.float32, cpu, TrainSlot(torch.float32, cpu))
    assert r.pack_jobs is None
    # multi-scale training (reference train.py:394-399): every shape of a slot is a set of views into ONE arena sized for the largest shape seen, and packs into the
    # SAME filter banks -- a change of shape allocates nothing once the largest shape has run
    assert slot.arena_allocations == 1 and p.x_in.view.buf.data_ptr() >= slot.arena.data_ptr()
    big = TrainPlan.build(m, 1, 128, 128, torch.float16, cpu, slot, siblings=[p])
    assert slot.arena_allocations == 2 and big.slot_generation == slot.generation == p.slot_generation   # p was pointed at the new arena, not rebuilt
    lo, hi = slot.arena.data_ptr(), slot.arena.data_ptr() + slot.arena.numel()
    small = TrainPlan.build(m, 1, 96, 64, torch.float16, cpu, slot, siblings=[p, big])
    assert slot.arena_allocations == 2 and small.slot_generation == slot.generation and slot.arena.numel() == big._act_off
    for pl in (p, big, small):
        spans = sorted((a.view.buf.data_ptr(), a.view.buf.data_ptr() + a.view.buf.numel() * 2) for a in pl.acts if getattr(a, "parent", None) is None)
        spans += [(u.u.buf.data_ptr(), u.u.buf.data_ptr() + u.u.buf.numel() * 2) for u in pl.units if isinstance(u, ConvUnit)]
        spans.sort()
        assert all(lo <= a and b <= hi for a, b in spans) and all(b0 <= a1 for (_, b0), (a1, _) in zip(spans, spans[1:])), "views leave the arena or overlap"
    assert len(slot.pack_jobs.jobs) == 71 + 3 and big.units[1].bank_fwd is small.units[1].bank_fwd   # one set of banks for all shapes


def test_deferred_shortcut_gradient_state_machine():
    """train_engine.Act: a Bottleneck's shortcut gradient is noted, not stored (defer), and handed to the next producer (take_deferred); it cannot be deferred
    onto a tensor that already has a gradient, a channel slice, or another shape"""
    import torch

    from yolov3_amd.ops import View
    from yolov3_amd.train_engine import Act

    def view(n, h, w, c):
        return View(torch.zeros(n * h * w * c, dtype=torch.float16), n, h, w, c, c, 0)

    x, dy = Act(view(2, 4, 4, 16)), view(2, 4, 4, 16)
    assert not x.is_ready() and x.take_deferred() is None
    assert x.defer(dy) and x.is_ready() and x.deferred is dy
    assert not x.defer(dy), "one deferred contribution at a time"
    assert x.take_deferred() is dy and x.deferred is None and not x.is_ready()
    x.mark_ready()
    assert not x.defer(dy), "a stored gradient must be accumulated into"
    x.drop_grad()
    assert x.defer(dy)
    x.drop_grad()
    assert x.deferred is None and not x.is_ready()
    assert not Act(view(2, 4, 4, 8)).defer(dy), "shape mismatch"
    assert not x.slice(0, 8).defer(view(2, 4, 4, 8)), "slices of a concat buffer keep the stored form"


def test_train_plan_pairs_bn_consumers(monkeypatch):
    """Consumer-side BatchNorm (TrainPlan._pair_bn_consumers): a Conv unit immediately followed by the 1x1 Conv unit that reads its output hands its normalise + activation
    (+ shortcut) pass to that unit's launch where the library's input-transform form covers the shape -- in yolov3 the first cv1 of the 160 / 80 stages behind their
    stride-2 Conv, and cv2 -> the next cv1 inside the 160 / 80 stages and the last neck block; never across a Concat slice, never for the 3x3 consumers."""
    import torch

    from yolov3_amd import DetectionModel
    from yolov3_amd.train_engine import ConvUnit, TrainPlan, TrainSlot

    cpu = torch.device("cpu")
    m = DetectionModel("yolov3.yaml", nc=80).train()
    p = TrainPlan.build(m, 2, 640, 640, torch.float16, cpu, TrainSlot(torch.float16, cpu))
    pairs = [(u.bn_in.label, u.label, u.bn_in.res is not None) for u in p.units if isinstance(u, ConvUnit) and u.bn_in is not None]
    assert pairs[:4] == [("L3", "L4.0.cv1", False), ("L4.0.cv2", "L4.1.cv1", True), ("L5", "L6.0.cv1", False), ("L6.0.cv2", "L6.1.cv1", True)]   # (64 -> 32 behind layer 1: measured slower, not covered)
    assert len(pairs) == 12 and pairs[-2:] == [("L26.0.cv2", "L27.0.cv1", False), ("L27.0.cv2", "L27.1.cv1", False)]
    for u in p.units:
        if isinstance(u, ConvUnit) and u.bn_in is not None:
            assert u.k == 1 and u.bn_in.act_in_consumer and u.x is u.bn_in.y and u.bnin_rows > 0 and u.cin <= 256
    assert sum(1 for u in p.units if isinstance(u, ConvUnit) and u.act_in_consumer) == 12
    # fp32 plans and Y3_BN_IN_CONSUMER=0 keep the separate passes
    r = TrainPlan.build(m, 2, 640, 640, torch.float32, cpu, TrainSlot(torch.float32, cpu))
    assert not any(isinstance(u, ConvUnit) and u.bn_in is not None for u in r.units)
    monkeypatch.setenv("Y3_BN_IN_CONSUMER", "0")
    q = TrainPlan.build(m, 2, 640, 640, torch.float16, cpu, TrainSlot(torch.float16, cpu))
    assert not any(isinstance(u, ConvUnit) and (u.bn_in is not None or u.act_in_consumer) for u in q.units)
