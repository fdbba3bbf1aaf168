"""CPU tests of the host side: the C-ABI library loads and exports every symbol include/yolov3_hip.h declares,
argument validation fails loudly without touching a GPU, the module mirror keeps the reference's state_dict
layout, and the multi-process replica helpers work (gloo, world_size 2)."""
import ctypes as C
import os
import re
import subprocess
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def lib():
    from yolov3_amd import _lib, build

    build.build(verbose=False)
    return _lib.lib()


def test_library_exports_every_declared_symbol(lib):
    from yolov3_amd import _lib

    header = (ROOT / "include" / "yolov3_hip.h").read_text()
    declared = set(re.findall(r"\b(y3_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.exported_symbols()), declared ^ set(_lib.exported_symbols())
    nm = subprocess.check_output(["nm", "-D", "--defined-only", str(_lib.LIB_PATH)], text=True)
    exported = set(re.findall(r" T (y3_[a-z0-9_]+)", nm))
    assert declared <= exported, declared - exported
    assert lib.y3_abi_version() == 5 == _lib.ABI_VERSION


def test_argument_validation_is_loud_and_gpu_free(lib):
    from yolov3_amd import _lib

    d = _lib.Y3ConvDesc(_lib.Y3_F16, 5, 1, 1, 0, 0, 64, 64)
    t = _lib.Y3Tensor(1 << 20, 1, 8, 8, 64, 64)
    assert lib.y3_conv2d_fwd(C.byref(d), C.byref(t), 1 << 20, 1 << 20, None, C.byref(t), None) != 0
    assert b"ksize 5" in lib.y3_last_error()
    d.ksize = 3
    t2 = _lib.Y3Tensor(1 << 20, 1, 8, 8, 32, 32)
    assert lib.y3_conv2d_fwd(C.byref(d), C.byref(t2), 1 << 20, 1 << 20, None, C.byref(t), None) != 0
    assert b"channels" in lib.y3_last_error()
    # ABI 4: a bank shorter than y3_packed_filter_elems (e.g. sized rows x Kpad by the ABI-1 rule: no fragment-ordered second copy behind it) is refused before
    # anything could read past its end; a bias that is not 16-byte aligned (conv_v10.h loads it as f32x4) likewise
    d3 = _lib.Y3ConvDesc(_lib.Y3_F16, 3, 1, 1, 0, 0, 128, 256, 0, 256 * 1152)
    tx, ty = _lib.Y3Tensor(1 << 20, 2, 20, 20, 128, 128), _lib.Y3Tensor(1 << 20, 2, 20, 20, 256, 256)
    assert lib.y3_packed_filter_elems(256, 128, 3) == 2 * 256 * 1152
    assert lib.y3_conv2d_fwd(C.byref(d3), C.byref(tx), 1 << 20, 1 << 20, None, C.byref(ty), None) != 0
    assert b"y3_packed_filter_elems" in lib.y3_last_error()
    d3.filter_elems = 2 * 256 * 1152
    assert lib.y3_conv2d_fwd(C.byref(d3), C.byref(tx), 1 << 20, (1 << 20) + 4, None, C.byref(ty), None) != 0
    assert b"16-byte aligned" in lib.y3_last_error()
    p = _lib.Y3NmsParams(0.6, 1.5, 1, 0, 300, 30000, 7680.0, 0)
    assert lib.y3_nms(1 << 20, 0, 1, 10, 80, C.byref(p), None, 1 << 20, 1 << 20, 1 << 20, 0, 1 << 20, 1 << 20, None) != 0
    assert b"Invalid Confidence threshold" in lib.y3_last_error()
    assert lib.y3_packed_filter_elems(255, 1024, 1) == 256 * 1024
    assert lib.y3_packed_filter_elems(32, 8, 3) == 128 * 128
    p = _lib.Y3NmsParams(0.6, 0.001, 1, 0, 300, 30000, 7680.0, 0)
    assert lib.y3_nms_workspace_bytes(32, 25200, 80, C.byref(p), 0) > 32 * 16384 * 60


def test_product_path_rejects_cpu_tensors():
    from yolov3_amd import DetectionModel, non_max_suppression

    m = DetectionModel("yolov3-tiny.yaml").eval()
    with pytest.raises(RuntimeError, match="no CPU"):
        m(torch.zeros(1, 3, 64, 64))
    with pytest.raises(RuntimeError, match="no CPU"):
        non_max_suppression(torch.zeros(1, 10, 85))
    with pytest.raises(AssertionError, match="Invalid IoU"):
        non_max_suppression(torch.zeros(1, 10, 85), 0.5, 1.5)


def test_product_never_imports_the_oracle():
    code = "import sys; import yolov3_amd, yolov3_amd.engine, yolov3_amd.general, yolov3_amd.parallel; assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), 'oracle imported'"
    subprocess.check_call([sys.executable, "-c", code], cwd=ROOT)
    for f in (ROOT / "yolov3_amd").glob("*.py"):
        assert "oracle" not in f.read_text().replace("# oracle", ""), f


def test_state_dict_layout_matches_reference_keys(golden_dir):
    """keys/shapes must equal the reference's: checked against the oracle's seeded state dict, which
    tests/golden/make_golden.py loads strict=True into the unmodified reference model."""
    import yaml
    from oracle import yolo_oracle as yo
    from yolov3_amd import DetectionModel

    for name, npar in [("yolov3", 61949149), ("yolov3-spp", 62998749), ("yolov3-tiny", 8852366)]:
        m = DetectionModel(f"{name}.yaml")
        assert sum(p.numel() for p in m.parameters()) == npar
        d = yaml.safe_load(open(ROOT / "yolov3_amd" / "cfg" / f"{name}.yaml"))
        layers, save, anchors, nc = yo.parse_cfg(d)
        sd = yo.seeded_state_dict(layers, nc, anchors, yo.model_strides(layers), seed=0)
        own = m.state_dict()
        assert set(own) == set(sd)
        assert all(own[k].shape == sd[k].shape for k in sd)
        assert m.save == save
    m = DetectionModel("yolov3.yaml", nc=365)
    assert m.model[-1].no == 370 and m.model[-1].m[0].out_channels == 1110
    assert m.stride.tolist() == [8.0, 16.0, 32.0]
    assert torch.allclose(m.model[-1].anchors[0, 0], torch.tensor([10 / 8, 13 / 8]))
    bn = next(x for x in m.modules() if isinstance(x, torch.nn.BatchNorm2d))
    assert bn.eps == 1e-3 and bn.momentum == 0.03


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, str(ROOT))
    from yolov3_amd import parallel

    r, lr, w = parallel.init("gloo")
    parallel.barrier()
    mx = parallel.max_over_ranks(1.0 + r)
    lo, hi = parallel.shard_range(65, r, w)
    g = parallel.describe()   # the process-group record bench.py attaches to its line: what the communicator saw
    q.put((r, w, mx, lo, hi, g["backend"], g["world_size"], sorted(x["rank"] for x in g["ranks"]), len({x["pid"] for x in g["ranks"]})))
    parallel.finalize()


def test_replica_helpers_world2_gloo():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in range(2))
    [p.join(60) for p in procs]
    assert [r[:3] for r in res] == [(0, 2, 2.0), (1, 2, 2.0)]
    assert (res[0][3], res[0][4], res[1][3], res[1][4]) == (0, 33, 33, 65)
    assert all(r[5:] == ("gloo", 2, [0, 1], 2) for r in res), res   # both ranks report the same two-process group


_DDP_SHAPES = [("a", (3, 5)), ("b", (7,)), ("c", (2, 2, 2)), ("d", (33,))]


def _ddp_worker(rank, world, port, q, exchange, arena):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, str(ROOT))
    import torch
    from yolov3_amd import parallel

    parallel.init("gloo")
    torch.manual_seed(rank)
    lin = torch.nn.Linear(8, 4)
    parallel.broadcast_parameters(lin, src=0)
    gb = parallel.GradBuckets(bucket_bytes=64, exchange=exchange)  # tiny buckets -> several collectives
    g = torch.Generator().manual_seed(100 + rank)
    grads = {name: torch.randn(shape, generator=g) for name, shape in _DDP_SHAPES}
    if arena:   # the training plan's layout: 64-element slices of one flat fp32 tensor, handed over in the order they sit in it -> reduced in place
        flat, off, views = torch.full((64 * 8,), float("nan")), 0, {}
        for k in ["d", "c", "b", "a"]:
            n = grads[k].numel()
            views[k] = flat[off : off + n].view(grads[k].shape)
            views[k].copy_(grads[k])
            off += (n + 63) // 64 * 64
        grads = views
    for k in ["d", "c", "b", "a"]:  # reverse layer order, like the backward plan
        gb.add(k, grads[k])
    out = gb.finish()
    in_place = all(out[k].data_ptr() == grads[k].data_ptr() for k in out)
    q.put((rank, {k: v.tolist() for k, v in out.items()}, lin.weight.detach().tolist(), dict(gb.collectives), in_place))
    parallel.finalize()


@pytest.mark.parametrize("world,exchange,arena", [(2, "all_reduce", False), (2, "all_reduce", True), (2, "direct", False), (2, "direct", True), (3, "direct", False)])
def test_gradient_buckets_average_gloo(world, exchange, arena):
    """the data-parallel exchange step (bucketed average, DDP semantics) on 2-3 CPU ranks: the all-reduce form and the two-phase direct form
    (all-to-all of the shards, owner's sum, all-gather), flattened buckets and in-place ranges of a gradient arena"""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() * 7 + world * 3 + len(exchange) + arena) % 2000
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, q, exchange, arena)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda r: r[0])
    [p.join(60) for p in procs]
    gens = [torch.Generator().manual_seed(100 + r) for r in range(world)]
    per_rank = [{name: torch.randn(shape, generator=g) for name, shape in _DDP_SHAPES} for g in gens]
    for k in "abcd":
        tot = per_rank[0][k].clone()
        for r in range(1, world):
            tot += per_rank[r][k]            # rank order: what the direct form's owner computes
        mean = tot * (1.0 / world) if exchange == "direct" else tot / world
        for r in range(world):
            got = torch.tensor(res[r][1][k])
            if world == 2 or exchange == "direct":
                assert torch.equal(got, mean), k
            else:
                torch.testing.assert_close(got, mean)
            assert res[r][1][k] == res[0][1][k]   # replicas stay bit-identical
    assert all(r[2] == res[0][2] for r in res)  # parameters broadcast from rank 0
    for r in res:
        assert r[3][exchange] >= 2 and r[3]["direct" if exchange == "all_reduce" else "all_reduce"] == 0, r[3]   # every bucket took the requested form
        assert r[4] == arena   # arena slices are averaged where they lie; loose tensors come back as views of the flattened bucket


def test_gradient_buckets_reject_unknown_exchange():
    from yolov3_amd import parallel

    with pytest.raises(ValueError, match="exchange"):
        parallel.GradBuckets(exchange="ring")


def test_reference_checkpoint_unpickles_into_hip_modules(tmp_path):
    """a ``.pt`` written by the UNMODIFIED reference (pickled models.yolo.DetectionModel, as train.py:470-488 does)
    loads through yolov3_amd.compat into MI355X-backed modules with identical parameters.  Needs /root/reference."""
    from oracle import ref_shim

    if not ref_shim.available():
        pytest.skip("reference tree not present on this box")
    ckpt = tmp_path / "ref_tiny.pt"
    writer = f"""
import sys, torch
sys.path.insert(0, {str(ROOT)!r})
from oracle import ref_shim
ns = ref_shim.load()
m = ns.DetectionModel({str(ROOT / 'yolov3_amd' / 'cfg' / 'yolov3-tiny.yaml')!r}, ch=3, nc=80)
for mod in m.modules():
    if isinstance(mod, torch.nn.BatchNorm2d):
        mod.running_mean.normal_(0, 0.1); mod.running_var.uniform_(0.5, 1.5)
torch.save({{"epoch": 3, "model": m.half(), "ema": None, "optimizer": None}}, {str(ckpt)!r})
torch.save(m.float().state_dict(), {str(tmp_path / 'sd.pt')!r})
"""
    subprocess.check_call([sys.executable, "-c", writer], cwd=tmp_path)
    reader = f"""
import sys, torch
sys.path.insert(0, {str(ROOT)!r})
from yolov3_amd import compat, DetectionModel, Detect
from yolov3_amd.common import Conv, MaxPool2d, Upsample, ZeroPad2d
m = compat.attempt_load({str(ckpt)!r}, device="cpu", fuse=False)
assert type(m) is DetectionModel and type(m.model[-1]) is Detect and type(m.model[0]) is Conv, type(m)
assert any(isinstance(x, MaxPool2d) for x in m.model) and any(isinstance(x, Upsample) for x in m.model) and any(isinstance(x, ZeroPad2d) for x in m.model)
sd = torch.load({str(tmp_path / 'sd.pt')!r})
own = m.state_dict()
assert set(own) == set(sd)
assert all(torch.equal(own[k].float(), sd[k].half().float()) for k in sd if sd[k].is_floating_point())
assert m.stride.tolist() == [16.0, 32.0] and m.save == [8, 14, 15, 19]
f = compat.attempt_load({str(ckpt)!r}, device="cpu", fuse=True)
assert not any('.bn.' in k for k in f.state_dict()) and not f.training
print("ok")
"""
    out = subprocess.check_output([sys.executable, "-c", reader], cwd=tmp_path, text=True)
    assert out.strip().endswith("ok")



def _run_py(code, cwd):
    return subprocess.check_output([sys.executable, "-c", code], cwd=cwd, text=True, stderr=subprocess.STDOUT)


def test_attempt_load_inside_a_live_reference_process(tmp_path):
    """the drop-in scenario itself (round-5 review, weak 1a): a process that has the REAL reference modules imported.
    (i) real ``models.yolo`` imported first (train.py:50-51, hubconf.py:48): attempt_load still returns yolov3_amd classes and the real
    modules stay what they were; (ii) only the real ``models.common`` imported (val.py:39, detect.py:46): attempt_load works and
    ``from models.common import AutoShape, DetectMultiBackend`` (utils/general.py:432) still resolves to the reference's own;
    (iii) an object of the reference's classes (built by the reference, or unpickled by its own torch.load) is re-classed by
    compat.adopt and its parameters are untouched."""
    from oracle import ref_shim

    if not ref_shim.available():
        pytest.skip("reference tree not present on this box")
    ckpt = ROOT / "tests" / "golden" / "ref_tiny_w025_fp16.pt"
    code = f"""
import sys, torch
sys.path.insert(0, {str(ROOT)!r})
from oracle import ref_shim
ns = ref_shim.load()                      # imports the unmodified models.yolo / models.common
import models.yolo as real_yolo, models.common as real_common
assert real_yolo.__file__.startswith('/root/reference')
from yolov3_amd import compat, DetectionModel, Detect
from yolov3_amd.common import Conv
before = (sys.modules['models'], sys.modules['models.yolo'], sys.modules['models.common'])
assert compat.install_aliases() is False and compat.install_aliases(force=True) is False
m = compat.attempt_load({str(ckpt)!r}, device='cpu', fuse=False)
assert type(m) is DetectionModel and type(m.model[-1]) is Detect and type(m.model[0]) is Conv, type(m)
assert before == (sys.modules['models'], sys.modules['models.yolo'], sys.modules['models.common'])
from models.common import AutoShape, DetectMultiBackend
assert AutoShape.__module__ == 'models.common' and sys.modules['models.common'].__file__.startswith('/root/reference')
# (iii) the reference's own torch.load gives the reference's classes; adopt re-classes them in place
ref_obj = torch.load({str(ckpt)!r}, map_location='cpu', weights_only=False)['model']
assert type(ref_obj) is real_yolo.DetectionModel
sd = {{k: v.clone() for k, v in ref_obj.state_dict().items()}}
ours = compat.adopt(ref_obj)
assert type(ours) is DetectionModel and all(not type(x).__module__.startswith('models.') for x in ours.modules())
assert set(sd) == set(ours.state_dict()) and all(torch.equal(sd[k], v) for k, v in ours.state_dict().items())
assert type(ours.float().fuse()) is DetectionModel and not any('.bn.' in k for k in ours.state_dict())
print('ok')
"""
    assert _run_py(code, tmp_path).strip().endswith("ok")
    code2 = f"""
import sys, torch
sys.path.insert(0, {str(ROOT)!r})
from oracle import ref_shim
ref_shim.install()                        # stubs for the absent third-party packages + /root/reference on sys.path
import models.common as real_common       # val.py:39 / detect.py:46 import models.common only
assert 'models.yolo' not in sys.modules
from yolov3_amd import compat, DetectionModel
m = compat.attempt_load({str(ckpt)!r}, device='cpu')
assert type(m) is DetectionModel
assert sys.modules['models.common'] is real_common and 'models.yolo' not in sys.modules
from models.common import AutoShape, DetectMultiBackend
assert AutoShape.__module__ == 'models.common' and real_common.__file__.startswith('/root/reference')
print('ok')
"""
    assert _run_py(code2, tmp_path).strip().endswith("ok")


def test_alias_modules_only_in_a_process_without_the_reference(tmp_path):
    """install_aliases is for plain torch.load in a bare process (the GPU box): there the aliases resolve the pickled paths, carry the
    names utils/general.py:432 imports, and uninstall cleanly; attempt_load itself never touches sys.modules."""
    ckpt = ROOT / "tests" / "golden" / "ref_tiny_w025_fp16.pt"
    code = f"""
import sys, torch
sys.path.insert(0, {str(ROOT)!r})
from yolov3_amd import compat, DetectionModel
m = compat.attempt_load({str(ckpt)!r}, device='cpu')
assert type(m) is DetectionModel and not any(k == 'models' or k.startswith('models.') for k in sys.modules)
assert compat.install_aliases() is True
from models.common import AutoShape, DetectMultiBackend, Conv
from models.experimental import attempt_load
obj = torch.load({str(ckpt)!r}, map_location='cpu', weights_only=False)['model']
assert type(obj) is DetectionModel
compat.uninstall_aliases()
assert not any(k == 'models' or k.startswith('models.') for k in sys.modules)
print('ok')
"""
    assert _run_py(code, tmp_path).strip().endswith("ok")


# ------------------------------------------------------------------------------------------------ host logic of the val / detect edges
def test_letterbox_geometry_matches_oracle_letterbox():
    """autoshape.letterbox_geometry (what the device letterbox is launched with) against the oracle's restatement of reference
    utils/augmentations.py:104-134 on many shapes: output size, resized size, top/left border, ratio and (dw, dh)."""
    import numpy as np

    from oracle import yolo_oracle as yo
    from yolov3_amd.autoshape import letterbox_geometry

    rng = np.random.default_rng(0)
    for _ in range(200):
        h0, w0 = int(rng.integers(8, 900)), int(rng.integers(8, 900))
        s = int(rng.choice([320, 416, 640]))
        g = s / max(h0, w0)
        shape1 = [int(np.ceil(int(v * g) / 32) * 32) for v in (h0, w0)]   # AutoShape's inference shape for a single image
        im = np.zeros((h0, w0, 3), np.uint8)
        im[...] = 7
        want, ratio, pad = yo.letterbox(im, shape1, auto=False)
        nh, nw, top, left, r, dwdh, full = letterbox_geometry((h0, w0), shape1)
        assert tuple(full) == want.shape[:2] == tuple(shape1)
        assert r == ratio and dwdh == pad
        inner = want[top : top + nh, left : left + nw]
        assert inner.shape[:2] == (nh, nw) and (inner == 7).all(), "resized image is not where the geometry says"
        border = want.copy()
        border[top : top + nh, left : left + nw] = 114
        assert (border == 114).all()


def test_scale_boxes_cpu_path_and_gain_pad_match_oracle():
    """general.scale_boxes on CPU tensors (the reference's tensor arithmetic, kept for label tensors) and general._gain_pad
    (the parameters handed to y3_scale_boxes) against the oracle."""
    import torch

    from oracle import yolo_oracle as yo
    from yolov3_amd import general

    boxes = yo.synth_scale_case((640, 640))
    for s0, rp in [((480, 640), None), ((1280, 720), None), ((427, 640), ((0.9, 0.9), (0.0, 0.15)))]:
        want = yo.scale_boxes((640, 640), boxes.clone()[:, :4], s0, rp)
        got = general.scale_boxes((640, 640), boxes.clone()[:, :4], s0, rp)
        assert torch.equal(got, want)
        gain, px, py = general._gain_pad((640, 640), s0, rp)
        manual = boxes.clone()[:, :4]
        manual[:, [0, 2]] -= px
        manual[:, [1, 3]] -= py
        manual /= gain
        manual[:, [0, 2]] = manual[:, [0, 2]].clamp(0, s0[1])
        manual[:, [1, 3]] = manual[:, [1, 3]].clamp(0, s0[0])
        assert torch.equal(manual, want)


# ------------------------------------------------------------------------------------------------ launch geometry through the C ABI (no GPU needed)
def _desc(dtype_code, k, s, cin, cout):
    from yolov3_amd._lib import Y3ConvDesc

    return Y3ConvDesc(dtype_code, k, s, 0, 0, 0, cin, cout, 0)


def test_plans_live_outside_the_module_and_are_bounded():
    """compiled plans hold ctypes blocks + device memory and must never be part of the module's copied / pickled state
    (deepcopy(model), torch.save of the model, ModelEMA: reference train.py:470-488); the cache is LRU-bounded per mode."""
    import copy
    import io

    import torch

    from yolov3_amd import DetectionModel
    from yolov3_amd.engine import PlanCache, plan_cache

    m = DetectionModel("yolov3-tiny.yaml", nc=3)
    pc = plan_cache(m)
    import ctypes as C
    for i in range(PlanCache.MAX_EVAL + 3):
        pc.put(("eval", 1, 32 * (i + 1), 32, torch.float16, 0, 0), C.byref(C.c_int(i)))   # un-picklable, like a real plan's argument blocks
    for s_ in range(4):
        pc.put(("train", 1, 64, 64, torch.float16, 0, s_), object())
    kinds = [k[0] for k in pc.plans]
    assert kinds.count("eval") == PlanCache.MAX_EVAL and kinds.count("train") == PlanCache.MAX_TRAIN
    # training plans are capped per SHAPE (slots of outstanding forwards) and by the number of shapes, least recently used shape first
    assert PlanCache.MAX_TRAIN_SHAPES >= 24   # the reference's multi-scale training draws from ~21 sizes (train.py:394-399): all of them stay compiled
    last = 64 + 32 * PlanCache.MAX_TRAIN_SHAPES
    for hw in range(96, last + 1, 32):
        pc.put(("train", 1, hw, hw, torch.float16, 0, 0), object())
    tshapes = [k[1:6] for k in pc.plans if k[0] == "train"]
    assert len(set(tshapes)) == PlanCache.MAX_TRAIN_SHAPES and (1, 64, 64, torch.float16, 0) not in tshapes and (1, last, last, torch.float16, 0) in tshapes
    assert [k[0] for k in pc.plans].count("eval") == PlanCache.MAX_EVAL
    assert ("eval", 1, 32, 32, torch.float16, 0, 0) not in pc.plans            # the oldest went first
    assert len(m._plans) == len(pc.plans) and "_plans" not in m.__dict__
    m2 = copy.deepcopy(m)                                                       # used to raise: cannot pickle 'CArgObject'
    assert len(m2._plans) == 0
    buf = io.BytesIO()
    torch.save({"model": copy.deepcopy(m).half(), "ema": None}, buf)          # the reference's checkpoint layout
    m.fuse()
    assert len(m._plans) == 0                                                   # fuse / .to() / .half() drop plans and packed filters
    pc.put(("eval", 1, 32, 32, torch.float16, 0, 0), object())
    m.half()
    assert len(m._plans) == 0


def test_conv_dispatch_variant_names_and_stat_rows():
    """y3_conv2d_fwd_variant / y3_conv2d_fwd_stats_rows are dry runs of the conv dispatcher (no launch).  The variant name is what
    bench.py groups its roofline by and what the GPU parity tests assert; the statistic rows must equal (pixel tiles of that
    variant) x (its rows per tile) -- checked on every conv shape of yolov3 at batch 32 and 64, with and without a workspace."""
    import ctypes as C

    from yolov3_amd import _lib
    from yolov3_amd._lib import Y3Tensor

    L = _lib.lib()
    ws = L.y3_conv_workspace_bytes()
    assert ws == 64 + 4096 + 2 * 256 * 256 * 256 * 4
    # (pixels per tile, statistics rows per tile); v10 cuts the pixel axis into 32-pixel column blocks and a block's run of them into tiles of 6 / 7 / 8 (conv_v10.h)
    tile_px = {"v6": (256, 2), "v3_bk64_128x128": (128, 2), "v3_bk32_128x128": (128, 2), "v3_bk32_128x256": (256, 2), "v3_bk32_64x256": (256, 2),
               "v10": (0, 4), "v10h": (0, 4), "strip": (0, 0), "s1x1": (64, 1)}   # s1x1 (conv_1x1s.h): one row per epilogue pass of 64 pixels (whole stages here)
    shapes = [(32, 64, 3, 2, 640), (64, 32, 1, 1, 320), (32, 64, 3, 1, 320), (64, 128, 3, 2, 320), (128, 64, 1, 1, 160), (64, 128, 3, 1, 160), (128, 256, 3, 2, 160),
              (256, 128, 1, 1, 80), (128, 256, 3, 1, 80), (256, 512, 3, 2, 80), (512, 256, 1, 1, 40), (256, 512, 3, 1, 40), (512, 1024, 3, 2, 40), (1024, 512, 1, 1, 20),
              (512, 1024, 3, 1, 20), (768, 256, 1, 1, 40), (384, 128, 1, 1, 80), (256, 256, 1, 1, 80), (512, 256, 1, 1, 20)]
    v10_shapes = {(128, 256, 3, 1, 80), (256, 512, 3, 1, 40), (512, 1024, 3, 1, 20)}   # 3x3, stride 1, cin >= 128, cout % 256 == 0: the persistent one-wave-per-SIMD kernel (needs no workspace)
    # tiles per filter tile = sum over its blocks of the tiles per block.  Cin <= 256: two half-size blocks per CU with tiles of 3 / 4 column blocks ("v10h": runs of
    # 12.5 / 6.25 and 25 / 12.5 column blocks); 512 -> 1024: one block per CU with tiles of 6 / 7 / 8 (runs of 6.25 / 12.5)
    v10_tiles = {(32, 80): 256 * 4 + 256 * 3, (32, 40): 256 * 2, (32, 20): 64 * 1, (64, 80): 512 * 7, (64, 40): 128 * 4 + 128 * 3, (64, 20): 64 * 2}
    for bs in (32, 64):
        for cin, cout, k, s, hin in shapes:
            ho = (hin + 2 * (k // 2) - k) // s + 1
            x = Y3Tensor(4096, bs, hin, hin, cin, cin)      # fake, 16-byte aligned device addresses: a dry run never dereferences them
            y = Y3Tensor(8192, bs, ho, ho, cout, cout)
            d = _desc(_lib.Y3_F16, k, s, cin, cout)
            name = C.create_string_buffer(64)
            assert L.y3_conv2d_fwd_variant(C.byref(d), C.byref(x), C.byref(y), 0, 0, name, 64) == 0, L.y3_last_error()
            plain = name.value.decode()
            assert plain in tile_px, plain
            # the HBM-bound 1x1 layers (Cin <= 384 at >= 32768 pixels): the persistent kernel with register-resident filters
            assert (plain == "s1x1") == (k == 1 and cin <= 384), (cin, cout, k, s, hin, plain)
            assert (plain in ("v10", "v10h")) == ((cin, cout, k, s, hin) in v10_shapes) and (plain == "v10h") == ((cin, cout, k, s, hin) in v10_shapes and cin <= 256), (cin, cout, k, s, hin, plain)
            rows = L.y3_conv2d_fwd_stats_rows(C.byref(d), C.byref(x), C.byref(y))
            assert rows > 0, L.y3_last_error()
            tp, per = tile_px[plain]
            m = bs * ho * ho
            if (cin, cout, k) == (64, 128, 3):
                # the strip kernel with register-resident filters (conv_strip.h, profiles/r03_conv_strip_ab.txt; knob conv_strip = 0: the tiles of the round-2
                # sweep, profiles/r02_conv_variant_sweep.txt)
                assert plain == "strip"
            if plain == "strip":
                assert rows == 256 * 2, rows   # 256 persistent blocks (one per CU: 8 waves, 144 filter registers per lane), two pixel tiles each
                assert L.y3_tune_set(b"conv_strip", 0) == 0
                try:
                    assert L.y3_conv2d_fwd_variant(C.byref(d), C.byref(x), C.byref(y), 0, 0, name, 64) == 0
                    assert name.value == (b"v3_bk32_128x256" if s == 1 else b"v3_bk64_128x128")
                finally:
                    L.y3_tune_reset()
                continue
            if plain in ("v10", "v10h"):
                assert rows == v10_tiles[(bs, hin)] * per, f"{cin}->{cout} @{hin} bs{bs}: {rows} rows"
            else:
                assert rows == -(-m // tp) * per, f"{cin}->{cout} k{k} s{s} @{hin} bs{bs}: {rows} rows for {plain}"
            assert L.y3_conv2d_fwd_variant(C.byref(d), C.byref(x), C.byref(y), 0, ws, name, 64) == 0
            with_ws = name.value.decode()
            assert with_ws == plain, (cin, cout, k, s, hin, with_ws)
            if (cin, cout, k, s, hin) in v10_shapes and cin >= 256:   # knob conv_v10 = 0: the 256 x 256 tile kernel of round 1, with or without a workspace
                assert L.y3_tune_set(b"conv_v10", 0) == 0
                try:
                    assert L.y3_conv2d_fwd_variant(C.byref(d), C.byref(x), C.byref(y), 0, ws, name, 64) == 0 and name.value == b"v6"
                    assert L.y3_conv2d_fwd_variant(C.byref(d), C.byref(x), C.byref(y), 0, 0, name, 64) == 0 and name.value == b"v6"
                finally:
                    L.y3_tune_reset()
    # small launches (below a quarter round of 256-pixel tiles): with a workspace the K-split form of conv_v10.h ("v10k": statistics rows = 64-pixel blocks of the
    # slab sum), without one the tile kernels
    for bs, cin, cout, hin in [(1, 512, 1024, 20), (4, 256, 512, 40), (2, 128, 256, 80)]:
        x, y = Y3Tensor(4096, bs, hin, hin, cin, cin), Y3Tensor(8192, bs, hin, hin, cout, cout)
        d = _desc(_lib.Y3_F16, 3, 1, cin, cout)
        name = C.create_string_buffer(64)
        assert L.y3_conv2d_fwd_variant(C.byref(d), C.byref(x), C.byref(y), 0, ws, name, 64) == 0 and name.value == b"v10k", name.value
        assert L.y3_conv2d_fwd_stats_rows_ws(C.byref(d), C.byref(x), C.byref(y), ws) == -(-bs * hin * hin // 64)
        assert L.y3_conv2d_fwd_variant(C.byref(d), C.byref(x), C.byref(y), 0, 0, name, 64) == 0 and name.value in (b"v6", b"v3_bk64_128x128"), name.value
    # the training-mode launches of the small-channel 3x3 / stride-1 layers (conv_strip.h): 64 -> 32 and 128 -> 64 are the data gradients of layers 2.cv2 / 4.x.cv2;
    # rows = blocks x pixel tiles (x 2 K-split waves at Cin = 128); small launches and residual launches stay on the tile kernels
    for (cin, cout, hin, rows_want) in [(64, 32, 320, 768 * 2), (128, 64, 160, 256 * 2 * 2)]:
        x, y = Y3Tensor(4096, 64, hin, hin, cin, cin), Y3Tensor(8192, 64, hin, hin, cout, cout)
        d = _desc(_lib.Y3_F16, 3, 1, cin, cout)
        assert L.y3_conv2d_fwd_variant(C.byref(d), C.byref(x), C.byref(y), 0, 0, name, 64) == 0 and name.value == b"strip", name.value
        rows = L.y3_conv2d_fwd_stats_rows(C.byref(d), C.byref(x), C.byref(y))
        assert abs(rows - rows_want) <= rows_want // 50, (cin, cout, rows)
        assert L.y3_conv2d_fwd_variant(C.byref(d), C.byref(x), C.byref(y), 1, 0, name, 64) == 0 and name.value != b"strip"   # with a residual
        xs, ys = Y3Tensor(4096, 2, 64, 64, cin, cin), Y3Tensor(8192, 2, 64, 64, cout, cout)
        assert L.y3_conv2d_fwd_variant(C.byref(d), C.byref(xs), C.byref(ys), 0, 0, name, 64) == 0 and name.value != b"strip"  # too few rows per block
    # fp32 -> the direct kernel; a too-small workspace never selects the persistent kernel
    x, y = Y3Tensor(4096, 2, 20, 20, 512, 512), Y3Tensor(8192, 2, 20, 20, 1024, 1024)
    name = C.create_string_buffer(64)
    assert L.y3_conv2d_fwd_variant(C.byref(_desc(_lib.Y3_F32, 3, 1, 512, 1024)), C.byref(x), C.byref(y), 0, ws, name, 64) == 0 and name.value == b"direct"
    assert L.y3_conv2d_fwd_variant(C.byref(_desc(_lib.Y3_F16, 3, 1, 512, 1024)), C.byref(x), C.byref(y), 0, 4096, name, 64) == 0 and name.value == b"v6"


def test_wgrad_geometry_fills_whole_rounds():
    """y3_conv2d_wgrad_plan / _workspace_bytes expose the filter-gradient launch geometry: the padded-position kernel for the stride-1 long-K layers of
    yolov3 at batch 64 (one block per CU), the 256x256-tile kernel for the stride-2 ones (its tiles x pixel slices fill 1 or 2 rounds of 256 CUs to >= 98 %);
    the other layers keep 128x128 tiles (64 KiB partial tiles)."""
    import ctypes as C

    from yolov3_amd import _lib
    from yolov3_amd._lib import Y3Tensor

    L = _lib.lib()
    tile, slices, xg = C.c_int32(0), C.c_int64(0), C.c_int32(0)
    # round 6: the stride-1 long-K layers run the padded-position kernel (csrc/wgrad_patch.h, plan code 4): (Cout / 128)(Cin / 64) block tiles of 128 x 576 accumulators,
    # as many position slices as give one block per CU, one 288 KiB slab per block
    for (cin, cout, hw), (tiles, sl) in {(128, 256, 80): (4, 64), (256, 512, 40): (16, 16), (512, 1024, 20): (64, 4)}.items():
        x = Y3Tensor(4096, 64, hw, hw, cin, cin)
        d = _desc(_lib.Y3_F16, 3, 1, cin, cout)
        assert L.y3_conv2d_wgrad_plan(C.byref(d), C.byref(x), C.byref(tile), C.byref(slices), C.byref(xg)) == 0
        assert (tile.value, slices.value, xg.value) == (4, sl, 1), (cin, cout, hw, tile.value, slices.value)
        assert L.y3_conv2d_wgrad_workspace_bytes(C.byref(d), C.byref(x)) >= tiles * sl * 128 * 576 * 4
        stages = -(-((64 * (hw + 1) + 1) * (hw + 2)) // 64)
        assert sl * -(-stages // sl) - stages < sl, "the slices differ by at most one 64-position stage"
    # the stride-2 long-K layers keep the 256 x 256 tiles
    big = {(128, 256, 160): 5, (256, 512, 80): 18, (512, 1024, 40): 72}      # (cin, cout, input map) -> 256x256 tiles
    for (cin, cout, hw), tiles in big.items():
        x = Y3Tensor(4096, 64, hw, hw, cin, cin)
        nbytes = L.y3_conv2d_wgrad_workspace_bytes(C.byref(_desc(_lib.Y3_F16, 3, 2, cin, cout)), C.byref(x))
        blocks = nbytes // (256 * 256 * 4)
        assert nbytes % (256 * 256 * 4) == 0 and blocks % tiles == 0, (cin, cout, hw, nbytes)
        rounds = -(-blocks // 256)
        assert rounds <= 2 and blocks / (256 * rounds) >= 0.98, f"{cin}->{cout}@{hw}: {blocks} blocks"
    for cin, cout, k, hw in [(128, 64, 3, 160), (256, 128, 1, 80), (64, 32, 1, 320), (1024, 512, 1, 20)]:
        x = Y3Tensor(4096, 64, hw, hw, cin, cin)
        nbytes = L.y3_conv2d_wgrad_workspace_bytes(C.byref(_desc(_lib.Y3_F16, k, 1, cin, cout)), C.byref(x))
        tiles = -(-cout // 128) * -(-(k * k * cin) // 128)
        assert nbytes % (128 * 128 * 4 * tiles) == 0, (cin, cout, k, hw)
    # the 3x3 layers with 32 -> 64 / 64 -> 128 channels on the large maps: the strip kernel (csrc/wgrad_strip.h) -- one [9 cin][64] fp32 partial tile per
    # persistent block and 64-filter half, 3 / 2 / 2 / 1 blocks per CU (what the LDS holds)
    for cin, cout, s, hw, blocks in [(32, 64, 1, 320, 768), (32, 64, 2, 640, 512), (64, 128, 1, 160, 256), (64, 128, 2, 320, 128)]:
        x = Y3Tensor(4096, 64, hw, hw, cin, cin)
        d = _desc(_lib.Y3_F16, 3, s, cin, cout)
        assert L.y3_conv2d_wgrad_plan(C.byref(d), C.byref(x), C.byref(tile), C.byref(slices), C.byref(xg)) == 0
        assert tile.value == 3 and xg.value == 0 and abs(slices.value - blocks) <= blocks // 50, (cin, cout, s, slices.value)
        assert L.y3_conv2d_wgrad_workspace_bytes(C.byref(d), C.byref(x)) >= slices.value * (cout // 64) * 9 * cin * 64 * 4
    # ... a small launch of the same layer stays on the tile kernel (too few K-steps per block to amortise the partial tiles)
    x = Y3Tensor(4096, 2, 64, 64, 32, 32)
    assert L.y3_conv2d_wgrad_plan(C.byref(_desc(_lib.Y3_F16, 3, 1, 32, 64)), C.byref(x), C.byref(tile), C.byref(slices), C.byref(xg)) == 0 and tile.value == 128


def test_map_parity_helpers_on_cpu():
    """tests/map_parity.py (the mAP-parity experiment of the GPU suite): the scene generator is seeded and emits the reference's label
    format with boxes that match the painted rectangles; the oracle evaluation runs end to end (random weights: no assertion on the value)
    and the metrics glue returns the four numbers."""
    import sys

    sys.path.insert(0, str(ROOT / "tests"))
    import map_parity as mp

    x1, l1 = mp.make_scenes(6, 96, 3, seed=9)
    x2, l2 = mp.make_scenes(6, 96, 3, seed=9)
    assert torch.equal(x1, x2) and torch.equal(l1, l2)
    assert x1.shape == (6, 3, 96, 96) and float(x1.min()) >= 0.0 and float(x1.max()) <= 1.0
    assert l1.shape[1] == 6 and set(l1[:, 0].long().tolist()) == set(range(6)) and int(l1[:, 1].max()) < 3
    assert float(l1[:, 2:].min()) > 0.0 and float((l1[:, 2] + l1[:, 4] / 2).max()) <= 1.0 + 1e-6 and float((l1[:, 3] + l1[:, 5] / 2).max()) <= 1.0 + 1e-6
    # the LAST rectangle painted into an image is fully visible: its interior carries its class colour
    i, c, xc, yc, w, h = l1[l1[:, 0] == 0][-1].tolist()
    px = x1[0, :, int(yc * 96), int(xc * 96)]
    base = torch.tensor(mp.COLOURS[int(c)])
    assert int(px.argmax()) == int(base.argmax()) or float((px / px.max() - base / base.max()).abs().max()) < 0.35
    tg = mp.batch_targets(l1, torch.tensor([4, 1]))
    assert set(tg[:, 0].long().tolist()) == {0, 1} and tg.shape[0] == int((l1[:, 0] == 4).sum() + (l1[:, 0] == 1).sum())
    lab = mp.labels_native(l1, 2, 96)
    assert lab.shape[1] == 5 and bool((lab[:, 3] > lab[:, 1]).all()) and bool((lab[:, 4] > lab[:, 2]).all())
    import yaml

    from oracle import yolo_oracle as yo

    d = yaml.safe_load(open(ROOT / "yolov3_amd" / "cfg" / "yolov3-tiny.yaml"))
    layers, _, anchors, nc = yo.parse_cfg(d, 3, 3)
    sd = yo.seeded_state_dict(layers, nc, anchors, yo.model_strides(layers), seed=1)
    (p, r, m50, m), n_det = mp.evaluate_oracle("yolov3-tiny", 3, sd, x1[:4], l1, 96, bs=4)
    assert all(0.0 <= v <= 1.0 for v in (p, r, m50, m)) and n_det >= 0


def test_grad_sink_flushes_the_bucket_before_the_tail_of_the_backward():
    """The last collective of a backward is the exposed one: once only `_GradSink.TAIL_BYTES` of parameters are still to come the sink asks
    the bucket object to send what it holds (otherwise ~47 MB of yolov3's gradients would wait for `finish()` behind a bucket boundary)."""
    import torch

    from yolov3_amd.train_engine import _GradSink

    class FakeSync:
        def __init__(self):
            self.events = []

        def add(self, k, g):
            self.events.append(("add", k))

        def flush(self):
            self.events.append(("flush",))

        def finish(self):
            return {}

    tail = _GradSink.TAIL_BYTES
    sizes = [tail // 4, tail // 4, tail // 4, tail // 8, tail // 8, tail // 8]          # elements of fp32: 1, 1, 1, .5, .5, .5 tails -> 4.5 tails in all
    fs = FakeSync()
    sink = _GradSink(fs, sum(sizes) * 4)
    for i, n in enumerate(sizes):
        sink[i] = torch.empty(n)
    kinds = [e[0] for e in fs.events]
    assert kinds.count("flush") == 1
    # 3.5 tails seen after the 4th gradient = total - 1 tail: flushed right there, the remaining two gradients fill the last (small) bucket
    assert kinds == ["add", "add", "add", "add", "flush", "add", "add"], kinds
    assert _GradSink(None).result() == {}


def test_conv_v10_tiles_cover_the_pixel_axis_and_neighbours_share_a_round():
    """y3_conv_v10_tiles enumerates the tiles of csrc/conv_v10.h with the functions the kernel itself runs (v10_share / v10_tile_cols).  For the benchmark's shapes,
    odd batch sizes (ragged shares: two share lengths, two tile counts), forced block counts / body widths and the K-split form: the tiles of a filter tile cover the
    column blocks of the pixel axis exactly once, no tile is wider than the body allows, and -- knob v10_group -- the blocks of a group (the blocks xcd_remap puts on one
    XCD) hold NEIGHBOURING tiles in every round, so the halo rows two tiles share are requested in one L2 at the same time; v10_group = 0 gives every block a contiguous
    run of its own (the order before the knob)."""
    import ctypes as C

    from yolov3_amd import _lib
    from yolov3_amd._lib import Y3Tensor

    L = _lib.lib()
    ws = L.y3_conv_workspace_bytes()

    def tiles(bs, cin, cout, hw, workspace=0):
        x, y, d = Y3Tensor(4096, bs, hw, hw, cin, cin), Y3Tensor(8192, bs, hw, hw, cout, cout), _desc(_lib.Y3_F16, 3, 1, cin, cout)
        n, cb, g = C.c_int64(0), C.c_int32(0), C.c_int32(0)
        assert L.y3_conv_v10_tiles(C.byref(d), C.byref(x), C.byref(y), workspace, None, 0, C.byref(n), C.byref(cb), C.byref(g)) == 0, L.y3_last_error()
        rec = (C.c_int32 * (4 * n.value))()
        assert L.y3_conv_v10_tiles(C.byref(d), C.byref(x), C.byref(y), workspace, rec, n.value, C.byref(n), C.byref(cb), C.byref(g)) == 0
        name = C.create_string_buffer(64)
        assert L.y3_conv2d_fwd_variant(C.byref(d), C.byref(x), C.byref(y), 0, workspace, name, 64) == 0
        return [tuple(rec[4 * i:4 * i + 4]) for i in range(n.value)], cb.value, g.value, name.value.decode()

    def check(recs, cb, g, max_body, grouped):
        assert cb == sum(r[3] for r in recs)
        pos = 0
        for _, _, c0, sz in sorted(recs, key=lambda r: r[2]):   # exact cover, no overlap, no gap
            assert c0 == pos and 1 <= sz <= max_body, (c0, pos, sz)
            pos += sz
        assert pos == cb
        per_block = {}
        for bi, t, c0, sz in recs:
            per_block.setdefault(bi, []).append((t, c0, sz))
        if not grouped:
            for bi, ts in per_block.items():   # a contiguous run per block
                ts.sort()
                assert all(ts[i][1] + ts[i][2] == ts[i + 1][1] for i in range(len(ts) - 1)), bi
            return
        # round t of a group: its blocks, in block order, hold consecutive tiles
        n_blocks = max(per_block) + 1
        for g0 in range(0, n_blocks, g):
            rounds = {}
            for bi in range(g0, min(g0 + g, n_blocks)):
                for t, c0, sz in per_block[bi]:
                    rounds.setdefault(t, []).append((bi, c0, sz))
            end_prev = None
            for t in sorted(rounds):
                r = sorted(rounds[t])
                assert all(r[i][1] + r[i][2] == r[i + 1][1] for i in range(len(r) - 1)), (g0, t)   # neighbours within the round
                if end_prev is not None:
                    assert r[0][1] == end_prev, (g0, t)                                            # the next round continues where this one ended
                end_prev = r[-1][1] + r[-1][2]

    try:
        for knob in (1, 0):
            assert L.y3_tune_set(b"v10_group", knob) == 0
            for bs, cin, cout, hw in [(32, 128, 256, 80), (32, 256, 512, 40), (32, 512, 1024, 20), (64, 128, 256, 80), (64, 512, 1024, 20), (37, 128, 256, 80), (29, 256, 512, 40),
                                      (51, 512, 1024, 20), (8, 128, 256, 160), (13, 256, 256, 52)]:
                recs, cb, g, name = tiles(bs, cin, cout, hw)
                assert name in ("v10", "v10h") and cb == (bs * hw * hw + 31) // 32, (name, cb)
                n_ct, blocks = cout // 256, max(r[0] for r in recs) + 1   # filter tiles, blocks per filter tile (two per CU in the half form, fewer on short pixel axes)
                assert blocks <= 256 * (2 if name == "v10h" else 1) // n_ct
                assert g == (max(1, min(n_ct * blocks // 8, blocks)) if knob else 1), (g, name, n_ct, blocks)   # an eighth of the grid = what xcd_remap puts on one XCD
                check(recs, cb, g, 4 if name == "v10h" else 8, bool(knob))
            # forced geometries of the GPU tests: few blocks walking many tiles, single-column-block bodies, the K-split form of a small launch
            for blocks, mp in ((3, 0), (7, 6), (50, 0)):
                assert L.y3_tune_set(b"v10_blocks", blocks) == 0 and L.y3_tune_set(b"v10_mp", mp) == 0 and L.y3_tune_set(b"conv_v10", 2) == 0
                recs, cb, g, name = tiles(3, 128, 256, 38)
                assert len({r[0] for r in recs}) == min(blocks, cb)
                check(recs, cb, g, 8, bool(knob))
                assert L.y3_tune_set(b"v10_blocks", 0) == 0 and L.y3_tune_set(b"v10_mp", 0) == 0 and L.y3_tune_set(b"conv_v10", 1) == 0
            recs, cb, g, name = tiles(2, 256, 512, 40, workspace=ws)
            assert name == "v10k"
            check(recs, cb, g, 8, bool(knob))
            # random problems, every eligible shape forced onto the kernel (small launches included: one block per few column blocks, single-tile shares)
            import random
            rnd = random.Random(7 + knob)
            assert L.y3_tune_set(b"conv_v10", 2) == 0
            for _ in range(150):
                bs, hw = rnd.randint(1, 70), rnd.choice([13, 20, 26, 40, 52, 76, 80, 104])
                cin, cout = rnd.choice([(128, 256), (256, 512), (512, 1024), (256, 256), (128, 512), (384, 768)])
                recs, cb, g, name = tiles(bs, cin, cout, hw)
                assert name in ("v10", "v10h") and cb == (bs * hw * hw + 31) // 32
                check(recs, cb, g, 4 if name == "v10h" else 8, bool(knob))
            assert L.y3_tune_set(b"conv_v10", 1) == 0
    finally:
        L.y3_tune_reset()


def test_bench_launches_its_own_ranks_when_asked_for_several_gpus(tmp_path):
    """`python bench.py --gpus N` with no torch.distributed.run around it starts N ranks itself (127.0.0.1 rendezvous, one rank per GPU) -- the driver's scaling run
    may use either spelling.  Without a GPU every rank stops at "bench.py needs an MI355X" AFTER the process group formed: the launcher path was taken, the group had
    world size 2, and nothing reached the hot path on a CPU."""
    sys.path.insert(0, str(ROOT))
    import bench

    assert not bench.self_launch_needed(1, {})
    assert bench.self_launch_needed(2, {"PATH": "x"})
    assert not bench.self_launch_needed(2, {"WORLD_SIZE": "2", "RANK": "0"})   # under torch.distributed.run: the ranks do not launch again
    cmd = bench.self_launch_command(4, ["--gpus", "4", "--steps", "3"], 29555)
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    if torch.cuda.is_available():
        pytest.skip("the GPU form of this run is tools/gpu_dist_smoke.sh")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--no-train", "--no-cpu-baseline"], capture_output=True, text=True,
                       timeout=300, env=env, cwd=str(tmp_path))
    assert p.returncode != 0
    assert p.stdout.strip() == ""                                   # no JSON line from a run that measured nothing
    assert p.stderr.count("bench.py needs an MI355X") >= 2, p.stderr[-2000:]   # both ranks were started and each got as far as the device check
    assert "AssertionError" not in p.stderr
