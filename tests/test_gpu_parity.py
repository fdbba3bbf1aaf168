"""-m gpu parity tests: the HIP path (through the C ABI) against the CPU oracle and the golden vectors.

Tolerances (stated per test):
  * integer / index work (NMS keep sets, candidate order): bit-exact;
  * fp32 engine (direct kernels): 1e-4 absolute on logits/boxes (BASELINE.json north_star);
  * fp16 / bf16 MFMA engine vs the fp32 oracle fed the SAME rounded inputs/weights: per-layer error is one
    output rounding (2^-11 / 2^-8 relative) plus fp32 accumulation-order noise.
"""
import math
from pathlib import Path

import pytest
import torch
import torch.nn.functional as F
import yaml

from oracle import yolo_oracle as yo

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parents[1]
CFG = ROOT / "yolov3_amd" / "cfg"


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda:0")


def _ops():
    from yolov3_amd import _lib, ops

    return _lib, ops


def checksum(t):
    return float(t.double().abs().sum())


# ------------------------------------------------------------------------------------------------ conv
_WS = {}


def conv_ws(dev):
    """one stream-K workspace for the whole test session (zero-filled once, re-armed by every launch)"""
    _lib, ops = _ops()
    if dev not in _WS:
        _WS[dev] = ops.conv_workspace(dev)
    return _WS[dev]


def run_conv(dev, dtype, n, h, w, cin, cout, k, s, act=True, residual=False, ups=False, sliced=False, algo=0, seed=0, cin_real=None, cout_real=None, ws=False, expect=None,
             repeat=1, check_ws=True):
    """Returns (hip output NCHW fp32 cpu, reference NCHW fp32 cpu computed from the SAME rounded operands).  ws: call through
    y3_conv2d_fwd_ws; expect: assert the kernel variant the dispatcher picks (so a tolerance is tied to the kernel that ran)."""
    _lib, ops = _ops()
    g = torch.Generator().manual_seed(seed)
    cin_real = cin_real or cin
    cout_real = cout_real or cout
    x = torch.randn(n, cin_real, h, w, generator=g)
    wt = torch.randn(cout_real, cin_real, k, k, generator=g) / math.sqrt(cin_real * k * k)
    b = torch.randn(cout_real, generator=g) * 0.5
    xq, wq = x.to(dtype).float(), wt.to(dtype).float()
    ho, wo = (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1
    res = torch.randn(n, cout_real, ho, wo, generator=g).to(dtype).float() if residual else None
    ref = F.conv2d(xq, wq, b, stride=s, padding=k // 2)
    if act:
        ref = F.silu(ref)
    if residual:
        ref = ref + res
    if ups:
        ref = F.interpolate(ref, scale_factor=2.0, mode="nearest")

    xv = ops.View.alloc(n, h, w, cin, dtype, dev)
    ops.nchw_to_nhwc(x.to(dev), xv)
    filt = ops.pack_filter(wt.to(dev), cout, cin, dtype)
    bias = torch.zeros(cout, device=dev)
    bias[:cout_real] = b.to(dev)
    up = 2 if ups else 1
    if sliced:  # write into the middle of a wider buffer (zero-copy concat)
        big = ops.View.alloc(n, ho * up, wo * up, cout + 24, dtype, dev)
        big.buf.fill_(7.0)
        yv = big.slice(16, cout)
    else:
        yv = ops.View.alloc(n, ho * up, wo * up, cout, dtype, dev)
    rv = None
    if residual:
        rv = ops.View.alloc(n, ho, wo, cout, dtype, dev)
        rv.buf.zero_()
        ops.nchw_to_nhwc(res.to(dev), rv)
    wsp = conv_ws(dev) if ws else None
    if expect is not None:
        got = ops.conv_variant(xv, yv, k, s, residual, ups, algo, workspace_bytes=wsp.numel() if ws else 0)
        assert got == expect, f"dispatcher picked {got}, the test is written for {expect}"
    first = None
    for r in range(repeat):   # same workspace, back-to-back launches: flags / ticket must re-arm, results must be bit-identical
        if r:
            yv.as_nhwc().fill_(-3.0)
        ops.conv2d(xv, filt, bias, yv, k, s, act, rv, ups, algo, workspace=wsp)
        torch.cuda.synchronize()
        cur = yv.as_nhwc().clone()
        if first is None:
            first = cur
        else:
            assert torch.equal(first, cur), f"launch {r} differs from launch 0 (non-deterministic or stale workspace state)"
    if ws and check_ws:   # the K-split form only uses the slabs behind the first 8 KiB of the workspace: the head stays as the caller zero-filled it
        assert int(wsp[:8192].view(torch.int32).abs().sum()) == 0, "the conv wrote into the head of its workspace"
    out = yv.as_nhwc().float().cpu().permute(0, 3, 1, 2)[:, :cout_real]
    if sliced:
        full = big.as_nhwc().float().cpu()
        assert torch.all(full[..., :16] == 7.0) and torch.all(full[..., 16 + cout :] == 7.0), "conv wrote outside its channel slice"
    return out, ref


CONV_CASES = [
    # name, (n,h,w,cin,cout,k,s), kwargs
    ("3x3s1_bk64_tc128", (2, 20, 20, 64, 128, 3, 1), {}),
    ("3x3s2_bk64_tc128", (2, 23, 19, 128, 256, 3, 2), {}),
    ("3x3s1_bk32_tc64", (1, 17, 33, 32, 64, 3, 1), {}),
    ("3x3s2_bk32_tc64", (2, 32, 32, 32, 64, 3, 2), {}),
    ("first_layer_cin3", (2, 40, 36, 8, 32, 3, 1), {"cin_real": 3}),
    ("tiny_cin16", (2, 26, 26, 16, 32, 3, 1), {}),
    ("tiny_first_cout16", (1, 32, 32, 8, 16, 3, 1), {"cin_real": 3}),
    ("1x1_cout32", (2, 40, 40, 64, 32, 1, 1), {}),
    ("1x1_cout64", (2, 20, 20, 128, 64, 1, 1), {}),
    ("1x1_k1024", (2, 10, 10, 1024, 512, 1, 1), {}),
    ("1x1_k768_concat_in", (1, 20, 20, 768, 256, 1, 1), {}),
    ("head_255", (2, 20, 20, 256, 256, 1, 1), {"cout_real": 255, "act": False}),
    ("residual", (2, 20, 20, 64, 128, 3, 1), {"residual": True}),
    ("upsample_scatter", (2, 10, 10, 512, 256, 1, 1), {"ups": True}),
    ("sliced_output", (2, 20, 20, 64, 128, 3, 1), {"sliced": True}),
    ("big_k_3x3_512", (1, 20, 20, 512, 1024, 3, 1), {}),
    ("partial_tiles", (1, 7, 5, 64, 136, 3, 1), {}),
]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("name,shape,kw", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_mfma_vs_fp32_reference(dev, dtype, name, shape, kw):
    out, ref = run_conv(dev, dtype, *shape, algo=1, **kw)
    eps = 2.0**-10 if dtype == torch.float16 else 2.0**-7
    err = (out - ref).abs()
    tol = eps * ref.abs() + 2e-3 if dtype == torch.float16 else eps * ref.abs() + 1.5e-2
    bad = (err > tol).sum().item()
    assert bad == 0, f"{name} {dtype}: {bad}/{err.numel()} outside tolerance, max abs err {err.max():.4g}, ref max {ref.abs().max():.3g}"


def _conv_tol_check(name, dtype, out, ref):
    eps = 2.0**-10 if dtype == torch.float16 else 2.0**-7
    err = (out - ref).abs()
    tol = eps * ref.abs() + (2e-3 if dtype == torch.float16 else 1.5e-2)
    bad = (err > tol).sum().item()
    assert bad == 0, f"{name} {dtype}: {bad}/{err.numel()} outside tolerance, max abs err {err.max():.4g}, ref max {ref.abs().max():.3g}"


# The K-split form of the persistent 3x3 kernel (conv_v10.h SPLIT + conv_v10_reduce_kernel: small launches) at sizes chosen to hit its edge logic: ragged last
# tile, image borders inside a tile, several images per tile, wide maps (the half-size geometry does not fit: full-size bodies), one channel block per slice, uneven
# slices, residual / no activation / channel-slice output.  (Round 2's stream-K kernel conv_v7.h served these launches until round 4.)
KSPLIT_CASES = [
    # name, (n,h,w,cin,cout,k,s), kwargs, slices (knob v10_slices; 0 = the host's plan), expected variant
    ("w20_two_ct", (6, 20, 20, 64, 512, 3, 1), {}, 0, "v10k"),
    ("w13_ragged_3_slices_of_3", (40, 11, 13, 96, 256, 3, 1), {"residual": True}, 3, "v10k"),
    ("w40_noact", (2, 40, 40, 128, 256, 3, 1), {"act": False}, 0, "v10k"),
    ("w80_two_requests", (1, 33, 80, 128, 256, 3, 1), {"residual": True}, 3, "v10k"),          # 4 channel blocks in 3 slices: 2 + 1 + 1
    ("w160_full_size_bodies", (1, 30, 160, 64, 256, 3, 1), {}, 2, "v10k"),                      # the 31 KiB patch buffer of the half form does not hold 160-pixel rows
    ("w190_too_wide", (1, 24, 190, 64, 256, 3, 1), {}, 0, None),                               # no form fits: the tile kernels take it
    ("deep_k_16_slices", (2, 20, 20, 512, 1024, 3, 1), {"residual": True}, 16, "v10k"),         # one channel block per slice
    ("sliced_out", (8, 24, 20, 128, 256, 3, 1), {"sliced": True}, 0, "v10k"),
    ("h1_rows", (40, 1, 70, 96, 256, 3, 1), {}, 2, "v10k"),
    ("w1_cols", (30, 90, 1, 96, 256, 3, 1), {}, 0, "v10k"),
]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("name,shape,kw,slices,variant", KSPLIT_CASES, ids=[c[0] for c in KSPLIT_CASES])
def test_conv_v10_ksplit_vs_fp32_reference(dev, dtype, name, shape, kw, slices, variant, tune):
    tune("conv_v10", 2)      # also the Cin < 128 shapes the dispatcher leaves to the small tiles
    tune("v10_ksplit", 2)    # the K-split form whatever the tile count
    tune("v10_slices", slices)
    out, ref = run_conv(dev, dtype, *shape, algo=1, ws=True, expect=variant, repeat=3, **kw)
    _conv_tol_check(name, dtype, out, ref)


# conv_1x1s.h ("s1x1"): persistent 1x1 kernel, filters in registers, a producer wave streaming the pixels through three fragment-ordered LDS stages.  Every
# instantiation; pixel counts that end inside a stage, inside an epilogue pass and inside a column block; fewer stages than CUs and more; residual, no activation,
# channel-slice output.  (knob conv_1x1s = 2: also below the pixel count where the dispatcher would pick it)
S1X1_CASES = [
    # name, (n,h,w,cin,cout,k,s), kwargs
    ("k16_256_128_many_stages", (9, 80, 80, 256, 128, 1, 1), {}),                        # 900 stages of 64 pixels on 256 blocks: 3-4 per block
    ("k16_ragged_res", (3, 37, 29, 256, 128, 1, 1), {"residual": True}),                # 3219 pixels: the last stage holds 19
    ("k24_384_128_sliced_noact", (2, 40, 40, 384, 128, 1, 1), {"sliced": True, "act": False}),
    ("k8_128_64_two_pixel_waves", (5, 33, 31, 128, 64, 1, 1), {"residual": True}),      # stages of 128 pixels, two waves along the pixels
    ("k8_128_256_two_filter_groups", (4, 24, 24, 128, 256, 1, 1), {}),                  # a consumer wave multiplies the stage once per filter group
    ("k4_64_128_four_passes", (2, 50, 50, 64, 128, 1, 1), {"act": False}),              # stages of 256 pixels: four epilogue passes per stage
    ("k16_one_pixel", (1, 1, 1, 256, 128, 1, 1), {}),
    ("k2_32_64_stages_of_512", (3, 40, 44, 32, 64, 1, 1), {"residual": True}),         # 5280 pixels: 10 stages and a third of one
    ("k4_64_32_four_pixel_waves", (2, 36, 36, 64, 32, 1, 1), {}),
    ("k16_256_255_head", (2, 30, 30, 256, 256, 1, 1), {"cout_real": 255, "act": False}),
    ("k8_128_384_three_filter_groups", (2, 30, 30, 128, 384, 1, 1), {"residual": True, "act": False}),
    ("k16_256_512_two_tiles_two_groups", (3, 40, 40, 256, 512, 1, 1), {"residual": True, "act": False}),
    ("k16_256_768_three_tiles", (2, 20, 20, 256, 768, 1, 1), {"act": False}),
]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("name,shape,kw", S1X1_CASES, ids=[c[0] for c in S1X1_CASES])
def test_conv_s1x1_vs_fp32_reference(dev, tune, dtype, name, shape, kw):
    tune("conv_1x1s", 2)
    out, ref = run_conv(dev, dtype, *shape, algo=1, expect="s1x1", repeat=2, **kw)
    _conv_tol_check(name, dtype, out, ref)
    # the same launch on the tile kernels: same products, another summation order
    tune("conv_1x1s", 0)
    out0, _ = run_conv(dev, dtype, *shape, algo=1, **kw)
    assert (out - out0).abs().max().item() <= (2.0 ** -7 if dtype == torch.float16 else 2.0 ** -4) * max(1.0, ref.abs().max().item())


BNIN_CASES = [
    # name, (n, h, w, cin, cout), shortcut, SiLU
    ("256_128_shortcut_ragged", (3, 37, 29, 256, 128), True, True),      # stages of 32 pixels (u + shortcut rows): 3219 pixels end inside a stage
    ("256_128_plain", (2, 40, 40, 256, 128), False, True),               # stages of 64 pixels
    ("256_128_many_stages", (20, 40, 40, 256, 128), True, True),         # 1000 stages on 256 blocks
    ("128_64_shortcut", (5, 33, 31, 128, 64), True, True),               # two pixel waves, one column block each
    ("128_64_plain_noact", (2, 50, 50, 128, 64), False, False),
    ("one_pixel", (1, 1, 1, 256, 128), True, True),
]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("name,shape,shortcut,silu", BNIN_CASES, ids=[c[0] for c in BNIN_CASES])
def test_conv1x1_bn_in_consumer_matches_separate_passes(dev, tune, dtype, name, shape, shortcut, silu):
    """y3_conv2d_fwd_bnin_stats (conv_1x1s.h, IN form): the producing layer's act(scale u + shift) (+ shortcut) applied on the way into its 1x1 consumer.  Against the two launches
    it replaces -- y3_bn_act_fwd, then y3_conv2d_fwd_stats on its output: the normalised tensor it stores, the convolution output and the statistics rows' sums are the SAME
    bits (same arithmetic on the same operands, same K order), every pixel and channel.  (Knob conv_1x1s = 2: the dispatcher leaves launches below 8192 pixels to the
    separate passes since round 6 -- most cases here are smaller.)"""
    _lib, ops = _ops()
    import ctypes as C

    tune("conv_1x1s", 2)
    n, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(7)
    M = n * h * w
    u_in = ops.View.alloc(n, h, w, cin, dtype, dev)
    u_in.buf.copy_((torch.randn(M * cin, generator=g) * 1.5).to(dtype))
    res = None
    if shortcut:
        res = ops.View.alloc(n, h, w, cin, dtype, dev)
        res.buf.copy_(torch.randn(M * cin, generator=g).to(dtype))
    scale = (torch.rand(cin, generator=g) + 0.5).to(dev)
    shift = torch.randn(cin, generator=g).to(dev) * 0.3
    wt = torch.randn(cout, cin, 1, 1, generator=g) / math.sqrt(cin)
    filt = ops.pack_filter(wt.to(dev), cout, cin, dtype)
    zb = torch.zeros(cout, device=dev)
    act = _lib.Y3_ACT_SILU if silu else _lib.Y3_ACT_NONE
    # the two launches
    y_ref = ops.View.alloc(n, h, w, cin, dtype, dev)
    ut, yt = u_in.y3(), y_ref.y3()
    rt = res.y3() if res is not None else None
    _lib.check(_lib.lib().y3_bn_act_fwd(C.byref(ut), scale.data_ptr(), shift.data_ptr(), C.byref(rt) if rt is not None else None, C.byref(yt), ops.dtype_code(dtype), act,
                                        ops.stream_ptr()), "y3_bn_act_fwd")
    o_ref = ops.View.alloc(n, h, w, cout, dtype, dev)
    rows_ref = ops.conv2d_stats_rows(y_ref, o_ref, 1, 1)
    buf_ref = torch.full((rows_ref * 2 * cout,), float("nan"), device=dev)
    ops.conv2d_stats(y_ref, filt, zb, o_ref, 1, 1, buf_ref, rows_ref)
    # the one launch
    y_f = ops.View.alloc(n, h, w, cin, dtype, dev)
    y_f.buf.fill_(-7.0)
    o_f = ops.View.alloc(n, h, w, cout, dtype, dev)
    o_f.buf.fill_(-7.0)
    rows = ops.conv1x1_bnin_rows(u_in, y_f, o_f, shortcut)
    assert rows > 0
    buf = torch.full((rows * 2 * cout,), float("nan"), device=dev)
    got = ops.conv1x1_bnin_stats(u_in, scale, shift, act, res, y_f, filt, zb, o_f, buf, rows)
    torch.cuda.synchronize()
    assert got == rows and ops.last_conv_variant() == "s1x1_bn"
    assert torch.equal(y_f.buf, y_ref.buf), f"normalised tensor differs: max {(y_f.buf.float() - y_ref.buf.float()).abs().max().item()}"
    assert torch.equal(o_f.buf, o_ref.buf), f"conv output differs: max {(o_f.buf.float() - o_ref.buf.float()).abs().max().item()}"
    tot, tot_ref = buf.view(rows, cout, 2).double().sum(0), buf_ref.view(rows_ref, cout, 2).double().sum(0)
    assert torch.isfinite(tot).all(), "a statistics row was not written"
    assert (tot - tot_ref).abs().max().item() <= 1e-6 * tot_ref.abs().max().item()


@pytest.mark.parametrize("shape", [(3, 40, 40, 256, 512, 3, 2), (4, 40, 40, 512, 256, 1, 1)], ids=["v6_3x3_s2", "v6_1x1"])
def test_conv_request_depth_bit_identical(dev, tune, shape):
    """knob "conv_ahead": the LDS-DMA requests of the 256x256 kernel run 3 K-steps ahead of the MFMAs (default, round 3) or 2 (the
    round-2 schedule).  Both schedules add the same products in the same order: bit-identical outputs, launch after launch (a stage
    overwritten while a wave still reads it would show up here as a flip)."""
    outs = []
    tune("conv", 15)   # force the 256x256 v6 tile whatever the per-shape dispatch would pick at this small batch
    for ahead in (3, 2):
        tune("conv_ahead", ahead)
        out, ref = run_conv(dev, torch.float16, *shape, algo=1, ws=True, expect="v6", repeat=3)
        _conv_tol_check(f"v6 ahead{ahead}", torch.float16, out, ref)
        outs.append(out)
    assert torch.equal(outs[-1], outs[-2]), "request depth 3 differs from depth 2"


def test_conv_v10_ksplit_slice_sweep(dev, tune):
    """the same small problem under 1 .. 8 slices of its 8 channel blocks (the default dispatch picks the K-split form here: below a quarter round of tiles, with a
    workspace): every split sums the same products in fp32, so the results agree to accumulation-order noise, each one is inside the conv tolerance, repeated
    launches are bit-identical, and one slice reproduces the unsplit kernel's sums exactly"""
    outs = {}
    for sl in (0, 1, 2, 3, 5, 8):
        tune("v10_slices", sl)
        out, ref = run_conv(dev, torch.float16, 4, 20, 20, 256, 512, 3, 1, algo=1, ws=True, expect="v10k", repeat=2, residual=True)
        _conv_tol_check(f"slices {sl}", torch.float16, out, ref)
        outs[sl] = out
    for sl, o in outs.items():
        assert (o - outs[0]).abs().max().item() <= 2.0**-9 * max(1.0, outs[0].abs().max().item()), sl
    tune("v10_slices", 0)
    tune("v10_ksplit", 0)
    tune("conv_v10", 2)
    whole, _ = run_conv(dev, torch.float16, 4, 20, 20, 256, 512, 3, 1, algo=1, ws=True, expect="v10h", residual=True)
    assert (outs[1] - whole).abs().max().item() <= 2.0**-10 * max(1.0, whole.abs().max().item())   # (same fp32 sums; the slab sum rounds once like the epilogue)


@pytest.mark.parametrize("slices", [0, 3])
def test_conv_v10_ksplit_statistics_rows(dev, tune, slices):
    """BatchNorm statistics rows of the K-split form come from the slab sum (one row per 64-pixel block, ragged last block): their fp64 sum equals the statistics
    of the stored tensor"""
    _lib, ops = _ops()
    tune("v10_slices", slices)
    n, h, w, cin, cout, k, s = 3, 20, 19, 128, 256, 3, 1
    dtype = torch.float16
    g = torch.Generator().manual_seed(4)
    x = torch.randn(n, cin, h, w, generator=g).to(dtype)
    wt = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
    xv = ops.View.alloc(n, h, w, cin, dtype, dev)
    ops.nchw_to_nhwc(x.to(dev), xv)
    filt = ops.pack_filter(wt.to(dev), cout, cin, dtype)
    zb = torch.zeros(cout, device=dev)
    y1 = ops.View.alloc(n, h, w, cout, dtype, dev)
    ws = conv_ws(dev)
    rows = ops.conv2d_stats_rows(xv, y1, k, s, workspace=ws)
    assert rows == -(-n * h * w // 64), rows
    buf = torch.full((rows * 2 * cout,), float("nan"), device=dev)
    assert ops.conv2d_stats(xv, filt, zb, y1, k, s, buf, rows, workspace=ws) == rows and ops.last_conv_variant() == "v10k"
    torch.cuda.synchronize()
    u = y1.as_nhwc().double().cpu().reshape(-1, cout)
    tot = buf.view(rows, cout, 2).double().sum(0).cpu()
    assert torch.isfinite(tot).all(), "a statistics row was not written"
    assert (tot[:, 0] - u.sum(0)).abs().max().item() <= 1e-5 * u.abs().sum(0).max().item()
    assert (tot[:, 1] - (u * u).sum(0)).abs().max().item() <= 1e-5 * (u * u).sum(0).max().item()
    ref = F.conv2d(x.float(), wt.to(dtype).float(), None, padding=1)
    assert (y1.as_nhwc().float().cpu().permute(0, 3, 1, 2) - ref).abs().max().item() < 2e-2


# BASELINE.json configs[1] / configs[3] / configs[4] layer shapes at their benchmarked batch: the branches of dispatch_igemm that the
# bench actually runs (multi-round grids, XCD remap at 200-3200 blocks, ragged last tiles at M = 12800 / 51200 / 204800).
BASELINE_CONV_CASES = [
    # name, (n,h,w,cin,cout,k,s), kwargs, variant with workspace
    ("L6cv2_128_256_80", (32, 80, 80, 128, 256, 3, 1), {"residual": True}, "v10h"),
    ("L8cv2_256_512_40", (32, 40, 40, 256, 512, 3, 1), {"residual": True}, "v10h"),
    ("L10cv2_512_1024_20", (32, 20, 20, 512, 1024, 3, 1), {"residual": True}, "v10"),
    ("L7_256_512_s2", (32, 80, 80, 256, 512, 3, 2), {}, "v6"),
    ("L9_512_1024_s2", (32, 40, 40, 512, 1024, 3, 2), {}, "v6"),
    ("L8cv1_512_256_40", (32, 40, 40, 512, 256, 1, 1), {}, "v6"),
    ("L10cv1_1024_512_20", (32, 20, 20, 1024, 512, 1, 1), {}, "v3_bk64_128x128"),
    ("L6cv1_256_128_80", (32, 80, 80, 256, 128, 1, 1), {}, "s1x1"),   # conv_1x1s.h (round 5): the HBM-bound 1x1 layers
    ("L4cv2_64_128_160", (32, 160, 160, 64, 128, 3, 1), {"residual": True}, "v3_bk32_128x256"),
    ("L3_64_128_s2_320", (32, 320, 320, 64, 128, 3, 2), {}, "strip"),   # conv_strip.h, stride-2 form (round 3)
    ("head255_20", (32, 20, 20, 1024, 256, 1, 1), {"cout_real": 255, "act": False}, "v3_bk64_128x128"),
    ("head255_80", (32, 80, 80, 256, 256, 1, 1), {"cout_real": 255, "act": False}, "s1x1"),
    ("L26cv1_384_128_80", (32, 80, 80, 384, 128, 1, 1), {}, "s1x1"),
    ("L4cv1_128_64_160", (32, 160, 160, 128, 64, 1, 1), {}, "s1x1"),
    ("head1110_40_c5", (8, 40, 40, 1024, 1112, 1, 1), {"cout_real": 1110, "act": False}, None),
    ("c5_128_256_160", (8, 160, 160, 128, 256, 3, 1), {"residual": True}, "v10"),
    ("c5_256_512_80", (8, 80, 80, 256, 512, 3, 1), {"residual": True}, "v10h"),
    ("ups_route_20", (32, 20, 20, 512, 256, 1, 1), {"ups": True}, None),
]


@pytest.mark.parametrize("name,shape,kw,variant", BASELINE_CONV_CASES, ids=[c[0] for c in BASELINE_CONV_CASES])
def test_conv_baseline_shapes_fp16(dev, name, shape, kw, variant):
    out, ref = run_conv(dev, torch.float16, *shape, algo=1, ws=True, expect=variant, **kw)
    _conv_tol_check(name, torch.float16, out, ref)


@pytest.mark.parametrize("name,shape,kw,variant", [c for c in BASELINE_CONV_CASES if c[0] in ("L8cv2_256_512_40", "L9_512_1024_s2", "head1110_40_c5", "c5_256_512_80")],
                         ids=["L8cv2_256_512_40", "L9_512_1024_s2", "head1110_40_c5", "c5_256_512_80"])
def test_conv_baseline_shapes_bf16(dev, name, shape, kw, variant):
    out, ref = run_conv(dev, torch.bfloat16, *shape, algo=1, ws=True, expect=variant, **kw)
    _conv_tol_check(name, torch.bfloat16, out, ref)


@pytest.mark.parametrize("name,shape,kw", CONV_CASES[:8], ids=[c[0] for c in CONV_CASES[:8]])
def test_conv_direct_fp32_vs_reference(dev, name, shape, kw):
    out, ref = run_conv(dev, torch.float32, *shape, algo=2, **kw)
    err = (out - ref).abs().max().item()
    assert err < 1e-4, f"{name}: max abs err {err:.3g}"  # fp32 path: north-star tolerance


def test_conv_mfma_matches_direct_kernel(dev):
    """same operands through both HIP kernels: identical up to fp32 summation order"""
    a, _ = run_conv(dev, torch.float16, 2, 20, 20, 128, 256, 3, 1, algo=1)
    b, _ = run_conv(dev, torch.float16, 2, 20, 20, 128, 256, 3, 1, algo=2)
    assert (a - b).abs().max().item() <= 2.0**-9 * max(1.0, b.abs().max().item())


# ------------------------------------------------------------------------------------------------ pooling / layout
STEM_CASES = [
    # name, (n, cin, h, w, cout), source dtype, divisor, act
    ("yolov3_l0", (2, 3, 40, 72, 32), torch.float16, 1.0, True),
    ("ragged_hw", (1, 3, 37, 131, 32), torch.float16, 1.0, True),        # H % 8 != 0, W % 64 != 0
    ("u8_div255", (2, 3, 24, 64, 32), torch.uint8, 255.0, True),         # val.py:354-360 ingest fused in
    ("f32_source", (1, 3, 16, 70, 32), torch.float32, 1.0, True),
    ("tiny_cout16", (2, 3, 32, 32, 16), torch.float16, 1.0, True),       # models/yolov3-tiny.yaml first layer
    ("cout64_noact", (1, 3, 19, 65, 64), torch.float16, 1.0, False),
    ("cin1", (1, 1, 16, 64, 32), torch.float16, 1.0, True),
    ("cin4", (1, 4, 9, 33, 24), torch.float16, 1.0, True),
]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("name,shape,sdt,div,act", STEM_CASES, ids=[c[0] for c in STEM_CASES])
def test_stem_conv_vs_fp32_reference(dev, dtype, name, shape, sdt, div, act):
    """csrc/stem.hip (layer 0 straight from the NCHW image) against conv2d(+SiLU) on the SAME rounded operands; tolerance =
    one output rounding + fp32 accumulation-order noise, as for the generic conv kernels."""
    _lib, ops = _ops()
    n, cin, h, w, cout = shape
    g = torch.Generator().manual_seed(11)
    if sdt == torch.uint8:
        x = torch.randint(0, 256, (n, cin, h, w), generator=g, dtype=torch.uint8)
    else:
        x = torch.rand(n, cin, h, w, generator=g).to(sdt)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    b = torch.randn(cout, generator=g) * 0.5
    xq = (x.float() / div).to(dtype).float()      # the kernel rounds the (divided) image to the compute dtype, like y3_nchw_to_nhwc
    ref = F.conv2d(xq, wt.to(dtype).float(), b, stride=1, padding=1)
    if act:
        ref = F.silu(ref)
    cpad = (cout + 7) // 8 * 8
    big = ops.View.alloc(n, h, w, cpad + 16, dtype, dev)   # write into a channel slice: the kernel must respect the pitch
    big.buf.fill_(3.0)
    yv = big.slice(8, cpad)
    filt = ops.pack_filter_stem(wt.to(dev), cpad, dtype)
    bias = torch.zeros(cpad, device=dev)
    bias[:cout] = b.to(dev)
    ops.stem_conv(x.to(dev), filt, bias, yv, act, div)
    torch.cuda.synchronize()
    full = big.as_nhwc().float().cpu()
    assert torch.all(full[..., :8] == 3.0) and torch.all(full[..., 8 + cpad :] == 3.0), "stem wrote outside its channel slice"
    out = full[..., 8 : 8 + cout].permute(0, 3, 1, 2)
    eps = 2.0**-10 if dtype == torch.float16 else 2.0**-7
    err = (out - ref).abs()
    tol = eps * ref.abs() + (2e-3 if dtype == torch.float16 else 1.5e-2)
    bad = (err > tol).sum().item()
    assert bad == 0, f"{name} {dtype}: {bad}/{err.numel()} outside tolerance, max abs err {err.max():.4g}"


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 3, 40, 72, 32), (1, 3, 37, 131, 32), (3, 3, 24, 200, 64), (2, 3, 32, 32, 16)], ids=["l0", "ragged", "cout64", "cout16"])
def test_stem_conv_statistics_rows(dev, dtype, shape):
    """y3_stem_conv_fwd_stats: the rows are (sum, sum of squares) of the STORED values of each block -- summed over the rows (fp64) they
    equal the per-channel sums of the output tensor the same launch wrote (fp32 partials per block: 1e-5 relative), the output is
    bit-identical to the launch without statistics, ragged tiles contribute only their valid pixels."""
    _lib, ops = _ops()
    n, cin, h, w, cout = shape
    g = torch.Generator().manual_seed(5)
    x = torch.rand(n, cin, h, w, generator=g).to(dtype).to(dev)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    filt = ops.pack_filter_stem(wt.to(dev), cout, dtype)
    bias = torch.zeros(cout, device=dev)
    y0, y1 = ops.View.alloc(n, h, w, cout, dtype, dev), ops.View.alloc(n, h, w, cout, dtype, dev)
    ops.stem_conv(x, filt, bias, y0, False)
    rows = ops.stem_conv_stats_rows(n, h, w)
    assert rows == n * ((h + 7) // 8) * ((w + 63) // 64)
    buf = torch.full((rows + 1, cout, 2), float("nan"), device=dev)
    got_rows = ops.stem_conv_stats(x, filt, bias, y1, buf, rows)
    torch.cuda.synchronize()
    assert got_rows == rows
    assert torch.equal(y0.as_nhwc(), y1.as_nhwc())
    assert torch.isnan(buf[rows]).all(), "wrote past the reported rows"
    r = buf[:rows].double().sum(0).cpu()
    v = y1.as_nhwc().double().reshape(-1, cout).cpu()
    torch.testing.assert_close(r[:, 0], v.sum(0), rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(r[:, 1], (v * v).sum(0), rtol=1e-5, atol=1e-4)
    with pytest.raises(RuntimeError, match="capacity"):
        ops.stem_conv_stats(x, filt, bias, y1, buf, rows - 1)


def test_stem_matches_generic_first_layer(dev, monkeypatch):
    """The same model with and without the stem kernel (Y3_STEM=0 = ingest + generic conv): identical up to fp32 summation order."""
    x = torch.rand(2, 3, 64, 96, generator=torch.Generator().manual_seed(3)).to(dev).half()
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("Y3_STEM", flag)
        m, _ = build_pair("yolov3", 80, 5, dev, torch.float16)
        with torch.no_grad():
            pred = m(x)[0]
        plan = next(iter(m._plans.values()))
        assert (plan.stem_x is not None) == (flag == "1")
        outs.append(pred.float().cpu())
    d = (outs[0] - outs[1]).abs()
    assert d.max().item() <= 2.0**-8 * max(1.0, outs[1].abs().max().item()), d.max().item()


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_layout_and_pool_kernels(dev, dtype):
    _lib, ops = _ops()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 16, 13, 14, generator=g).to(dtype)
    xv = ops.View.alloc(2, 13, 14, 16, dtype, dev)
    ops.nchw_to_nhwc(x.to(dev), xv)
    assert torch.equal(ops.nhwc_to_nchw(xv).cpu(), x)
    # MaxPool2d(2,2,0)
    yv = ops.View.alloc(2, 6, 7, 16, dtype, dev)
    ops.maxpool2d(xv, yv, 2, 2, 0)
    assert torch.equal(ops.nhwc_to_nchw(yv).cpu(), F.max_pool2d(x.float(), 2, 2, 0).to(dtype))
    # ZeroPad2d([0,1,0,1]) + MaxPool2d(2,1,0)  (yolov3-tiny layers 11-12)
    yv = ops.View.alloc(2, 13, 14, 16, dtype, dev)
    ops.maxpool2d(xv, yv, 2, 1, 0, 1, 1)
    assert torch.equal(ops.nhwc_to_nchw(yv).cpu(), F.max_pool2d(F.pad(x.float(), [0, 1, 0, 1]), 2, 1, 0).to(dtype))
    # SPP pyramid into channel slices of a 4C buffer
    cat = ops.View.alloc(2, 13, 14, 64, dtype, dev)
    cat.buf.zero_()
    ops.copy_slice(xv, cat.slice(0, 16))
    ops.spp_pyramid(cat.slice(0, 16), cat.slice(16, 48))
    ref = torch.cat([x.float()] + [F.max_pool2d(x.float(), k, 1, k // 2) for k in (5, 9, 13)], 1).to(dtype)
    assert torch.equal(ops.nhwc_to_nchw(cat).cpu(), ref)
    # nearest x2
    uv = ops.View.alloc(2, 26, 28, 16, dtype, dev)
    ops.upsample2x(xv, uv)
    assert torch.equal(ops.nhwc_to_nchw(uv).cpu(), F.interpolate(x.float(), scale_factor=2.0, mode="nearest").to(dtype))
    # uint8 ingest with /255 in the output dtype (val.py:358-359)
    u8 = torch.randint(0, 256, (2, 3, 9, 11), generator=g, dtype=torch.uint8)
    iv = ops.View.alloc(2, 9, 11, 8, dtype, dev)
    ops.nchw_to_nhwc(u8.to(dev), iv, 255.0)
    got = ops.nhwc_to_nchw(iv).cpu()
    ref = u8.to(dtype) / 255
    assert torch.equal(got[:, :3], ref) and torch.all(got[:, 3:] == 0)


# ------------------------------------------------------------------------------------------------ decode
@pytest.mark.parametrize("key,dtype,nc", [("nc80-float32", torch.float32, 80), ("nc80-float16", torch.float16, 80), ("nc3-float16", torch.float16, 3)])
def test_detect_decode_vs_reference_golden(dev, golden_dir, key, dtype, nc):
    from yolov3_amd import Detect

    gold = torch.load(golden_dir / "decode.pt")[key]
    no = nc + 5
    g = torch.Generator().manual_seed(21)
    xs = [torch.randn(2, 3 * no, s, s + 1, generator=g) * 2.0 for s in gold["sizes"]]
    anchors = [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]
    det = Detect(nc, anchors, ch=(3 * no,) * 3)
    det.stride = torch.tensor([8.0, 16.0, 32.0])
    det.anchors /= det.stride.view(-1, 1, 1)
    for conv in det.m:  # identity head so the decode sees exactly the seeded maps
        conv.weight.data = torch.eye(3 * no).view(3 * no, 3 * no, 1, 1)
        conv.bias.data.zero_()
    det = det.to(dev).to(dtype).eval()
    z, raw = det([x.to(dev).to(dtype) for x in xs])
    torch.cuda.synchronize()
    zc, ref = z.float().cpu(), gold["z"].float()
    assert z.dtype == gold["z"].dtype and zc.shape == ref.shape
    for r, x in zip(raw, xs):
        exp = x.to(dtype).view(2, 3, no, x.shape[2], x.shape[3]).permute(0, 1, 3, 4, 2)
        assert torch.equal(r.cpu(), exp), "raw (bs,na,ny,nx,no) layout mismatch"
    if dtype == torch.float32:
        # sigmoid differs by <= 2 ulp between libm implementations; everything else is exact
        torch.testing.assert_close(zc, ref, rtol=3e-6, atol=1e-6)
    else:
        # every op is rounded through fp16 like torch does; a 1-ulp sigmoid difference may flip a rounding
        ulp = torch.maximum(ref.abs(), torch.tensor(2.0**-14)) * 2.0**-10
        diff = (zc - ref).abs()
        assert (diff > 2 * ulp).sum().item() == 0, f"max diff {diff.max()}"
        assert (diff > 0).float().mean().item() < 2e-3, "too many fp16 rounding flips"


@pytest.mark.parametrize("cin,cout,hw,res", [(128, 256, 40, True), (256, 512, 20, False), (512, 1024, 20, True)])
def test_conv_v10_deferred_epilogue_form_is_bit_identical(dev, tune, cin, cout, hw, res):
    """conv_v10d.h (knob v10_defer, off by default: measured 9-13 % slower, profiles/r06_v10_deferred_epilogue.txt): the previous tile's activation inside the next
    tile's K loop, two accumulator sets, stores behind the K loop.  Same arithmetic per value as the shipped form: the outputs must be the same bits -- tiles of 3 and
    4 column blocks, blocks that walk several tiles, the residual added in the pack step, first / last tile of a block."""
    from yolov3_amd import ops

    g = torch.Generator().manual_seed(cin + hw)
    n, dt = 6, torch.float16
    xv = ops.View.alloc(n, hw, hw, cin, dt, dev)
    xv.buf.copy_(torch.randn(xv.buf.shape, generator=g).to(dt))
    rv = ops.View.alloc(n, hw, hw, cout, dt, dev)
    rv.buf.copy_(torch.randn(rv.buf.shape, generator=g).to(dt))
    f = ops.pack_filter((torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin**0.5)).to(dev), cout, cin, dt)
    b = torch.randn(cout, generator=g).to(dev)
    outs = []
    for defer, blocks in ((0, 0), (2, 0), (2, 3)):
        tune("conv_v10", 2)
        tune("v10_defer", defer)
        tune("v10_blocks", blocks)
        yv = ops.View.alloc(n, hw, hw, cout, dt, dev)
        yv.buf.fill_(7.0)
        ops.conv2d(xv, f, b, yv, 3, 1, True, residual=rv if res else None)
        assert ops.last_conv_variant() == ("v10d" if defer else ("v10h" if cin <= 256 else "v10")), ops.last_conv_variant()
        outs.append(yv.as_nhwc().clone())
    assert torch.isfinite(outs[0].float()).all()
    assert torch.equal(outs[0], outs[1]), "deferred form differs from the shipped form"
    assert torch.equal(outs[0], outs[2]), "deferred form, three blocks per filter tile (many tiles per block)"


# ------------------------------------------------------------------------------------------------ NMS
def _canon(t):
    """rows ordered by (-score, then x1,y1,x2,y2,cls): removes the arbitrary order the reference's unstable argsort
    gives to EXACT score ties (apriori label rows all have conf 1.0)"""
    t = t.float().cpu()
    keys = torch.stack((-t[:, 4], t[:, 0], t[:, 1], t[:, 2], t[:, 3], t[:, 5]), 1).tolist()
    order = sorted(range(len(keys)), key=lambda i: keys[i])
    return t[order]


def _cmp_nms(res, gold, what=""):
    assert len(res) == len(gold)
    for i, (a, b) in enumerate(zip(res, gold)):
        a = a.float().cpu()
        assert a.shape == b.shape, f"{what} image {i}: {tuple(a.shape)} vs {tuple(b.shape)}"
        assert torch.equal(a, b.float()), f"{what} image {i}: rows differ (first bad row {(a != b.float()).any(1).nonzero()[:1].tolist()})"


def test_nms_known_answer(dev, golden_dir):
    from yolov3_amd import non_max_suppression

    gold = torch.load(golden_dir / "nms.pt")
    p = torch.tensor(
        [[[50, 50, 20, 20, 0.9, 0.9, 0.5, 0.0], [200, 200, 30, 30, 0.8, 0.1, 0.2, 0.95], [52, 51, 20, 20, 0.7, 0.8, 0.6, 0.0], [400, 400, 10, 10, 0.0005, 0.9, 0.9, 0.9]]]
    ).to(dev)
    _cmp_nms(non_max_suppression(p, 0.001, 0.6, multi_label=True), gold["kat_val"], "kat_val")
    _cmp_nms(non_max_suppression(p, 0.25, 0.45), gold["kat_det"], "kat_det")
    _cmp_nms(non_max_suppression(p, 0.25, 0.45, classes=[2]), gold["kat_cls2"], "kat_cls2")
    _cmp_nms(non_max_suppression(p, 0.25, 0.45, agnostic=True), gold["kat_agn"], "kat_agn")
    _cmp_nms(non_max_suppression((p, [None]), 0.25, 0.45), gold["kat_det"], "tuple input")
    with pytest.raises(AssertionError, match="Invalid Confidence threshold"):
        non_max_suppression(p, 1.5, 0.45)
    with pytest.raises(AssertionError, match="Invalid IoU"):
        non_max_suppression(p, 0.5, -0.1)


NMS_CASES = ["val_fp32", "det_fp32", "det_agnostic", "det_classes", "val_nc3_maxdet", "single_class", "det_fp16", "all_filtered"]


@pytest.mark.parametrize("name", NMS_CASES)
def test_nms_vs_reference_golden(dev, golden_dir, name):
    from yolov3_amd import non_max_suppression

    rec = torch.load(golden_dir / "nms.pt")[name]
    gk = dict(rec["gen"])
    if "dtype" in gk:
        gk["dtype"] = getattr(torch, gk["dtype"].split(".")[-1])
    pred = yo.synth_predictions(**gk)
    assert checksum(pred) == rec["in_sum"]
    before = pred.clone()
    res = non_max_suppression(pred.to(dev), **rec["nms"])
    _cmp_nms(res, rec["out"], name)
    assert torch.equal(pred, before)
    assert all(r.dtype == torch.float32 and r.device.type == "cuda" for r in res)


def test_nms_labels_vs_reference_golden(dev, golden_dir):
    from yolov3_amd import non_max_suppression

    rec = torch.load(golden_dir / "nms.pt")["labels"]
    pred = yo.synth_predictions(bs=2, n_rows=800, nc=80, seed=10)
    res = non_max_suppression(pred.to(dev), 0.25, 0.45, labels=rec["lb"])
    _cmp_nms([_canon(r) for r in res], [_canon(r) for r in rec["out"]], "labels")


@pytest.mark.parametrize(
    "gen,kw",
    [
        (dict(bs=4, n_rows=25200, nc=80, seed=12), dict(conf_thres=0.001, iou_thres=0.6, multi_label=True, max_det=300)),  # val.py:374 regime, full size
        (dict(bs=4, n_rows=25200, nc=80, seed=13), dict(conf_thres=0.25, iou_thres=0.45, max_det=1000)),  # detect.py:200 regime
        (dict(bs=2, n_rows=25200, nc=80, seed=14, dtype=torch.float16, hits=0.002), dict(conf_thres=0.25, iou_thres=0.45)),
        (dict(bs=2, n_rows=6000, nc=80, seed=15, hits=0.5), dict(conf_thres=0.001, iou_thres=0.6, multi_label=True, agnostic=True)),
        (dict(bs=2, n_rows=100800, nc=365, seed=16, hits=0.01), dict(conf_thres=0.01, iou_thres=0.6, multi_label=True)),  # config 5 head
    ],
    ids=["val_full", "detect_full", "detect_fp16", "agnostic_dense", "objects365_1280"],
)
def test_nms_vs_oracle_full_size(dev, gen, kw):
    from yolov3_amd import non_max_suppression

    pred = yo.synth_predictions(**gen)
    if pred.dtype == torch.float16:  # ties make the reference undefined; the oracle uses the same stable order we do
        pass
    ref = yo.non_max_suppression(pred, **kw)
    res = non_max_suppression(pred.to(dev), **kw)
    _cmp_nms(res, ref, str(kw))


def test_nms_capacity_overflow_retry_and_big_boxes(dev):
    """(a) > default capacity candidates -> the adapter re-runs with a larger workspace, result still exact;
    (b) boxes wider than max_wh/2 disable per-class partitioning (cross-class overlap is possible) -> same answer
    as the oracle's single global NMS; (c) > max_nms candidates are cut in score order."""
    from yolov3_amd import non_max_suppression

    g = torch.Generator().manual_seed(3)
    pred = torch.rand(1, 3000, 25, generator=g)
    pred[..., :2] *= 600
    pred[..., 2:4] *= 80
    pred[..., 4] = 0.5 + 0.5 * pred[..., 4]
    kw = dict(conf_thres=0.001, iou_thres=0.6, multi_label=True, max_det=300)  # 3000*20 = 60k candidates > 16384 and > max_nms
    _cmp_nms(non_max_suppression(pred.to(dev), **kw), yo.non_max_suppression(pred, **kw), "overflow")
    big = pred[:, :400].clone()
    big[0, :50, 2:4] = 9000.0  # spans several class offsets
    kw = dict(conf_thres=0.3, iou_thres=0.2, multi_label=True, max_det=300)
    _cmp_nms(non_max_suppression(big.to(dev), **kw), yo.non_max_suppression(big, **kw), "big boxes")


@pytest.mark.parametrize("max_det", [37, 64, 65, 300, 1000])
@pytest.mark.parametrize("agnostic", [False, True])
def test_nms_blocked_greedy_many_kept_boxes(dev, max_det, agnostic):
    """nms_greedy_kernel walks a segment in blocks of 64 boxes (suppression rows of a block in parallel, one wave resolves them in order, the block's kept boxes
    suppress what follows).  The load that made the one-box-at-a-time loop slow -- boxes that rarely suppress each other, hundreds kept per segment, like the
    predictions of the benchmark's model -- with the kept count crossing max_det inside a block, at a block boundary and never; segments of 1 .. 4000 boxes
    (3 classes + one image with a single box; agnostic: one segment per image).  Bit-exact against the oracle's torchvision loop."""
    from yolov3_amd import non_max_suppression

    g = torch.Generator().manual_seed(100 + max_det)
    bs, n_rows, nc = 3, 4000, 3
    pred = torch.zeros(bs, n_rows, 5 + nc)
    pred[..., :2] = torch.rand(bs, n_rows, 2, generator=g) * 600 + 20
    pred[..., 2:4] = torch.rand(bs, n_rows, 2, generator=g) * 50 + 8      # 8 .. 58 px boxes on a 640 px image: a few neighbours overlap by > 0.6, most do not
    pred[..., 4] = torch.rand(bs, n_rows, generator=g) * 0.9 + 0.05
    pred[..., 5:] = torch.rand(bs, n_rows, nc, generator=g)
    pred[1, :, 5 + 1] = 0.0                                                 # image 1: class 1 has no candidates (an empty segment between two full ones)
    pred[2, 1:, 4] = 0.0                                                    # image 2: one box in all
    kw = dict(conf_thres=0.25, iou_thres=0.6, multi_label=True, max_det=max_det, agnostic=agnostic)
    _cmp_nms(non_max_suppression(pred.to(dev), **kw), yo.non_max_suppression(pred, **kw), f"max_det {max_det} agnostic {agnostic}")


def test_nms_properties(dev):
    """size-independent properties: idempotence (NMS of survivors keeps them all), score order, max_det cap"""
    from yolov3_amd import non_max_suppression

    pred = yo.synth_predictions(bs=8, n_rows=25200, nc=80, seed=33).to(dev)
    out = non_max_suppression(pred, 0.25, 0.45, max_det=300)
    for o in out:
        assert o.shape[0] <= 300 and o.shape[1] == 6
        assert torch.all(o[1:, 4] <= o[:-1, 4]), "not in descending score order"
        if o.shape[0]:
            # rebuild a prediction tensor from the survivors: nothing more may be suppressed
            n = o.shape[0]
            p2 = torch.zeros(1, n, 85, device=dev)
            p2[0, :, 0] = (o[:, 0] + o[:, 2]) / 2
            p2[0, :, 1] = (o[:, 1] + o[:, 3]) / 2
            p2[0, :, 2] = o[:, 2] - o[:, 0]
            p2[0, :, 3] = o[:, 3] - o[:, 1]
            p2[0, :, 4] = 1.0
            p2[0, torch.arange(n), 5 + o[:, 5].long()] = o[:, 4]
            again = non_max_suppression(p2, 0.25, 0.45, max_det=300)[0]
            assert again.shape[0] == n, f"idempotence: {n} -> {again.shape[0]}"


@pytest.mark.parametrize("case", ["ties_fp16", "nc600_two_class_passes", "bf16", "max_nms_cut_fp32", "ragged_images", "agnostic_fp16"])
def test_nms_block_sort_equals_device_sorts_and_oracle(dev, tune, case):
    """nms_sort_kernel (one block per image: stable counting passes by score digits, then by class) against the two rocPRIM device sorts it replaces (knob
    nms_sort = 0) and against the oracle: heavy score ties (stability = nonzero order), more than 512 classes (two class passes), bf16 / fp32 score digits,
    the max_nms cut in score order, images with 0 / 1 / a few / many candidates in one batch, the single-segment (agnostic) form."""
    from yolov3_amd import ops

    g = torch.Generator().manual_seed(77)
    kw = dict(conf_thres=0.05, iou_thres=0.5, classes=None, agnostic=False, multi_label=True, max_det=300)
    extra = {}
    oracle = True
    if case == "ties_fp16":
        pred = yo.synth_predictions(bs=3, n_rows=5000, nc=20, seed=5, dtype=torch.float16, hits=0.3)
        pred[..., 4] = (pred[..., 4].float() * 8).round().div(8).half()         # 9 objectness levels
        pred[..., 5:] = (pred[..., 5:].float() * 4).round().div(4).half()       # 5 class-score levels: thousands of exact ties per image
    elif case == "nc600_two_class_passes":
        pred = torch.rand(2, 1500, 5 + 600, generator=g)
        pred[..., :2] *= 600
        pred[..., 2:4] = pred[..., 2:4] * 60 + 4
        pred[..., 5:] *= (torch.rand(2, 1500, 600, generator=g) < 0.02)          # ~12 labels per row, classes 0 .. 599
    elif case == "bf16":
        pred = yo.synth_predictions(bs=2, n_rows=6000, nc=80, seed=6, hits=0.1).to(torch.bfloat16)
        oracle = False                                                            # (the CPU reference has no bf16 path for every op; the device sorts are the check)
    elif case == "max_nms_cut_fp32":
        pred = yo.synth_predictions(bs=2, n_rows=4000, nc=30, seed=7, hits=0.5)
        extra = dict(max_nms=1000)
        oracle = False                                                            # (max_nms is a constant inside the reference function)
    elif case == "ragged_images":
        pred = yo.synth_predictions(bs=5, n_rows=3000, nc=10, seed=8, dtype=torch.float16, hits=0.4)
        pred[0, :, 4] = 0                                                         # no candidate
        pred[2, 1:, 4] = 0                                                        # one row
        pred[3, 70:, 4] = 0                                                       # fewer than one group per wave
    else:
        pred = yo.synth_predictions(bs=2, n_rows=5000, nc=80, seed=9, dtype=torch.float16, hits=0.2)
        kw["agnostic"] = True

    def run(form):
        tune("nms_sort", form)
        rows, counts = ops.nms_raw(pred.to(dev), kw["conf_thres"], kw["iou_thres"], None, kw["agnostic"], True, kw["max_det"], **extra)
        return [rows[i, :c].cpu() for i, c in enumerate(counts)]

    own, old = run(1), run(0)
    assert sum(o.shape[0] for o in own) > 50, "the case keeps too few boxes to say anything"
    _cmp_nms(own, old, case + " (own sort vs device sorts)")
    if oracle:
        okw = {k: v for k, v in kw.items() if k != "classes"}
        _cmp_nms(own, yo.non_max_suppression(pred, **okw), case + " (oracle)")


# ------------------------------------------------------------------------------------------------ full model
def build_pair(name, nc, seed, dev, dtype):
    from yolov3_amd import DetectionModel

    d = yaml.safe_load(open(CFG / f"{name}.yaml"))
    layers, save, anchors, nc_v = yo.parse_cfg(d, 3, nc)
    strides = yo.model_strides(layers)
    sd = yo.seeded_state_dict(layers, nc_v, anchors, strides, seed=seed)
    m = DetectionModel(f"{name}.yaml", nc=nc)
    m.load_state_dict(sd)
    m = m.to(dev).to(dtype).eval()
    return m, (layers, save, sd, strides)


@pytest.mark.parametrize("key", ["yolov3-tiny-nc80-64-bs2", "yolov3-nc80-64-bs2", "yolov3-spp-nc80-64-bs1", "yolov3-nc7-96-bs1"])
def test_model_fp32_vs_reference_golden(dev, golden_dir, key):
    """fp32 engine (direct HIP kernels) against the UNMODIFIED reference's eval output: 1e-4 on raw logits,
    1e-4 relative on decoded boxes (north_star tolerance)."""
    gold = torch.load(golden_dir / "model_fwd.pt")[key]
    name, nc, hw, bs = key.rsplit("-", 3)
    nc, hw, bs = int(nc[2:]), int(hw), int(bs[2:])
    m, _ = build_pair(name, nc, 11, dev, torch.float32)
    x = torch.rand(bs, 3, hw, hw, generator=torch.Generator().manual_seed(5))
    assert checksum(x) == gold["x_sum"]
    pred, raw = m(x.to(dev))
    torch.cuda.synchronize()
    for a, b in zip(raw, gold["eval_raw"]):
        err = (a.cpu() - b).abs().max().item()
        assert err < 1e-4, f"{key}: raw logits max abs err {err:.3g}"
    torch.testing.assert_close(pred.cpu(), gold["eval_pred"], rtol=1e-4, atol=1e-4)
    m.fuse()
    predf, _ = m(x.to(dev))
    torch.testing.assert_close(predf.cpu(), gold["fused_pred"], rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize("name,hw,bs,dtype", [("yolov3-tiny", 416, 4, torch.float16), ("yolov3", 128, 2, torch.float16), ("yolov3-spp", 128, 2, torch.float16), ("yolov3", 128, 2, torch.bfloat16)])
def test_model_half_vs_fp32_oracle(dev, name, hw, bs, dtype):
    """MFMA engine (fp16/bf16 storage, fp32 accumulate) against the fp32 CPU oracle with the same weights.
    75 layers of half-precision storage rounding: compare raw logits with a tolerance relative to the logit scale."""
    m, (layers, save, sd, strides) = build_pair(name, 80, 21, dev, dtype)
    x = torch.rand(bs, 3, hw, hw, generator=torch.Generator().manual_seed(6))
    pred, raw = m(x.to(dev))
    torch.cuda.synchronize()
    with torch.no_grad():
        refp, refraw = yo.forward(layers, save, sd, x, strides, training=False)
    bnd = HALF_BOUNDS[dtype]
    for lvl, (a, b) in enumerate(zip(raw, refraw)):
        rms, mx, corr = _rel_errors(a.float().cpu(), b)
        print(f"[half-vs-oracle {name} {hw} {dtype}] level {lvl}: rel RMS {rms:.5f}  max/range {mx:.5f}  corr {corr:.6f}")
        assert rms < bnd["rms"] and mx < bnd["mx"] and corr > bnd["corr"], f"{name} {dtype} level {lvl}: rms {rms:.4g} max {mx:.4g} corr {corr:.6f}"
    assert pred.shape == refp.shape and pred.dtype == dtype


@pytest.mark.parametrize("key", ["yolov3-tiny-nc80-416-bs2", "yolov3-nc80-640-bs1", "yolov3-spp-nc80-640-bs1"])
def test_model_fp32_vs_reference_golden_benchmark_resolutions(dev, golden_dir, key):
    """fp32 engine against the UNMODIFIED reference at 416x416 (BASELINE configs[0]) and 640x640 (configs[1] / [3]): sampled raw logits
    within 1e-4, decoded boxes 1e-4 relative, whole-tensor |.| sums 1e-5 relative (north_star tolerance)."""
    gold = torch.load(golden_dir / "model_fwd_big.pt")[key]
    name, nc, hw, bs = key.rsplit("-", 3)
    nc, hw, bs = int(nc[2:]), int(hw), int(bs[2:])
    m, _ = build_pair(name, nc, 11, dev, torch.float32)
    x = torch.rand(bs, 3, hw, hw, generator=torch.Generator().manual_seed(5))
    assert checksum(x) == gold["x_sum"]
    pred, raw = m(x.to(dev))
    torch.cuda.synchronize()
    st = gold["step"]
    for a, rows, sm in zip(raw, gold["raw_rows"], gold["raw_sum"]):
        a = a.cpu()
        err = (a.reshape(a.shape[0], -1, a.shape[-1])[:, ::st] - rows).abs().max().item()
        assert err < 1e-4, f"{key}: raw logits max abs err {err:.3g}"
        assert abs(checksum(a) - sm) < 1e-5 * sm
    torch.testing.assert_close(pred.cpu()[:, ::st], gold["pred_rows"], rtol=1e-4, atol=1e-4)
    assert abs(checksum(pred.cpu()) - gold["pred_sum"]) < 1e-5 * gold["pred_sum"]


def _rel_errors(a, b):
    """(relative RMS error, max abs error / max |b|, Pearson correlation) of a against the reference b"""
    a, b = a.double().flatten(), b.double().flatten()
    rms = ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()
    mx = ((a - b).abs().max() / b.abs().max()).item()
    corr = torch.corrcoef(torch.stack((a, b)))[0, 1].item()
    return rms, mx, corr


# Bounds for the half-precision engines against the fp32 oracle with the same weights, raw logits of every level.  Model: each of the
# ~75 stored tensors is rounded once (relative 2^-11 fp16 / 2^-8 bf16, uniform), errors of successive layers add in quadrature and the
# residual trunk carries them forward: relative RMS ~ sqrt(75) * eps / sqrt(3) = 0.24 % fp16 / 2.0 % bf16 if every layer's error survived
# to the output; BatchNorm-folded convolutions average most of it away.  Measured on MI355X (round 2, gpurun d_pytest.log): fp16 rel RMS
# 0.00031-0.00034, max/range 0.0006-0.0008, corr 0.999999; bf16 rel RMS 0.0019-0.0027, max/range 0.0046-0.0062, corr 0.99982-0.99993 --
# the same at 128x128, 640x640 and 1280x1280.  The asserted bounds sit ~3x above the measurements.
HALF_BOUNDS = {torch.float16: dict(rms=0.001, mx=0.003, corr=0.99999), torch.bfloat16: dict(rms=0.008, mx=0.02, corr=0.9995)}


@pytest.mark.parametrize("name,hw,bs,dtype", [("yolov3", 640, 12, torch.float16), ("yolov3-spp", 640, 12, torch.float16), ("yolov3", 640, 4, torch.bfloat16),
                                              ("yolov3", 1280, 2, torch.bfloat16), ("yolov3", 640, 32, torch.float16)])   # the last: BASELINE configs[1] at ITS batch
def test_model_half_vs_fp32_oracle_benchmark_shapes(dev, name, hw, bs, dtype):
    """The BENCHMARKED engines at their own resolution (640x640 fp16 yolov3 / yolov3-spp = configs[1] / [3]; bf16 at 640 and 1280 =
    configs[4]'s dtype and map sizes) against the fp32 CPU oracle: every conv launch goes through the variants the bench runs
    (v10 / v10h / v10k / v6 / v3 with multi-round grids), and the error is bounded per level in relative RMS, max-abs and correlation."""
    nc = 80 if hw == 640 else 365
    m, (layers, save, sd, strides) = build_pair(name, nc, 21, dev, dtype)
    x = torch.rand(bs, 3, hw, hw, generator=torch.Generator().manual_seed(6))
    pred, raw = m(x.to(dev).to(dtype))
    torch.cuda.synchronize()
    plan = next(iter(m._plans.values()))
    variants = {plan.conv_variant(ln) for ln in plan.launches if ln.flops and not ln.kernel}
    assert any(v in ("v10k", "v10", "v10h") for v in variants) and "direct" not in variants, variants
    if bs >= 12 and hw == 640:
        assert "s1x1" in variants, variants   # the persistent 1x1 kernel takes the 80 x 80 / 160 x 160 cv1 layers from 32768 pixels up
    with torch.no_grad():
        refp, refraw = yo.forward(layers, save, sd, x[: (2 if bs >= 32 else min(bs, 4))], strides, training=False)   # the oracle on the first images (CPU time)
    b = HALF_BOUNDS[dtype]
    for lvl, (a, r) in enumerate(zip(raw, refraw)):
        rms, mx, corr = _rel_errors(a[: r.shape[0]].float().cpu(), r)
        print(f"[half-vs-oracle {name} {hw} {dtype}] level {lvl}: rel RMS {rms:.5f}  max/range {mx:.5f}  corr {corr:.6f}")
        assert rms < b["rms"] and mx < b["mx"] and corr > b["corr"], f"{name} {hw} {dtype} level {lvl}: rms {rms:.4g} max {mx:.4g} corr {corr:.6f}"
    assert pred.shape[0] == bs and pred.dtype == dtype and torch.isfinite(pred.float()).all()


def test_end_to_end_detections_fp32(dev):
    """fp32 engine forward + HIP NMS == oracle forward + oracle NMS on the HIP prediction tensor (index-exact),
    and close to the all-oracle pipeline."""
    from yolov3_amd import non_max_suppression

    m, (layers, save, sd, strides) = build_pair("yolov3-tiny", 80, 23, dev, torch.float32)
    x = torch.rand(2, 3, 160, 160, generator=torch.Generator().manual_seed(7))
    pred, _ = m(x.to(dev))
    res = non_max_suppression(pred, 0.001, 0.6, multi_label=True)
    ref = yo.non_max_suppression(pred.cpu(), 0.001, 0.6, multi_label=True)
    _cmp_nms(res, ref, "e2e")


# ------------------------------------------------------------------------------------------------ loss
def _loss_setup(dev, name, nc, hw, hyp):
    from yolov3_amd import ComputeLoss, DetectionModel

    m = DetectionModel(f"{name}.yaml", nc=nc).to(dev)
    m.hyp = hyp
    return m, ComputeLoss(m)


LOSS_CASES = ["yolov3-nc80-128-synth", "yolov3-tiny-nc80-96-synth", "yolov3-nc80-64-empty", "yolov3-nc5-64-dups", "yolov3-nc5-64-edges", "yolov3-nc5-64-dups_sorted"]


@pytest.mark.parametrize("key", LOSS_CASES)
def test_loss_vs_reference_golden(dev, golden_dir, key):
    """ComputeLoss value, items and d loss / d predictions against the UNMODIFIED reference (fp32): 1e-4 (north_star)."""
    rec = torch.load(golden_dir / "loss.pt")[key]
    name, nc, hw, mode = key.rsplit("-", 3)
    nc, hw = int(nc[2:]), int(hw)
    m, crit = _loss_setup(dev, name, nc, hw, rec["hyp"])
    crit.sort_obj_iou = mode.endswith("_sorted")   # ComputeLoss.sort_obj_iou (utils/loss.py:101,156-158): a cell matched several times keeps its largest iou
    strides = [int(s) for s in m.stride.tolist()]
    bs = rec["bs"]
    p_cpu = yo.synth_raw_predictions([(bs, 3, hw // s, hw // s, nc + 5) for s in strides], seed=31)
    assert sum(checksum(t) for t in p_cpu) == rec["p_sum"]
    torch.testing.assert_close(m.model[-1].anchors.cpu(), rec["anchors_grid"])
    p = [t.to(dev).requires_grad_(True) for t in p_cpu]
    loss, items = crit(p, rec["targets"].to(dev))
    loss.backward()
    torch.cuda.synchronize()
    assert loss.shape == (1,) and items.shape == (3,) and not items.requires_grad
    torch.testing.assert_close(loss.detach().cpu(), rec["loss"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(items.cpu(), rec["items"], rtol=1e-4, atol=1e-6)
    for a, b in zip(p, rec["grads"]):
        torch.testing.assert_close(a.grad.cpu(), b, rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("variant", ["fp32_scaled_grad", "focal", "smoothing_pw", "fp16", "sort_obj_iou"])
def test_loss_vs_oracle_variants(dev, variant):
    hyp = dict(box=0.05, cls=0.5, cls_pw=1.0, obj=1.0, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0, label_smoothing=0.0)
    if variant == "focal":
        hyp["fl_gamma"] = 1.5
    if variant == "smoothing_pw":
        hyp.update(label_smoothing=0.1, cls_pw=0.7, obj_pw=1.3)
    nc, hw, bs = 80, 160, 4
    m, crit = _loss_setup(dev, "yolov3", nc, hw, hyp)
    p_cpu = yo.synth_raw_predictions([(bs, 3, hw // s, hw // s, nc + 5) for s in (8, 16, 32)], seed=5)
    tg = yo.synth_targets(bs, nc, seed=3)
    if variant == "sort_obj_iou":   # ComputeLoss.sort_obj_iou (utils/loss.py:156-158) on cells matched twice by boxes of different size: the larger iou must stay
        tg = torch.cat((tg, tg[: tg.shape[0] // 2] * torch.tensor([1, 1, 1, 1, 0.8, 1.25])))
        crit.sort_obj_iou = True
        plain = yo.compute_loss([t.clone() for t in p_cpu], tg, m.model[-1].anchors.cpu(), hyp, nc)[1]
    dtype = torch.float16 if variant == "fp16" else torch.float32
    p_ref = [t.to(dtype).float().clone().requires_grad_(True) for t in p_cpu]
    ref_loss, ref_items, _ = yo.compute_loss(p_ref, tg, m.model[-1].anchors.cpu(), hyp, nc, sort_obj_iou=variant == "sort_obj_iou")
    if variant == "sort_obj_iou":
        assert abs(float(plain[1]) - float(ref_items[1])) > 1e-5 * float(ref_items[1]), "the case does not distinguish the two orders"
    scale = 1024.0 if variant in ("fp32_scaled_grad", "fp16") else 1.0  # GradScaler-style upstream gradient
    (ref_loss * scale).sum().backward()
    p = [t.detach().to(dev).to(dtype).requires_grad_(True) for t in p_cpu]
    loss, items = crit(p, tg.to(dev))
    (loss * scale).sum().backward()
    torch.cuda.synchronize()
    tol = dict(rtol=1e-4, atol=1e-5) if dtype == torch.float32 else dict(rtol=2e-3, atol=2e-3)  # fp16: tobj is rounded to half like the reference's autocast path
    torch.testing.assert_close(loss.detach().cpu(), ref_loss.detach(), **tol)
    torch.testing.assert_close(items.cpu(), ref_items, **tol)
    for a, b in zip(p, p_ref):
        g = a.grad.float().cpu()
        assert a.grad.dtype == dtype
        if dtype == torch.float32:
            torch.testing.assert_close(g, b.grad, rtol=1e-4, atol=1e-6 * scale)
        else:
            rel = (g - b.grad).abs().max().item() / b.grad.abs().max().item()
            assert rel < 2e-3, rel


def test_loss_full_size_properties(dev):
    """config-3 shapes (bs 16 of the 64/GPU, 640x640): finite, deterministic run-to-run, zero gradient on unmatched
    non-objectness channels, gradient linear in the upstream scale."""
    hyp = dict(box=0.05, cls=0.5, cls_pw=1.0, obj=1.0, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0, label_smoothing=0.0)
    nc, hw, bs = 80, 640, 16
    m, crit = _loss_setup(dev, "yolov3", nc, hw, hyp)
    p = [(torch.rand(bs, 3, hw // s, hw // s, nc + 5, device=dev) * 6 - 3).requires_grad_(True) for s in (8, 16, 32)]
    tg = yo.synth_targets(bs, nc, seed=9).to(dev)
    l1, it1 = crit(p, tg)
    l1.backward()
    g1 = [t.grad.clone() for t in p]
    for t in p:
        t.grad = None
    l2, it2 = crit(p, tg)
    (l2 * 3.0).sum().backward()
    torch.cuda.synchronize()
    assert torch.isfinite(l1).all() and torch.equal(l1, l2) and torch.equal(it1, it2)
    for a, t in zip(g1, p):
        torch.testing.assert_close(t.grad, a * 3.0, rtol=1e-5, atol=1e-9)
        others = a.clone()
        others[..., 4] = 0
        assert (others.abs().sum(-1) > 0).float().mean().item() < 0.05  # only matched cells carry box/cls gradient


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_loss_backward_is_bit_deterministic_on_duplicated_cells(dev, dtype):
    """The reference trains under torch.use_deterministic_algorithms(True) (train.py:191, utils/general.py:191-205).  The loss backward has no floating-point atomics:
    every matched slot writes its own gradient row and the winner slot of a cell that several slots matched adds the rows in slot order (loss_scatter_kernel).
    Targets stacked three deep on the same cells (every cell matched 3+ times, by boxes of different size) with shuffled rows: five backward passes are bit-identical,
    and the summed rows agree with the oracle's autograd (which accumulates duplicates by index_put)."""
    hyp = dict(box=0.05, cls=0.5, cls_pw=1.0, obj=1.0, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0, label_smoothing=0.0)
    nc, hw, bs = 80, 320, 8
    m, crit = _loss_setup(dev, "yolov3", nc, hw, hyp)
    p_cpu = yo.synth_raw_predictions([(bs, 3, hw // s, hw // s, nc + 5) for s in (8, 16, 32)], seed=11)
    base = yo.synth_targets(bs, nc, seed=6)
    tg = torch.cat((base, base * torch.tensor([1, 1, 1, 1, 0.9, 1.1]), base * torch.tensor([1, 1, 1, 1, 1.15, 0.85])))
    tg[:, 1] = tg[:, 1].round().clamp(0, nc - 1)
    tg = tg[torch.randperm(tg.shape[0], generator=torch.Generator().manual_seed(1))]
    p = [t.to(dev).to(dtype).requires_grad_(True) for t in p_cpu]
    runs = []
    for r in range(5):
        for t in p:
            t.grad = None
        loss, _ = crit(p, tg.to(dev))
        (loss * 512.0).sum().backward()
        torch.cuda.synchronize()
        runs.append([t.grad.clone() for t in p])
        if r == 1:   # other work in between: the order in which waves retire changes, the sums must not
            torch.randn(1 << 22, device=dev).sort()
    for other in runs[1:]:
        for a, b in zip(runs[0], other):
            assert torch.equal(a, b), "loss backward differs between two runs on the same inputs"
    if dtype == torch.float32:
        p_ref = [t.clone().requires_grad_(True) for t in p_cpu]
        ref_loss, _, _ = yo.compute_loss(p_ref, tg, m.model[-1].anchors.cpu(), hyp, nc)
        (ref_loss * 512.0).sum().backward()
        for a, b in zip(runs[0], p_ref):
            torch.testing.assert_close(a.cpu(), b.grad, rtol=1e-4, atol=1e-6 * 512.0)


# ------------------------------------------------------------------------------------------------ training
@pytest.mark.parametrize("key", ["yolov3-tiny-nc80-64-bs2", "yolov3-nc80-64-bs2"])
def test_train_forward_vs_reference_golden(dev, golden_dir, key):
    """train-mode forward (batch-statistics BN) in fp32 against the unmodified reference: raw logits within 1e-4,
    running statistics updated like nn.BatchNorm2d(momentum 0.03)."""
    gold = torch.load(golden_dir / "model_fwd.pt")[key]
    name, nc, hw, bs = key.rsplit("-", 3)
    nc, hw, bs = int(nc[2:]), int(hw), int(bs[2:])
    m, (layers, save, sd, strides) = build_pair(name, nc, 11, dev, torch.float32)
    m.train()
    x = torch.rand(bs, 3, hw, hw, generator=torch.Generator().manual_seed(5))
    raws = m(x.to(dev))
    torch.cuda.synchronize()
    for a, b in zip(raws, gold["train_raw"]):
        err = (a.detach().cpu() - b).abs().max().item()
        assert err < 1e-4, f"{key}: train-mode raw logits max abs err {err:.3g}"
    stats = {}
    with torch.no_grad():
        yo.forward(layers, save, sd, x, strides, training=True, stats=stats)
    own = m.state_dict()
    for k, v in stats.items():
        torch.testing.assert_close(own[k].cpu(), v, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("name,hw", [("yolov3-tiny", 96), ("yolov3", 64), ("yolov3-spp", 64)])
def test_train_step_gradients_vs_oracle_autograd(dev, name, hw):
    """forward + ComputeLoss + backward on the GPU (fp32) against torch autograd over the CPU oracle: every parameter
    gradient (75 conv filters, 72 BN gamma/beta, Detect convs) within 2e-3 of the tensor's gradient scale."""
    from yolov3_amd import ComputeLoss

    nc, bs = 80, 2
    hyp = dict(box=0.05, cls=0.5, cls_pw=1.0, obj=1.0, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0, label_smoothing=0.0)
    m, (layers, save, sd, strides) = build_pair(name, nc, 17, dev, torch.float32)
    m.train()
    m.hyp = hyp
    x = torch.rand(bs, 3, hw, hw, generator=torch.Generator().manual_seed(8))
    tg = yo.synth_targets(bs, nc, seed=4)
    # oracle
    sdg = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items()}
    raws_ref = yo.forward(layers, save, sdg, x, strides, training=True)
    loss_ref, _, _ = yo.compute_loss(raws_ref, tg, sd[[k for k in sd if k.endswith("anchors")][0]], hyp, nc)
    loss_ref.backward()
    # HIP
    crit = ComputeLoss(m)
    raws = m(x.to(dev))
    loss, items = crit(raws, tg.to(dev))
    loss.backward()
    torch.cuda.synchronize()
    torch.testing.assert_close(loss.detach().cpu(), loss_ref.detach(), rtol=1e-4, atol=1e-5)
    worst = []
    for k, p in m.named_parameters():
        ref = sdg[k].grad
        assert p.grad is not None, f"{k}: no gradient"
        g = p.grad.cpu()
        rel = (g - ref).abs().max().item() / (ref.abs().max().item() + 1e-12)
        worst.append((rel, k))
    worst.sort(reverse=True)
    assert worst[0][0] < 2e-3, f"worst gradient mismatches: {worst[:5]}"


@pytest.mark.parametrize("name,hw,adt", [("yolov3", 128, torch.float16), ("yolov3-tiny", 160, torch.float16), ("yolov3-spp", 128, torch.float16), ("yolov3", 96, torch.bfloat16)])
def test_train_step_autocast_fp16(dev, name, hw, adt):
    """autocast(fp16 / bf16) training step through the MFMA kernels (stem kernel for layer 0, BatchNorm statistics from the conv
    epilogue, 256-tile and 128-tile filter gradients, stride-2 parity-class data gradients, SPP / max-pool / upsample
    backward): loss close to the fp32 oracle, finite gradients close to the oracle's autograd in direction, and an SGD step
    changes the next forward (plans re-pack the updated fp32 master weights)."""
    from yolov3_amd import ComputeLoss

    nc, bs = 80, 4
    hyp = dict(box=0.05, cls=0.5, cls_pw=1.0, obj=1.0, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0, label_smoothing=0.0)
    m, (layers, save, sd, strides) = build_pair(name, nc, 19, dev, torch.float32)
    m.train()
    m.hyp = hyp
    x = torch.rand(bs, 3, hw, hw, generator=torch.Generator().manual_seed(2))
    tg = yo.synth_targets(bs, nc, seed=6)
    sdg = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items()}
    raws_ref = yo.forward(layers, save, sdg, x, strides, training=True)
    loss_ref, _, _ = yo.compute_loss(raws_ref, tg, sd[[k for k in sd if k.endswith("anchors")][0]], hyp, nc)
    loss_ref.backward()
    crit = ComputeLoss(m)
    opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9)
    with torch.autocast("cuda", dtype=adt):
        raws = m(x.to(dev))
        loss, _ = crit(raws, tg.to(dev))
    assert raws[0].dtype == adt
    (loss * 128.0).backward()
    torch.cuda.synchronize()
    tol = 0.002 if adt == torch.float16 else 0.01   # measured (round 2): 1e-4..2e-4 fp16, 1.5e-3 bf16
    assert abs(loss.item() - loss_ref.item()) / loss_ref.item() < tol
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    # gradient direction against the oracle's fp32 autograd: cosine over the large tensors (half-precision activations and
    # gradients through up to 75 layers; bf16 keeps 8 mantissa bits)
    cos_min, worst = 1.0, None
    for k, p_ in m.named_parameters():
        ref = sdg[k].grad
        if ref is None or ref.numel() < 4096:
            continue
        g = p_.grad.float().cpu() / 128.0
        c = torch.nn.functional.cosine_similarity(g.flatten(), ref.flatten(), dim=0).item()
        if c < cos_min:
            cos_min, worst = c, k
    print(f"[autocast {name} {adt}] loss rel err {abs(loss.item() - loss_ref.item()) / loss_ref.item():.4f}, min gradient cosine {cos_min:.4f} at {worst}")
    # measured (round 2): fp16 0.9919-0.9961; bf16 0.9217-0.9673 -- the MIN over ~60 tensors of a quantity set by 8-bit-mantissa
    # rounding noise: two builds whose fp32 arithmetic differs only in instruction selection (packed vs scalar fp32 in the BN kernels)
    # land anywhere in that range, on a different tensor each time
    assert cos_min > (0.985 if adt == torch.float16 else 0.90), f"gradient direction: cosine {cos_min:.4f} at {worst}"
    for p in m.parameters():
        p.grad /= 128.0
    opt.step()
    with torch.autocast("cuda", dtype=adt):
        loss2, _ = crit(m(x.to(dev)), tg.to(dev))
    assert loss2.item() != loss.item()


WGRAD_CASES = [
    ("3x3s1", (2, 20, 20, 64, 128, 3, 1), {}),
    ("3x3s2_odd", (2, 23, 19, 128, 256, 3, 2), {}),
    ("first_layer", (2, 40, 36, 8, 32, 3, 1), {"cin_real": 3}),
    ("1x1_deep", (2, 10, 10, 1024, 512, 1, 1), {}),
    ("head_255", (2, 20, 20, 256, 256, 1, 1), {"cout_real": 255}),
    ("cin32_3x3", (1, 33, 17, 32, 64, 3, 1), {}),
    ("many_pixels", (4, 80, 80, 64, 64, 3, 1), {}),
    ("3x3s2_even_cin32", (2, 32, 48, 32, 64, 3, 2), {"dgrad_variant": "v3_quad"}),
    ("3x3s2_deep", (2, 10, 10, 512, 1024, 3, 2), {}),
    ("3x3s2_even_5tiles", (3, 36, 44, 64, 128, 3, 2), {"dgrad_variant": "v3_quad"}),     # 5 pixel tiles per class: the last group of 8 block ids is partial
    ("3x3s2_even_cin128", (2, 64, 64, 128, 256, 3, 2), {"dgrad_variant": "v3_quad"}),
]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("name,shape,kw", WGRAD_CASES, ids=[c[0] for c in WGRAD_CASES])
def test_conv_wgrad_and_dgrad_vs_autograd(dev, dtype, name, shape, kw):
    """filter gradient (MFMA kernel for f16/bf16, direct kernel for f32) and data gradient (forward kernel on the
    flipped filter bank, dilated input for stride 2) against torch autograd in fp32 on the same rounded operands."""
    _lib, ops = _ops()
    n, h, w, cin, cout, k, s = shape
    cin_real, cout_real = kw.get("cin_real", cin), kw.get("cout_real", cout)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, cin_real, h, w, generator=g).to(dtype).float().requires_grad_(True)
    wt = (torch.randn(cout_real, cin_real, k, k, generator=g) / math.sqrt(cin_real * k * k)).to(dtype).float().requires_grad_(True)
    y = F.conv2d(x, wt, None, stride=s, padding=k // 2)
    gy = torch.randn(y.shape, generator=g).to(dtype).float()
    y.backward(gy)
    ho, wo = y.shape[2], y.shape[3]
    xv = ops.View.alloc(n, h, w, cin, dtype, dev)
    ops.nchw_to_nhwc(x.detach().to(dev), xv)
    gv = ops.View.alloc(n, ho, wo, cout, dtype, dev)
    ops.nchw_to_nhwc(gy.to(dev), gv)
    dw, db = ops.conv2d_wgrad(xv, gv, k, s, cout_real, cin_real, want_bias=True)
    filt_d = ops.pack_filter_dgrad(wt.detach().to(dev), cout, cin, dtype)
    gx = ops.View.alloc(n, h, w, cin, dtype, dev)
    gx.buf.zero_()
    ops.conv2d(gv, filt_d, torch.zeros(cin, device=dev), gx, k, 1, act=False, residual=gx, in_dilation=s)
    torch.cuda.synchronize()
    tol = {torch.float32: 2e-5, torch.float16: 2e-3, torch.bfloat16: 1.5e-2}[dtype]
    if s == 2 and k == 3 and dtype != torch.float32:  # parity-class form of the stride-2 data gradient, write then accumulate
        g2 = ops.View.alloc(n, h, w, cin, dtype, dev)
        g2.buf.fill_(float("nan"))
        ops.conv2d_dgrad_s2(wt.detach().to(dev), gv, g2, accumulate=False)
        if "dgrad_variant" in kw:   # even sizes + a v3 tile: the four parity classes go out as one launch
            assert ops.last_conv_variant() == kw["dgrad_variant"]
        d2 = g2.as_nhwc().float().cpu().permute(0, 3, 1, 2)[:, :cin_real]
        assert (d2 - x.grad).abs().max().item() / x.grad.abs().max().item() < tol, "dgrad_s2 (write)"
        ops.conv2d_dgrad_s2(wt.detach().to(dev), gv, g2, accumulate=True)
        d3 = g2.as_nhwc().float().cpu().permute(0, 3, 1, 2)[:, :cin_real]
        assert (d3 - 2 * x.grad).abs().max().item() / x.grad.abs().max().item() < 2 * tol, "dgrad_s2 (accumulate)"
    e_w = (dw.cpu() - wt.grad).abs().max().item() / wt.grad.abs().max().item()
    e_b = (db.cpu() - gy.sum((0, 2, 3))[:cout_real]).abs().max().item() / gy.sum((0, 2, 3)).abs().max().item()
    dx = gx.as_nhwc().float().cpu().permute(0, 3, 1, 2)[:, :cin_real]
    e_x = (dx - x.grad).abs().max().item() / x.grad.abs().max().item()
    assert e_w < tol and e_b < max(tol, 2e-3) and e_x < tol, f"{name} {dtype}: wgrad {e_w:.2e} bias {e_b:.2e} dgrad {e_x:.2e}"


STRIP_WGRAD_CASES = [
    # name, (n, h, w, cin, cout, k, s), K-steps per block (knob wgrad_strip; 2 = the host's plan)
    ("c32_s1_one_strip", (2, 20, 64, 32, 64, 3, 1), 2),
    ("c32_s1_ragged_strips_walk7", (3, 19, 150, 32, 64, 3, 1), 7),      # 3 strips, the last 22 pixels wide; blocks cross strips and images
    ("c32_s2_walk5", (2, 46, 200, 32, 64, 3, 2), 5),                    # stride 2: two new input rows per K-step, 100 output columns
    ("c32_s2_odd", (2, 37, 131, 32, 64, 3, 2), 3),                      # odd input sizes: the last input row / column only under some taps
    ("c64_s1_two_halves_walk9", (2, 24, 96, 64, 128, 3, 1), 9),         # 128 filters = two 64-filter blocks per strip
    ("c64_s2_walk4", (2, 40, 140, 64, 128, 3, 2), 4),
    ("c64_s1_tall_one_block", (1, 70, 64, 64, 128, 3, 1), 70),          # one block walks a whole strip: the row ring wraps many times
]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("name,shape,per", STRIP_WGRAD_CASES, ids=[c[0] for c in STRIP_WGRAD_CASES])
def test_conv_wgrad_strip_vs_autograd(dev, tune, dtype, name, shape, per):
    """the strip-walking filter-gradient kernel (csrc/wgrad_strip.h: all nine taps from three resident input rows, one block per 64-pixel column strip
    segment) against torch autograd in fp32 on the same rounded operands, and against the tile kernel it replaces on these shapes (knob 0)"""
    _lib, ops = _ops()
    n, h, w, cin, cout, k, s = shape
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, cin, h, w, generator=g).to(dtype).float()
    wt = torch.zeros(cout, cin, k, k, requires_grad=True)
    y = F.conv2d(x, wt, None, stride=s, padding=1)
    gy = torch.randn(y.shape, generator=g).to(dtype).float()
    y.backward(gy)
    xv = ops.View.alloc(n, h, w, cin, dtype, dev)
    ops.nchw_to_nhwc(x.to(dev), xv)
    gv = ops.View.alloc(n, y.shape[2], y.shape[3], cout, dtype, dev)
    ops.nchw_to_nhwc(gy.to(dev), gv)
    tune("wgrad_strip", per)
    tile, blocks, _ = ops.conv2d_wgrad_plan(xv, cout, k, s)
    assert tile == 3 and blocks >= 1, (tile, blocks)
    dw, _ = ops.conv2d_wgrad(xv, gv, k, s, cout, cin)
    dw2, _ = ops.conv2d_wgrad(xv, gv, k, s, cout, cin)
    tune("wgrad_strip", 0)
    assert ops.conv2d_wgrad_plan(xv, cout, k, s)[0] == 128
    dw_old, _ = ops.conv2d_wgrad(xv, gv, k, s, cout, cin)
    torch.cuda.synchronize()
    assert torch.equal(dw, dw2), "not run-to-run deterministic"
    ref = wt.grad
    scale = ref.abs().max().item()
    e_w = (dw.cpu() - ref).abs().max().item() / scale
    e_o = (dw.cpu() - dw_old.cpu()).abs().max().item() / scale
    # same products, fp32 accumulation in another order: 1e-6-level against autograd and against the tile kernel
    assert e_w < 2e-5 and e_o < 2e-5, f"{name} {dtype}: vs autograd {e_w:.2e}, vs tile kernel {e_o:.2e}"


STRIP_QUAD_CASES = [
    # name, (n, h, w) of the 32-channel gradient, du rows per block (knob conv_strip; 2 = the host's plan)
    ("one_strip", (2, 40, 128), 2),
    ("ragged_strips_walk7", (3, 38, 300), 7),        # du is 19 x 150: 3 strips, the last 22 pixels wide; blocks cross strips and images
    ("tall_one_block", (1, 140, 128), 70),           # one block walks a whole strip: the row ring wraps many times
    ("narrow_map", (4, 80, 26), 3),
]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("chans", [(32, 64), (64, 128)], ids=["c32_64", "c64_128_two_roles"])
@pytest.mark.parametrize("name,shape,per", STRIP_QUAD_CASES, ids=[c[0] for c in STRIP_QUAD_CASES])
def test_conv_dgrad_s2_strip_quad_vs_autograd(dev, tune, dtype, chans, name, shape, per):
    """the stride-2 data gradient of a 32 -> 64 / 64 -> 128 layer through conv_strip_quad_kernel (csrc/conv_strip.h: du rows staged once for the nine (tap, parity class)
    pairs, the four classes' filters in registers) against torch autograd on the same rounded operands, write and accumulate forms, and against the
    tile kernel (knob conv_strip = 0: v3_quad)"""
    _lib, ops = _ops()
    n, h, w = shape
    cin, cout = chans   # of the layer: du has cout channels, the gradient cin (64 -> 128: two waves per (pixel tile, filter tile), split by parity class)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(n, cin, h, w, generator=g).to(dtype).float().requires_grad_(True)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)).to(dtype).float()
    y = F.conv2d(x, wt, None, stride=2, padding=1)
    gy = torch.randn(y.shape, generator=g).to(dtype).float()
    y.backward(gy)
    gv = ops.View.alloc(n, y.shape[2], y.shape[3], cout, dtype, dev)
    ops.nchw_to_nhwc(gy.to(dev), gv)
    tune("conv_strip", per)
    g2 = ops.View.alloc(n, h, w, cin, dtype, dev)
    g2.buf.fill_(float("nan"))
    ops.conv2d_dgrad_s2(wt.to(dev), gv, g2, accumulate=False)
    assert ops.last_conv_variant() == "strip_quad", ops.last_conv_variant()
    new = g2.as_nhwc().clone()
    tol = {torch.float16: 2e-3, torch.bfloat16: 1.5e-2}[dtype]
    d2 = new.float().cpu().permute(0, 3, 1, 2)
    assert (d2 - x.grad).abs().max().item() / x.grad.abs().max().item() < tol, "strip_quad (write)"
    ops.conv2d_dgrad_s2(wt.to(dev), gv, g2, accumulate=True)   # a residual: the tile kernels
    assert ops.last_conv_variant() != "strip_quad"
    d3 = g2.as_nhwc().float().cpu().permute(0, 3, 1, 2)
    assert (d3 - 2 * x.grad).abs().max().item() / x.grad.abs().max().item() < 2 * tol, "accumulate"
    tune("conv_strip", 0)
    g3 = ops.View.alloc(n, h, w, cin, dtype, dev)
    g3.buf.fill_(float("nan"))
    ops.conv2d_dgrad_s2(wt.to(dev), gv, g3, accumulate=False)
    assert ops.last_conv_variant() == "v3_quad"
    torch.cuda.synchronize()
    # (the nine (tap, class) products of a pixel are summed shift by shift here, tap by tap there: equal to rounding)
    assert (g3.as_nhwc().float() - new.float()).abs().max().item() <= tol * x.grad.abs().max().item(), "strip_quad and v3_quad differ"


BIG_WGRAD_CASES = [
    ("3x3s1_80", (4, 80, 80, 128, 256, 3, 1)),        # 5 column tiles (4.5 used), slices chosen for one round of 256 blocks
    ("3x3s2_odd", (16, 67, 63, 128, 256, 3, 2)),      # stride 2, odd extents: halo + ragged last K-step
    ("3x3s1_20_deep", (44, 20, 20, 256, 512, 3, 1)),  # small maps: the 32-pixel K-step spans rows and images (cursor wraps)
]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("name,shape", BIG_WGRAD_CASES, ids=[c[0] for c in BIG_WGRAD_CASES])
def test_conv_wgrad_256_tile_vs_autograd(dev, tune, dtype, name, shape):
    """the 8-wave 256x256-tile filter-gradient kernel (Cout % 256 == 0, K >= 1152, >= 16384 pixels: since round 6 the stride-2 layers and
    whatever wgrad_patch.h does not take -- forced here with knob wgrad_patch = 0) against torch autograd in fp32 on the same rounded operands."""
    _lib, ops = _ops()
    tune("wgrad_patch", 0)
    n, h, w, cin, cout, k, s = shape
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, cin, h, w, generator=g).to(dtype).float()
    wt = (torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)).requires_grad_(True)
    y = F.conv2d(x, wt, None, stride=s, padding=k // 2)
    gy = torch.randn(y.shape, generator=g).to(dtype).float()
    y.backward(gy)
    ho, wo = y.shape[2], y.shape[3]
    assert n * ho * wo >= 16384
    xv = ops.View.alloc(n, h, w, cin, dtype, dev)
    ops.nchw_to_nhwc(x.to(dev), xv)
    gv = ops.View.alloc(n, ho, wo, cout, dtype, dev)
    ops.nchw_to_nhwc(gy.to(dev), gv)
    dw, _ = ops.conv2d_wgrad(xv, gv, k, s, cout, cin)
    dw2, _ = ops.conv2d_wgrad(xv, gv, k, s, cout, cin)
    torch.cuda.synchronize()
    assert torch.equal(dw, dw2), "filter gradient is not run-to-run deterministic"
    tol = {torch.float16: 2e-3, torch.bfloat16: 1.5e-2}[dtype]
    e_w = (dw.cpu() - wt.grad).abs().max().item() / wt.grad.abs().max().item()
    assert e_w < tol, f"{name} {dtype}: wgrad {e_w:.2e}"


PATCH_WGRAD_CASES = [
    # name, (n, h, w, cin, cout), channel-slice operands (pitch > C)
    ("80x80_4tiles", (3, 80, 80, 128, 256), False),          # the 128 -> 256 layers: 4 block tiles, slices cross rows and images
    ("40x40_16tiles", (5, 40, 40, 256, 512), False),
    ("20x20_one_tile", (7, 20, 20, 64, 128), False),         # one block tile: every block is a slice of it; ring base wraps many times
    ("odd_37x23", (3, 23, 37, 64, 128), False),              # odd extents: pad columns / rows at every phase of the 64-position stage
    ("narrow_w5_tall", (2, 50, 5, 64, 128), False),          # W + 2 = 7: a stage spans 9 padded rows; back = 16
    ("wide_w141", (1, 9, 141, 64, 128), False),              # the widest map the ring holds (back = 144, 352 mirrored rows)
    ("one_row", (4, 1, 64, 64, 256), False),                 # H = 1: every position's vertical taps are pad rows
    ("sliced_operands", (2, 26, 30, 128, 128), True),        # x and du are channel slices of wider buffers (Concat inputs, pitch > C)
    ("tiny_one_stage", (1, 2, 3, 64, 128), False),           # 15 padded positions: one stage, one slice per tile
]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("name,shape,sliced", PATCH_WGRAD_CASES, ids=[c[0] for c in PATCH_WGRAD_CASES])
def test_conv_wgrad_patch_vs_autograd(dev, tune, dtype, name, shape, sliced):
    """the padded-position filter-gradient kernel (csrc/wgrad_patch.h: 3x3 / stride 1, Cin % 64 == 0, Cout % 128 == 0; x staged once per position in a mirrored
    ring, the nine taps as row offsets) on every eligible shape (knob wgrad_patch = 2) against torch autograd in fp32 on the same rounded operands, against the
    tile kernels it replaces (knob 0), and run to run (bit-identical: fixed slice order, no atomics)."""
    _lib, ops = _ops()
    n, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(13)
    x = torch.randn(n, cin, h, w, generator=g).to(dtype).float()
    wt = torch.zeros(cout, cin, 3, 3, requires_grad=True)
    y = F.conv2d(x, wt, None, stride=1, padding=1)
    gy = torch.randn(y.shape, generator=g).to(dtype).float()
    y.backward(gy)
    if sliced:
        xw = ops.View.alloc(n, h, w, cin + 64, dtype, dev)
        xw.buf.fill_(float("nan"))
        gw = ops.View.alloc(n, h, w, cout + 32, dtype, dev)
        gw.buf.fill_(float("nan"))
        xv, gv = xw.slice(32, cin), gw.slice(16, cout)
    else:
        xv, gv = ops.View.alloc(n, h, w, cin, dtype, dev), ops.View.alloc(n, h, w, cout, dtype, dev)
    ops.nchw_to_nhwc(x.to(dev), xv)
    ops.nchw_to_nhwc(gy.to(dev), gv)
    tune("wgrad_patch", 2)
    tile, slices, _ = ops.conv2d_wgrad_plan(xv, cout, 3, 1)
    assert tile == 4 and slices >= 1, (tile, slices)
    dw, _ = ops.conv2d_wgrad(xv, gv, 3, 1, cout, cin)
    dw2, _ = ops.conv2d_wgrad(xv, gv, 3, 1, cout, cin)
    tune("wgrad_patch", 0)
    assert ops.conv2d_wgrad_plan(xv, cout, 3, 1)[0] in (128, 256)
    dw_old, _ = ops.conv2d_wgrad(xv, gv, 3, 1, cout, cin)
    torch.cuda.synchronize()
    assert torch.equal(dw, dw2), "not run-to-run deterministic"
    ref = wt.grad
    scale = ref.abs().max().item()
    e_w = (dw.cpu() - ref).abs().max().item() / scale
    e_o = (dw.cpu() - dw_old.cpu()).abs().max().item() / scale
    print(f"[wgrad_patch {name} {dtype}] slices {slices}: vs autograd {e_w:.2e}, vs tile kernel {e_o:.2e}")
    # same products, fp32 accumulation in another order
    assert e_w < 2e-5 and e_o < 2e-5, f"{name} {dtype}: vs autograd {e_w:.2e}, vs tile kernel {e_o:.2e}"


def test_fused_sgd_vs_torch_reference(dev):
    """unscale + clip_grad_norm_(10) + SGD(nesterov, 3 groups) + EMA in the fused kernel vs torch's reference ops (fp32)."""
    from yolov3_amd.optim import FusedSGD, ModelEMA

    torch.manual_seed(0)
    shapes = [(64, 32, 3, 3), (64,), (255, 128, 1, 1), (255,), (1000003,)]
    ps = [torch.nn.Parameter(torch.randn(s, device=dev)) for s in shapes]
    ref = [torch.nn.Parameter(p.detach().cpu().clone()) for p in ps]
    groups = lambda q: [{"params": [q[1], q[3]], "weight_decay": 0.0}, {"params": [q[0], q[2], q[4]], "weight_decay": 5e-4}]
    opt = FusedSGD(groups(ps), lr=0.01, momentum=0.937, nesterov=True)
    topt = torch.optim.SGD(groups(ref), lr=0.01, momentum=0.937, nesterov=True)

    class Holder(torch.nn.Module):
        def __init__(self, q):
            super().__init__()
            self.q = torch.nn.ParameterList(q)

    ema = ModelEMA(Holder(ps))
    ema_ref = [p.detach().clone() for p in ref]
    upd = 0
    for step in range(3):
        scale = 1024.0
        for p, r in zip(ps, ref):
            g = torch.randn(r.shape) * (30.0 if step == 1 else 1.0)  # step 1 exceeds max_norm
            r.grad = g.clone()
            p.grad = (g * scale).to(dev)
        norm_ref = torch.nn.utils.clip_grad_norm_(ref, max_norm=10.0)
        topt.step()
        upd += 1
        d = 0.9999 * (1 - math.exp(-upd / 2000))
        for e, r in zip(ema_ref, ref):
            e.mul_(d).add_(r.detach(), alpha=1 - d)
        opt.step(grad_scale=scale, max_norm=10.0, ema=ema)
        torch.cuda.synchronize()
        assert abs(opt.last_norm.item() - norm_ref.item()) / norm_ref.item() < 1e-5
        for p, r in zip(ps, ref):
            torch.testing.assert_close(p.detach().cpu(), r.detach(), rtol=1e-5, atol=1e-6)
        for p, e in zip(ps, ema_ref):
            torch.testing.assert_close(ema.shadow[p].cpu(), e, rtol=1e-5, atol=1e-6)
    # inf gradient -> step skipped
    before = [p.detach().clone() for p in ps]
    for p in ps:
        p.grad = torch.full_like(p, float("inf"))
    opt.step(grad_scale=1.0, max_norm=10.0)
    torch.cuda.synchronize()
    assert opt.found_inf.item() == 1 and all(torch.equal(a, b.detach()) for a, b in zip(before, ps))


def test_grad_scaler_dynamic_scale_matches_torch(dev):
    """optim.GradScaler (device-resident scale, growth counter and found-inf flag; reference train.py:345,411-418) against
    torch.amp.GradScaler driving torch.optim.SGD on the same gradient sequence: clean steps, an overflow step (skipped, scale
    halves), growth after `growth_interval` clean steps -- parameters and the scale agree after every step, with no host sync in
    our loop until the final compare."""
    from yolov3_amd.optim import FusedSGD, GradScaler

    torch.manual_seed(1)
    shapes = [(32, 16, 3, 3), (32,), (70001,)]
    ps = [torch.nn.Parameter(torch.randn(s, device=dev)) for s in shapes]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt = FusedSGD(ps, lr=0.05, momentum=0.9, nesterov=True)
    topt = torch.optim.SGD(ref, lr=0.05, momentum=0.9, nesterov=True)
    kw = dict(init_scale=2.0**10, growth_factor=2.0, backoff_factor=0.5, growth_interval=2)
    ours, theirs = GradScaler(**kw), torch.amp.GradScaler("cuda", **kw)
    scales = []
    for step in range(7):
        overflow = step in (2, 5)
        theirs.scale(torch.ones(1, device=dev))   # torch initialises its scale tensor lazily in scale()
        g = [torch.randn(s, device=dev) for s in shapes]
        for p, r, gi in zip(ps, ref, g):
            p.grad = gi * ours._scale if ours._scale is not None else gi * kw["init_scale"]
            r.grad = gi * theirs.get_scale()
            if overflow:
                p.grad.view(-1)[3] = float("inf")
                r.grad.view(-1)[3] = float("inf")
        if ours._scale is None:
            ours._lazy(dev)
        ours.unscale_(opt)
        ours.step(opt, max_norm=0.0)
        ours.update()
        theirs.step(topt)
        theirs.update()
        scales.append((ours._scale.clone(), theirs.get_scale()))
    torch.cuda.synchronize()
    for (a, b) in scales:
        assert a.item() == b, (a.item(), b)
    assert [b for _, b in scales] == [1024.0, 2048.0, 1024.0, 1024.0, 2048.0, 1024.0, 1024.0]
    for p, r in zip(ps, ref):
        torch.testing.assert_close(p.detach(), r.detach(), rtol=1e-5, atol=1e-6)
    x = torch.ones(3, device=dev, requires_grad=True)
    assert torch.equal(ours.scale(x.sum()), x.sum() * ours._scale)
    sd = ours.state_dict()
    assert sd["scale"] == 1024.0 and sd["growth_interval"] == 2


@pytest.mark.parametrize("name,h,w,bs,dtype", [("yolov3", 96, 160, 1, torch.float32), ("yolov3-tiny", 128, 96, 3, torch.float32), ("yolov3-spp", 160, 96, 2, torch.float16)])
def test_model_rectangular_and_odd_batches(dev, name, h, w, bs, dtype):
    """rect inference (val.py pads batches to rectangles, e.g. 640x512), batch sizes that do not fill a pixel tile,
    export mode and plan re-use across shapes."""
    m, (layers, save, sd, strides) = build_pair(name, 80, 29, dev, dtype)
    x = torch.rand(bs, 3, h, w, generator=torch.Generator().manual_seed(4))
    pred, raw = m(x.to(dev).to(dtype))
    torch.cuda.synchronize()
    with torch.no_grad():
        refp, refraw = yo.forward(layers, save, sd, x, strides, training=False)
    assert pred.shape == refp.shape
    for a, b in zip(raw, refraw):
        a = a.float().cpu()
        assert a.shape == b.shape
        if dtype == torch.float32:
            assert (a - b).abs().max().item() < 1e-4
        else:
            assert (a - b).abs().max().item() / b.abs().max().item() < 0.03
    # a second shape compiles a second plan; the first one is still valid afterwards
    x2 = torch.rand(bs + 1, 3, h + 32, w, generator=torch.Generator().manual_seed(5))
    p2, _ = m(x2.to(dev).to(dtype))
    p1, _ = m(x.to(dev).to(dtype))
    torch.cuda.synchronize()
    assert torch.equal(p1, pred) and p2.shape[0] == bs + 1
    m.model[-1].export = True
    out = m(x.to(dev).to(dtype))
    assert isinstance(out, tuple) and len(out) == 1 and torch.equal(out[0], pred)
    m.model[-1].export = False


# ------------------------------------------------------------------------------------------------ output edge after NMS
def test_process_batch_vs_reference_golden(dev, golden_dir):
    """y3_match_detections behind the reference signature process_batch(detections, labels, iouv): bit-exact boolean
    matrices against the unmodified reference (val.py:147-188) on every seeded case, one image at a time and all images in
    one batched launch (ragged counts, per-image label ranges)."""
    from yolov3_amd import process_batch, process_batch_batched

    gold = torch.load(golden_dir / "val_edge.pt")["match"]
    iouv = torch.linspace(0.5, 0.95, 10)
    cases = [(name, *yo.synth_val_case(**rec["gen"]), rec["correct"]) for name, rec in gold.items()]
    for name, det, lab, want in cases:
        got = process_batch(det.to(dev), lab.to(dev), iouv.to(dev))
        assert got.dtype == torch.bool and got.shape == want.shape and torch.equal(got.cpu(), want), name
    max_det = max(c[1].shape[0] for c in cases)
    rows = torch.full((len(cases), max_det, 6), float("nan"))
    counts, offs, labs = [], [0], []
    for i, (_, det, lab, _) in enumerate(cases):
        rows[i, : det.shape[0]] = det
        counts.append(det.shape[0])
        labs.append(lab)
        offs.append(offs[-1] + lab.shape[0])
    correct = process_batch_batched(rows.to(dev), torch.tensor(counts, dtype=torch.int32, device=dev), torch.cat(labs).to(dev), torch.tensor(offs, dtype=torch.int32), iouv)
    torch.cuda.synchronize()
    for i, (name, det, _, want) in enumerate(cases):
        assert torch.equal(correct[i, : det.shape[0]].bool().cpu(), want), f"batched {name}"
        assert int(correct[i, det.shape[0] :].sum()) == 0, f"batched {name}: rows beyond the count must be 0"


def test_scale_boxes_vs_reference_golden(dev, golden_dir):
    """y3_scale_boxes behind scale_boxes(img1_shape, boxes, img0_shape, ratio_pad): bit-exact fp32 against the reference
    (utils/general.py:613-626) on a column view of (n, 6) rows (the call shape of val.py:397), confidences/classes untouched;
    and the batched form over images with different native shapes in one launch."""
    from yolov3_amd import scale_boxes, scale_boxes_batched

    gold = torch.load(golden_dir / "val_edge.pt")["scale"]
    for name, rec in gold.items():
        rows = yo.synth_scale_case(rec["img1"]).to(dev)
        keep = rows.clone()
        out = scale_boxes(rec["img1"], rows[:, :4], rec["img0"], rec["ratio_pad"])
        torch.cuda.synchronize()
        assert torch.equal(rows[:, :4].cpu(), rec["out"]) and torch.equal(out.cpu(), rec["out"]), name
        assert torch.equal(rows[:, 4:], keep[:, 4:]), name
    same = [(n, r) for n, r in gold.items() if tuple(r["img1"]) == (640, 640)]
    rows = torch.stack([yo.synth_scale_case(r["img1"]) for _, r in same]).to(dev)
    counts = torch.tensor([400, 123, 0][: len(same)], dtype=torch.int32, device=dev)
    before = rows.clone()
    scale_boxes_batched((640, 640), rows, counts, [r["img0"] for _, r in same], [r["ratio_pad"] for _, r in same])
    torch.cuda.synchronize()
    for i, (name, r) in enumerate(same):
        c = int(counts[i])
        assert torch.equal(rows[i, :c, :4].cpu(), r["out"][:c]), f"batched {name}"
        assert torch.equal(rows[i, c:], before[i, c:]) and torch.equal(rows[i, :, 4:], before[i, :, 4:]), f"batched {name}: untouched parts"


def test_val_edge_pipeline_full_size_vs_oracle(dev):
    """NMS (val settings) -> scale_boxes -> process_batch on a full-size batch (8 x 25200 x 85 fp16), everything batched on the
    device, against the oracle run image by image: identical correct-matrices."""
    from yolov3_amd import non_max_suppression_batched, process_batch_batched, scale_boxes_batched

    bs = 8
    pred = yo.synth_predictions(bs=bs, n_rows=25200, nc=80, seed=41, dtype=torch.float16)
    shapes = [(480, 640), (640, 480), (720, 1280), (333, 500), (640, 640), (500, 375), (1080, 1920), (427, 640)]
    iouv = torch.linspace(0.5, 0.95, 10)
    ref = yo.non_max_suppression(pred, 0.001, 0.6, multi_label=True, max_det=300)
    want_rows = []
    labels, offs = [], [0]
    g = torch.Generator().manual_seed(9)
    for i in range(bs):
        w = ref[i].clone()
        yo.scale_boxes((640, 640), w[:, :4], shapes[i])
        want_rows.append(w)
        # ground truth = every 7th detection, jittered (IoU spread over the thresholds), so that the matrices are not all zero
        pick = w[::7]
        lab = torch.cat((pick[:, 5:6], pick[:, :4] * (0.9 + 0.2 * torch.rand(pick.shape[0], 4, generator=g))), 1)
        labels.append(lab)
        offs.append(offs[-1] + lab.shape[0])
    rows, counts_t, counts = non_max_suppression_batched(pred.to(dev), 0.001, 0.6, multi_label=True, max_det=300)
    scale_boxes_batched((640, 640), rows, counts_t, shapes)
    correct = process_batch_batched(rows, counts_t, torch.cat(labels), torch.tensor(offs, dtype=torch.int32), iouv)
    torch.cuda.synchronize()
    hits = 0
    for i in range(bs):
        assert counts[i] == want_rows[i].shape[0] and torch.equal(rows[i, : counts[i]].cpu(), want_rows[i]), f"image {i}: scaled rows"
        want = yo.process_batch(want_rows[i], labels[i], iouv) if want_rows[i].shape[0] else torch.zeros(0, 10, dtype=torch.bool)
        assert torch.equal(correct[i, : counts[i]].bool().cpu(), want), f"image {i}: correct matrix"
        hits += int(want.sum())
    assert hits > 50, "the test input should produce matches"


# ------------------------------------------------------------------------------------------------ input edge: letterbox / AutoShape
LETTERBOX_CASES = [
    ("pad_only", (480, 640), (640, 640)),        # r = 1: no resize, 80-row borders
    ("downscale_hd", (720, 1280), (640, 640)),
    ("upscale_odd", (333, 500), (640, 640)),
    ("portrait_rect", (1080, 810), (640, 480)),
    ("tiny_x12", (37, 53), (448, 640)),
    ("half_pixel_pad", (375, 500), (512, 672)),  # odd padding: top/bottom and left/right differ by one (round(d -/+ 0.1))
]


@pytest.mark.parametrize("name,shape0,shape1", LETTERBOX_CASES, ids=[c[0] for c in LETTERBOX_CASES])
def test_letterbox_u8_vs_oracle(dev, name, shape0, shape1):
    """y3_letterbox_u8 (cv2.resize INTER_LINEAR + copyMakeBorder(114) + HWC->CHW in one pass) bit-exact against the oracle's
    restatement of reference utils/augmentations.py:104-134 (cv2's 8-bit fixed-point resize restated; cv2 parity unpinned)."""
    from yolov3_amd import letterbox_batch

    im = yo.synth_image_u8(shape0[0], shape0[1], seed=3)
    want, _, _ = yo.letterbox(im, shape1, auto=False)
    assert want.shape[:2] == tuple(shape1)
    got = letterbox_batch([im, im[:, ::-1].copy()], shape1, dev)
    torch.cuda.synchronize()
    assert got.dtype == torch.uint8 and tuple(got.shape) == (2, 3, *shape1)
    assert torch.equal(got[0].cpu(), torch.from_numpy(want).permute(2, 0, 1)), name
    want2, _, _ = yo.letterbox(im[:, ::-1].copy(), shape1, auto=False)
    assert torch.equal(got[1].cpu(), torch.from_numpy(want2).permute(2, 0, 1)), name + " (second image of the batch)"


def test_autoshape_numpy_images_vs_oracle_pipeline(dev):
    """AutoShape on raw numpy images of different sizes (reference models/common.py:819-874): device letterbox == oracle
    letterbox (bit-exact), uint8 ingest with the /255 inside the first kernel == oracle forward on x/255 (fp32, 1e-4 on
    logits-derived boxes), and the returned per-image detections == oracle NMS + oracle scale_boxes on the same predictions
    (row-exact)."""
    import numpy as np

    from yolov3_amd import AutoShape, letterbox_batch

    m, (layers, save, sd, strides) = build_pair("yolov3-tiny", 80, 23, dev, torch.float32)
    ims = [yo.synth_image_u8(240, 320, seed=1), yo.synth_image_u8(300, 200, seed=2), yo.synth_image_u8(96, 128, seed=3)]
    size = 320
    a = AutoShape(m)
    a.conf, a.iou, a.multi_label, a.max_det = 0.0, 0.45, False, 300   # conf 0: a random-weight model's objectness is ~0.003 everywhere
    det = a([im.copy() for im in ims], size=size)
    torch.cuda.synchronize()
    # the reference's shape logic (models/common.py:861-865)
    shape1 = [[int(y * size / max(im.shape[:2])) for y in im.shape[:2]] for im in ims]
    shape1 = [int(np.ceil(v / 32) * 32) for v in np.array(shape1).max(0)]
    assert det.s == (3, 3, *shape1) and det.n == 3
    x_ref = np.stack([yo.letterbox(im, shape1, auto=False)[0] for im in ims]).transpose(0, 3, 1, 2)
    x_dev = letterbox_batch(ims, shape1, dev)
    assert torch.equal(x_dev.cpu(), torch.from_numpy(np.ascontiguousarray(x_ref)))
    m.model[-1].export = False
    pred, _ = m(x_dev)   # uint8 in: divided by 255 inside the ingest
    with torch.no_grad():
        pred_ref, _ = yo.forward(layers, save, sd, torch.from_numpy(np.ascontiguousarray(x_ref)).float() / 255, strides, training=False)
    err = (pred.cpu() - pred_ref).abs().max().item() / pred_ref.abs().max().item()
    assert err < 1e-4, f"uint8-ingest forward differs from the oracle by {err:.2e} (relative to max)"
    ref = yo.non_max_suppression(pred.cpu(), 0.0, 0.45, multi_label=False, max_det=300)
    for i, r in enumerate(ref):
        yo.scale_boxes(shape1, r[:, :4], ims[i].shape[:2])
        assert det.pred[i].shape == r.shape and torch.equal(det.pred[i].cpu(), r), f"image {i}: detections differ from oracle NMS + scale_boxes"
    assert sum(p.shape[0] for p in det.pred) > 0
    assert torch.allclose(det.xywhn[0][:, :4].cpu(), (torch.cat(((ref[0][:, :2] + ref[0][:, 2:4]) / 2, ref[0][:, 2:4] - ref[0][:, :2]), 1) / torch.tensor([320.0, 240.0, 320.0, 240.0])), atol=1e-6)


def test_config5_1280_objects365_bf16_vs_fp32_engine(dev):
    """BASELINE configs[4] geometry on one GPU: yolov3, 1280x1280, Objects365 head (nc 365 -> 370 outputs per anchor, head convs
    1110 channels, 100800 rows), bf16 MFMA engine against the fp32 direct-kernel engine (itself pinned to the reference
    goldens at 1e-4) on the same weights: the large-activation regime (L0 output 105 MB per image) through every kernel
    family, then batched NMS on the (1, 100800, 370) prediction tensor against the oracle (row-exact)."""
    from yolov3_amd import non_max_suppression

    m32, _ = build_pair("yolov3", 365, 29, dev, torch.float32)
    x = torch.rand(1, 3, 1280, 1280, generator=torch.Generator().manual_seed(12))
    with torch.no_grad():
        p32, raw32 = m32(x.to(dev))
        mb = m32.to(torch.bfloat16)
        pb, rawb = mb(x.to(dev).to(torch.bfloat16))
    torch.cuda.synchronize()
    assert tuple(pb.shape) == (1, 100800, 370) and pb.dtype == torch.bfloat16
    bnd = HALF_BOUNDS[torch.bfloat16]   # the bounds of the oracle comparison at this resolution (test_model_half_vs_fp32_oracle_benchmark_shapes), not a looser one
    for lvl, (a, b) in enumerate(zip(rawb, raw32)):
        rms, mx, corr = _rel_errors(a.float().cpu(), b.cpu())
        assert rms < bnd["rms"] and mx < bnd["mx"] and corr > bnd["corr"], f"bf16 vs fp32 engine, level {lvl}: rel RMS {rms:.4g} max/range {mx:.4g} corr {corr:.6f}"
    got = non_max_suppression(pb, 0.001, 0.6, multi_label=True, max_det=300)
    want = yo.non_max_suppression(pb.cpu(), 0.001, 0.6, multi_label=True, max_det=300)
    _cmp_nms(got, want, "1280/365")


@pytest.mark.parametrize("n,h,w,c", [(3, 20, 20, 512), (2, 40, 40, 512), (1, 15, 23, 40), (1, 80, 80, 64)], ids=["640_spp", "1280_spp", "odd_rect_cg1", "direct_fallback"])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_spp_pyramid_shapes(dev, dtype, n, h, w, c):
    """SPP 5/9/13 pyramid (reference models/common.py:287-290) on the shapes of yolov3-spp at 640 / 1280 and on maps that force
    the narrow-group and the direct (no LDS) variants: bit-exact against three nn.MaxPool2d(k, 1, k // 2)."""
    _lib, ops = _ops()
    x = torch.randn(n, c, h, w, generator=torch.Generator().manual_seed(2)).to(dtype)
    cat = ops.View.alloc(n, h, w, 4 * c, dtype, dev)
    cat.buf.zero_()
    ops.nchw_to_nhwc(x.to(dev), cat.slice(0, c))
    ops.spp_pyramid(cat.slice(0, c), cat.slice(c, 3 * c))
    ref = torch.cat([x.float()] + [F.max_pool2d(x.float(), k, 1, k // 2) for k in (5, 9, 13)], 1).to(dtype)
    assert torch.equal(ops.nhwc_to_nchw(cat).cpu(), ref)


STAT_CASES = [
    ("v3_bk64", (2, 40, 40, 128, 256, 3, 1)),
    ("v6_ragged_last_tile", (4, 20, 20, 512, 512, 3, 1)),
    ("cout64_odd_pixels", (2, 33, 17, 32, 64, 3, 1)),
    ("1x1", (2, 40, 40, 256, 128, 1, 1)),
    ("stride2_odd", (2, 41, 37, 64, 128, 3, 2)),
    ("v2_small_cin", (1, 20, 20, 16, 32, 3, 1)),
    ("cout_not_tile_multiple", (2, 24, 24, 64, 200, 3, 1)),
    ("many_rows_two_level_sum", (8, 96, 96, 32, 64, 3, 1)),   # 576 rows > 512: first-level sums into the fp64 partial rows
    ("s1x1_k16", (3, 37, 29, 256, 128, 1, 1)),                 # conv_1x1s.h (forced: knob conv_1x1s = 2): one row per (stage, pixel wave, epilogue pass), ragged last stage
    ("s1x1_two_filter_groups", (2, 30, 30, 128, 256, 1, 1)),
    ("s1x1_two_pixel_waves", (2, 33, 31, 128, 64, 1, 1)),
    ("s1x1_four_passes", (1, 50, 50, 64, 128, 1, 1)),
    ("s1x1_two_filter_tiles", (2, 33, 31, 256, 512, 1, 1)),     # 256 filters per block, the two blocks of a pixel range write the two halves of a row
]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("name,shape", STAT_CASES, ids=[c[0] for c in STAT_CASES])
def test_conv_epilogue_bn_statistics(dev, tune, dtype, name, shape):
    """y3_conv2d_fwd_stats: same output as y3_conv2d_fwd (bit-exact) and per-(tile, wave) rows whose fp64 sum equals the
    per-channel (sum, sum of squares) of the STORED output (1e-5 relative: fp32 partial sums over <= 128 pixels), for every
    tile variant incl. ragged pixel / filter tiles; then y3_bn_finalize_rows against nn.BatchNorm2d's batch statistics."""
    _lib, ops = _ops()
    n, h, w, cin, cout, k, s = shape
    if name.startswith("s1x1"):
        tune("conv_1x1s", 2)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(n, cin, h, w, generator=g).to(dtype)
    wt = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
    xv = ops.View.alloc(n, h, w, cin, dtype, dev)
    ops.nchw_to_nhwc(x.to(dev), xv)
    ho, wo = (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1
    filt = ops.pack_filter(wt.to(dev), cout, cin, dtype)
    zb = torch.zeros(cout, device=dev)
    y0 = ops.View.alloc(n, ho, wo, cout, dtype, dev)
    ops.conv2d(xv, filt, zb, y0, k, s, act=False)
    assert (ops.last_conv_variant() == "s1x1") == name.startswith("s1x1"), ops.last_conv_variant()
    y1 = ops.View.alloc(n, ho, wo, cout, dtype, dev)
    rows = ops.conv2d_stats_rows(xv, y1, k, s)
    buf = torch.full((rows * 2 * cout,), float("nan"), device=dev)
    got_rows = ops.conv2d_stats(xv, filt, zb, y1, k, s, buf, rows)
    torch.cuda.synchronize()
    assert got_rows == rows and torch.equal(y0.buf, y1.buf)
    u = y1.as_nhwc().double().cpu().reshape(-1, cout)
    tot = buf.view(rows, cout, 2).double().sum(0).cpu()
    assert torch.isfinite(tot).all(), "a statistics row was not written"
    ref0, ref1 = u.sum(0), (u * u).sum(0)
    assert (tot[:, 0] - ref0).abs().max().item() <= 1e-5 * u.abs().sum(0).max().item(), "sum"
    assert (tot[:, 1] - ref1).abs().max().item() <= 1e-5 * ref1.max().item(), "sum of squares"
    # finalize from the rows == BatchNorm2d(eps 1e-3, momentum 0.03) batch statistics of the stored tensor
    sums = ops.bn_scratch(cout, dev)
    gamma, beta = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
    rm, rv = torch.zeros(cout, device=dev), torch.ones(cout, device=dev)
    scale, shift, mean, invstd = (torch.empty(cout, device=dev) for _ in range(4))
    cnt = n * ho * wo
    _lib.check(_lib.lib().y3_bn_finalize_rows(buf.data_ptr(), rows, cnt, cout, sums.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-3, 0.03, rm.data_ptr(), rv.data_ptr(),
                                              scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(), ops.stream_ptr()), "y3_bn_finalize_rows")
    torch.cuda.synchronize()
    mu, var = u.mean(0), u.var(0, unbiased=False)
    torch.testing.assert_close(mean.double().cpu(), mu, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(invstd.double().cpu(), 1.0 / torch.sqrt(var + 1e-3), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(rm.double().cpu(), 0.03 * mu, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(rv.double().cpu(), 0.97 + 0.03 * u.var(0, unbiased=True), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(scale.double().cpu(), gamma.double().cpu() / torch.sqrt(var + 1e-3), rtol=1e-4, atol=1e-5)


def test_detect_batches_overlapped_equals_sequential(dev):
    """val.detect_batches (NMS of batch i on a second stream beside the forward of batch i+1) returns, batch by batch, exactly
    the detections of the sequential model -> non_max_suppression pair."""
    from yolov3_amd import detect_batches, non_max_suppression

    m, _ = build_pair("yolov3-tiny", 80, 23, dev, torch.float16)
    g = torch.Generator().manual_seed(3)
    batches = [torch.rand(4, 3, 160, 160, generator=g).to(dev).half() for _ in range(5)]
    kw = dict(conf_thres=0.0, iou_thres=0.45, max_det=100)
    want = [non_max_suppression(m(x)[0], **kw) for x in batches]
    got = list(detect_batches(m, batches, **kw))
    torch.cuda.synchronize()
    assert len(got) == len(want)
    for b, (a_list, w_list) in enumerate(zip(got, want)):
        _cmp_nms(a_list, [w.cpu() for w in w_list], f"batch {b}")
    assert sum(d.shape[0] for dets in got for d in dets) > 0


PAIR_CASES = [
    ("many_tiles_per_block", (13, 320, 352), torch.float16, 1.0),   # 13 * 40 * 6 = 3120 tiles > 512 persistent blocks: the grid-stride loop and its prefetch
    ("square", (2, 64, 64), torch.float16, 1.0),
    ("odd_rect", (1, 37, 53), torch.float16, 1.0),
    ("wide_multi_tile", (3, 96, 160), torch.float16, 1.0),
    ("smaller_than_a_tile", (1, 5, 7), torch.float16, 1.0),
    ("u8_div255", (2, 48, 80), torch.uint8, 255.0),
    ("fp32_source", (1, 40, 72), torch.float32, 1.0),
]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("name,shape,sdt,div", PAIR_CASES, ids=[c[0] for c in PAIR_CASES])
def test_stem_pair_vs_fp32_reference(dev, dtype, name, shape, sdt, div):
    """y3_stem_pair_fwd (layers 0 + 1 of yolov3 in one kernel, layer 0 kept in LDS) against torch fp32 convolutions on the same
    rounded operands, layer 0's output rounded to the storage dtype in between as the unfused path stores it; borders (layer 1's
    zero padding applies to layer 0's OUTPUT), odd sizes, images smaller than a tile, uint8 / fp32 sources, more tiles than
    persistent blocks."""
    _lib, ops = _ops()
    n, h, w = shape
    g = torch.Generator().manual_seed(11)
    if sdt == torch.uint8:
        x = torch.randint(0, 256, (n, 3, h, w), generator=g, dtype=torch.uint8)
        xr = (x.to(dtype) / 255).float()
    else:
        x = torch.rand(n, 3, h, w, generator=g).to(sdt)
        xr = x.to(dtype).float()
    w0 = (torch.randn(32, 3, 3, 3, generator=g) / math.sqrt(27)).to(dtype).float()
    b0 = torch.randn(32, generator=g) * 0.1
    w1 = (torch.randn(64, 32, 3, 3, generator=g) / math.sqrt(288)).to(dtype).float()
    b1 = torch.randn(64, generator=g) * 0.1
    y0 = F.silu(F.conv2d(xr, w0, b0, stride=1, padding=1)).to(dtype).float()
    ref = F.silu(F.conv2d(y0, w1, b1, stride=2, padding=1))
    ho, wo = ref.shape[2], ref.shape[3]
    f0 = ops.pack_filter_stem(w0.to(dev), 32, dtype)
    f1 = ops.pack_filter(w1.to(dev), 64, 32, dtype)
    yv = ops.View.alloc(n, ho, wo, 64, dtype, dev)
    yv.buf.fill_(float("nan"))
    ops.stem_pair(x.to(dev), f0, b0.to(dev), True, f1, b1.to(dev), True, yv, div)
    torch.cuda.synchronize()
    got = yv.as_nhwc().float().cpu().permute(0, 3, 1, 2)
    assert torch.isfinite(got).all(), "an output pixel was not written"
    tol = 2.0**-8 if dtype == torch.float16 else 2.0**-5
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    assert err < tol, f"{name} {dtype}: {err:.2e}"


def test_stem_pair_matches_unfused_model(dev, monkeypatch):
    """yolov3 forward with layers 0 + 1 fused against the same model with Y3_STEM_PAIR=0 (stem kernel + generic layer 1):
    identical raw head tensors up to fp32 summation order (2^-9 of the logit range)."""
    x = torch.rand(2, 3, 96, 128, generator=torch.Generator().manual_seed(4))
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("Y3_STEM_PAIR", flag)
        m, _ = build_pair("yolov3", 80, 21, dev, torch.float16)
        pred, raw = m(x.to(dev).half())
        kinds = [ln.kernel for ln in next(iter(m._plans.values())).launches]
        assert ("stem_pair" in kinds) == (flag == "1")
        outs.append([r.float().cpu() for r in raw])
    for a, b in zip(*outs):
        assert (a - b).abs().max().item() <= 2.0**-9 * b.abs().max().item()


def test_bneck_pair_matches_unfused_model(dev, monkeypatch):
    """yolov3 forward with layer 2 (Bottleneck(64, 64)) as one kernel against the same model with Y3_BNECK_PAIR=0 (two generic
    launches): the plan uses the kernel, and the raw head tensors agree up to fp32 summation order."""
    x = torch.rand(2, 3, 96, 128, generator=torch.Generator().manual_seed(4))
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("Y3_BNECK_PAIR", flag)
        m, _ = build_pair("yolov3", 80, 21, dev, torch.float16)
        pred, raw = m(x.to(dev).half())
        kinds = [ln.kernel for ln in next(iter(m._plans.values())).launches]
        assert ("bneck_pair" in kinds) == (flag == "1")
        outs.append([r.float().cpu() for r in raw])
    for a, b in zip(*outs):
        assert (a - b).abs().max().item() <= 2.0**-9 * b.abs().max().item()


# ------------------------------------------------------------------------------------------------ round 2: plans, scaling traps, exchange step, checkpoints
def test_plans_survive_deepcopy_and_checkpoint_roundtrip(dev, tmp_path):
    """After a real forward (plans + packed filters exist) the reference's checkpoint path must work: deepcopy(model) (ModelEMA),
    torch.save({'model': deepcopy(model).half()}) (train.py:470-488) and loading it back through compat.attempt_load /
    DetectMultiBackend; the reloaded fp16 model reproduces the original's fp16 output bit for bit."""
    import copy

    from yolov3_amd import DetectMultiBackend

    m, _ = build_pair("yolov3-tiny", 80, 29, dev, torch.float32)
    x = torch.rand(2, 3, 96, 128, generator=torch.Generator().manual_seed(1)).to(dev)
    with torch.no_grad():
        p32 = m(x)[0]
    assert len(m._plans) == 1
    ema = copy.deepcopy(m)                      # used to raise: cannot pickle 'CArgObject'
    assert len(ema._plans) == 0
    with torch.no_grad():
        assert torch.equal(ema(x)[0], p32)
    ck = tmp_path / "last.pt"
    torch.save({"epoch": 0, "model": copy.deepcopy(m).half(), "ema": None, "optimizer": None}, ck)
    back = DetectMultiBackend(str(ck), device=dev, fp16=True, fuse=False)
    mh = copy.deepcopy(m).half()
    with torch.no_grad():
        a = back(x)[0]
        b = mh(x.half())[0]
    assert a.dtype == torch.float16 and torch.equal(a, b)


def test_reference_checkpoint_fixture_runs_on_gpu(dev, golden_dir):
    """a .pt pickled by the UNMODIFIED reference (tests/golden/ref_tiny_w025_fp16.pt, written by make_golden.py as
    train.py:470-488 does) through DetectMultiBackend(weights, fp16=...) (models/common.py:471-476 -> attempt_load -> fuse -> eval):
    fp32 within 1e-4 of the reference's own eval output of that checkpoint, fp16 within the half-precision bounds."""
    from yolov3_amd import DetectMultiBackend

    gold = torch.load(golden_dir / "ref_tiny_w025_eval.pt")
    x = torch.rand(2, 3, 96, 160, generator=torch.Generator().manual_seed(9))
    assert checksum(x) == gold["x_sum"]
    from yolov3_amd import compat

    m32 = DetectMultiBackend(str(golden_dir / "ref_tiny_w025_fp16.pt"), device=dev, fp16=False)
    assert not any(".bn." in k for k in m32.model.state_dict()) and m32.stride == 32 and m32.names[3] == "c3"
    y = m32(x.to(dev))
    assert isinstance(y, list)
    torch.testing.assert_close(y[0].cpu(), gold["pred"], rtol=1e-4, atol=2e-4)
    m16 = DetectMultiBackend(str(golden_dir / "ref_tiny_w025_fp16.pt"), device=dev, fp16=True)
    y16 = m16(x.to(dev))   # fp32 images are cast to half like the reference's forward does (models/common.py:650-651)
    assert y16[0].dtype == torch.float16
    compat.uninstall_aliases()
    for lvl, (a, b) in enumerate(zip(y16[1], gold["raw"])):
        rms, mx, corr = _rel_errors(a.float().cpu(), b)
        assert rms < HALF_BOUNDS[torch.float16]["rms"] and corr > 0.9999, (lvl, rms, mx, corr)


def test_rect_batches_share_packed_filters_and_plans_are_bounded(dev):
    """val.py's rect batches give dozens of (h, w) shapes (utils/dataloaders.py:548-570).  Every shape gets a plan (activations,
    workspace) but the packed filter banks are shared: filter memory stays constant after the first shape and at most
    PlanCache.MAX_EVAL plans stay alive."""
    from yolov3_amd.engine import PlanCache, plan_cache

    m, _ = build_pair("yolov3-tiny", 80, 31, dev, torch.float16)
    pc = plan_cache(m)
    shapes = [(96, 160), (128, 160), (160, 160), (160, 128), (160, 96), (64, 160), (160, 64), (96, 96), (128, 128), (192, 128), (128, 192), (224, 160)]
    n_banks, ptrs = None, None
    outs = {}
    for h, w in shapes:
        x = torch.rand(2, 3, h, w, generator=torch.Generator().manual_seed(h * 1000 + w)).to(dev).half()
        with torch.no_grad():
            outs[(h, w)] = m(x)[0].float().cpu()
        banks = sorted(cw.filt.data_ptr() for cw in pc.weights.values())
        if n_banks is None:
            n_banks, ptrs = len(banks), banks
        assert len(banks) == n_banks and banks == ptrs, "a new input shape re-packed the filters"
        assert len(pc.plans) <= PlanCache.MAX_EVAL
    assert len(pc.plans) == min(len(shapes), PlanCache.MAX_EVAL)
    # an evicted shape compiles again and gives the same answer
    h, w = shapes[0]
    x = torch.rand(2, 3, h, w, generator=torch.Generator().manual_seed(h * 1000 + w)).to(dev).half()
    with torch.no_grad():
        assert torch.equal(m(x)[0].float().cpu(), outs[(h, w)])
    # in-place parameter updates (an optimizer step) invalidate the banks
    with torch.no_grad():
        next(m.parameters()).mul_(1.5)
        again = m(x)[0].float().cpu()
    assert not torch.equal(again, outs[(h, w)]) and len(pc.plans) == 1


def test_conv_beyond_2gib_output(dev):
    """VERDICT r1 'scaling traps': a conv whose output exceeds the 2^31-byte reach of a buffer descriptor (batch 128 @640x640 layer 1,
    batch 32 @1280x1280) runs as several launches over image ranges; checked on the first / boundary / last images against conv2d."""
    _lib, ops = _ops()
    n, h, w, cin, cout = 42, 320, 320, 32, 256        # output 42 x 320 x 320 x 256 x 2 B = 2.2 GB -> two launches of 21 images
    g = torch.Generator().manual_seed(5)
    dtype = torch.float16
    wt = torch.randn(cout, cin, 1, 1, generator=g) / math.sqrt(cin)
    b = torch.randn(cout, generator=g) * 0.5
    filt = ops.pack_filter(wt.to(dev), cout, cin, dtype)
    xv = ops.View.alloc(n, h, w, cin, dtype, dev)
    xv.buf.copy_((torch.rand(xv.buf.numel(), generator=g) * 2 - 1).to(dev).to(dtype))
    yv = ops.View.alloc(n, h, w, cout, dtype, dev)
    assert yv.buf.numel() * 2 > 2**31
    ops.conv2d(xv, filt, b.to(dev), yv, 1, 1, True, None, workspace=conv_ws(dev))
    torch.cuda.synchronize()
    for img in (0, 20, 21, 41):
        xi = xv.as_nhwc()[img].float().cpu().permute(2, 0, 1)[None]
        ref = F.silu(F.conv2d(xi, wt.to(dtype).float(), b))
        out = yv.as_nhwc()[img].float().cpu().permute(2, 0, 1)[None]
        _conv_tol_check(f"img{img}", dtype, out, ref)


def test_train_two_outstanding_forwards_keep_their_activations(dev):
    """two train-mode forwards of one shape before a backward (micro-batches whose losses are summed) used to share the
    saved activations of ONE static plan -> silently wrong gradients.  Each outstanding forward now owns a plan; gradients of
    loss(x1) + loss(x2) equal the sum of the separately computed ones, and a forward whose state was overwritten raises."""
    from yolov3_amd import ComputeLoss

    hyp = dict(box=0.05, cls=0.5, cls_pw=1.0, obj=1.0, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0, label_smoothing=0.0)
    m, _ = build_pair("yolov3-tiny", 80, 37, dev, torch.float32)
    m.train()
    m.hyp = hyp
    crit = ComputeLoss(m)
    for mod in m.modules():   # frozen running statistics: the three runs below must see the same BatchNorm state
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.momentum = 0.0
    xs = [torch.rand(2, 3, 96, 96, generator=torch.Generator().manual_seed(s)).to(dev) for s in (1, 2)]
    tgs = [yo.synth_targets(2, 80, seed=s).to(dev) for s in (3, 4)]
    sep = []
    for x, tg in zip(xs, tgs):
        m.zero_grad(set_to_none=True)
        crit(m(x), tg)[0].backward()
        sep.append([p.grad.clone() for p in m.parameters()])
    m.zero_grad(set_to_none=True)
    l1 = crit(m(xs[0]), tgs[0])[0]
    with torch.no_grad():
        m(xs[1])                                  # a no_grad train-mode pass in between must not disturb the saved state either
    l2 = crit(m(xs[1]), tgs[1])[0]
    (l1 + l2).backward()
    torch.cuda.synchronize()
    for p, a, b in zip(m.parameters(), sep[0], sep[1]):
        torch.testing.assert_close(p.grad, a + b, rtol=1e-4, atol=1e-6)
    # more outstanding forwards than plans: the oldest one's backward must fail loudly, not compute garbage
    m.zero_grad(set_to_none=True)
    held = [crit(m(xs[0]), tgs[0])[0] for _ in range(3)]
    with pytest.raises(RuntimeError, match="overwritten by a later forward"):
        held[0].backward()
    held[2].backward()
    # a forward whose graph is dropped without a backward frees its slot (round-2 advisor finding): afterwards a fresh forward / backward
    # pair runs in slot 0 again and no plan is left marked busy
    from yolov3_amd.engine import plan_cache

    del held
    m.zero_grad(set_to_none=True)
    dropped = crit(m(xs[0]), tgs[0])[0]
    del dropped
    import gc

    gc.collect()
    tplans = [p_ for k, p_ in plan_cache(m).plans.items() if k[0] == "train"]
    assert tplans and not any(p_.outstanding for p_ in tplans), [p_.outstanding for p_ in tplans]
    crit(m(xs[0]), tgs[0])[0].backward()
    torch.cuda.synchronize()
    for p, a in zip(m.parameters(), sep[0]):
        torch.testing.assert_close(p.grad, a, rtol=1e-4, atol=1e-6)


def test_gradient_exchange_over_rccl_one_rank(dev):
    """the data-parallel exchange step on a DEVICE: parallel.init('nccl') (RCCL communicator), GradBuckets with
    the bucket / side-stream / event machinery forced on at world size 1, gradients written into the backward's flat arena and
    all-reduced (AVG) in place.  With one rank the average is the identity, so every parameter gradient must equal the one computed
    without grad_sync bit for bit; at least two collectives were issued and they ran on arena ranges (no flatten copies)."""
    import torch.distributed as dist

    from yolov3_amd import ComputeLoss, parallel

    hyp = dict(box=0.05, cls=0.5, cls_pw=1.0, obj=1.0, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0, label_smoothing=0.0)
    m, _ = build_pair("yolov3-tiny", 80, 41, dev, torch.float32)
    m.train()
    m.hyp = hyp
    crit = ComputeLoss(m)
    x = torch.rand(4, 3, 128, 128, generator=torch.Generator().manual_seed(1)).to(dev)
    tg = yo.synth_targets(4, 80, seed=2).to(dev)

    def grads():
        m.zero_grad(set_to_none=True)
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.momentum = 0.0
        with torch.autocast("cuda", dtype=torch.float16):
            loss, _ = crit(m(x), tg)
        (loss * 64.0).backward()
        torch.cuda.synchronize()
        return [p.grad.clone() for p in m.parameters()]

    base = grads()
    try:
        parallel.init("nccl", force=True)
        assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
        parallel.broadcast_parameters(m)
        gb = parallel.GradBuckets(bucket_bytes=8 << 20, force=True)
        launches = []
        orig = gb._reduce
        gb._reduce = lambda flat: (launches.append((flat.numel(), flat._base is not None)), orig(flat))[1]
        m.grad_sync = gb
        synced = grads()
        assert len(launches) >= 2 and all(in_arena for _, in_arena in launches), launches
        for a, b in zip(base, synced):
            assert torch.equal(a, b)
        assert parallel.max_over_ranks(1.5, dev) == 1.5
    finally:
        m.grad_sync = None
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("backend,exchange", [("nccl", "all_reduce"), ("gloo", "all_reduce"), ("nccl", "direct"), ("gloo", "direct")])
def test_gradient_exchange_two_ranks(tmp_path, backend, exchange):
    """two ranks through torch.distributed.run: both ranks end with the same averaged gradients = mean of the two single-rank gradients, on
    device buffers, with either form of the bucket collective (parallel.GradBuckets: one all-reduce, or all-to-all + owner's sum + all-gather).  nccl (RCCL, one GPU per rank) needs two GPUs; gloo runs on the 1-GPU box too -- the two ranks share the device
    (parallel.local_device), the collective goes through the host: the same rendezvous, rank / device mapping, bucket, side-stream and
    SUM + divide code as a multi-GPU job, which is what has to work the first time the driver launches N > 1."""
    import subprocess
    import sys

    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (the round-end 8-GPU node; the 1-GPU test box runs the gloo form)")

    script = tmp_path / "two_rank.py"
    script.write_text(f"""
import sys, torch, yaml
sys.path.insert(0, {str(ROOT)!r})
from yolov3_amd import ComputeLoss, DetectionModel, parallel
from oracle import yolo_oracle as yo
rank, local_rank, world = parallel.init({backend!r})
dev = parallel.local_device(local_rank)
torch.cuda.set_device(dev)
torch.manual_seed(0)
m = DetectionModel("yolov3-tiny.yaml", nc=80).to(dev).train()
m.hyp = dict(box=0.05, cls=0.5, cls_pw=1.0, obj=1.0, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0, label_smoothing=0.0)
parallel.broadcast_parameters(m)
crit = ComputeLoss(m)
x = torch.rand(4, 3, 128, 128, generator=torch.Generator().manual_seed(10 + rank)).to(dev)
tg = yo.synth_targets(4, 80, seed=20 + rank).to(dev)
def grads(sync):
    m.grad_sync = parallel.GradBuckets(bucket_bytes=8 << 20, exchange={exchange!r}) if sync else None
    m.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.float16):   # the product path: MFMA kernels, fixed-order reductions (the fp32 parity path's direct filter gradient uses fp32 atomics)
        loss = crit(m(x), tg)[0]
    (loss * 64.0).backward()
    torch.cuda.synchronize()
    return torch.cat([p.grad.flatten() for p in m.parameters()])
own = grads(False)
avg = grads(True)
assert m.grad_sync.collectives[{exchange!r}] >= 2 and sum(m.grad_sync.collectives.values()) == m.grad_sync.collectives[{exchange!r}], m.grad_sync.collectives
ref = own.clone()
torch.distributed.all_reduce(ref)
ref /= world
# (the two backward passes of a rank run the same kernels on the same data and nothing in them depends on the order waves retire in -- the loss backward
#  sums duplicated cells in slot order, no atomics -- so the averaged gradients equal the mean of the unsynchronised ones bit for bit, also with two processes on one GPU)
assert torch.equal(avg, ref), (float((avg - ref).abs().max()), float(ref.abs().max()))
print("rank", rank, "ok")
parallel.finalize()
""")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(29533 + (exchange == "direct")), str(script)],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.count("ok") == 2, out.stdout + out.stderr


def test_multi_scale_training_reuses_one_arena(dev):
    """The reference's --multi-scale training (train.py:394-399) draws a new size from [0.5, 1.5] x imgsz in steps of the grid size for EVERY batch: 21 sizes at
    imgsz 640.  All of them stay compiled (PlanCache.MAX_TRAIN_SHAPES >= 24) as views into ONE activation arena per slot: after the first pass over the sizes no plan
    is built, the arena is not re-allocated, and a size that comes round again gives bit-identical gradients (nothing of another shape's run leaks into it)."""
    import random

    from yolov3_amd import ComputeLoss, train_engine
    from yolov3_amd.engine import plan_cache

    hyp = dict(box=0.05, cls=0.5, cls_pw=1.0, obj=1.0, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0, label_smoothing=0.0)
    m, _ = build_pair("yolov3-tiny", 80, 41, dev, torch.float32)
    m.train()
    m.hyp = hyp
    for mod in m.modules():   # frozen running statistics: a shape's second visit sees the same state as its first
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.momentum = 0.0
    crit = ComputeLoss(m)
    imgsz, gs, bs = 640, 32, 2
    sizes = sorted({random.Random(s_).randrange(int(imgsz * 0.5), int(imgsz * 1.5) + gs) // gs * gs for s_ in range(4000)})
    assert len(sizes) == 21 and sizes[0] == 320 and sizes[-1] == 960
    tg = yo.synth_targets(bs, 80, seed=5).to(dev)

    def grads(sz):
        x = torch.rand(bs, 3, sz, sz, generator=torch.Generator().manual_seed(sz)).to(dev)
        m.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16):
            loss, _ = crit(m(x), tg)
        (loss * 64.0).backward()
        return torch.cat([p_.grad.flatten() for p_ in m.parameters()])

    order = sizes[:]
    random.Random(7).shuffle(order)
    first = {sz: grads(sz) for sz in order}
    torch.cuda.synchronize()
    pc = plan_cache(m)
    slot = pc.train_slots(torch.float16, dev, train_engine.TrainSlot)[0]
    builds, allocs, ptr = train_engine.PLAN_BUILDS, slot.arena_allocations, slot.arena.data_ptr()
    assert sum(1 for k in pc.plans if k[0] == "train") == 21, [k for k in pc.plans if k[0] == "train"]
    random.Random(8).shuffle(order)
    for sz in order + order[::-1]:
        g = grads(sz)
        assert torch.equal(g, first[sz]), f"{sz} x {sz}: gradients changed between two visits of the shape"
    torch.cuda.synchronize()
    assert train_engine.PLAN_BUILDS == builds, f"{train_engine.PLAN_BUILDS - builds} plans were rebuilt in the second / third pass"
    assert slot.arena_allocations == allocs and slot.arena.data_ptr() == ptr
    # the arena holds the largest shape exactly once, not the sum of the shapes
    big = pc.plans[("train", bs, 960, 960, torch.float16, dev.index, 0)]
    assert slot.arena.numel() == big._act_off


@pytest.mark.parametrize("cfg,hw", [("yolov3-tiny", 160), ("yolov3", 128)])
def test_filter_gradients_on_the_side_stream_are_bit_identical(dev, monkeypatch, cfg, hw):
    """Round 6: the filter gradients of a backward run on a second HIP stream by default (train_engine.TrainPlan.wgrad; Y3_WGRAD_STREAM=0: one stream) -- nothing
    downstream in the backward needs them.  Same kernels, same split-K order, one workspace used in stream order: every parameter gradient must equal the
    single-stream backward's to the last bit, over several steps (the workspace and the gradient arena are re-used / re-allocated between them) and with other work
    queued on the compute stream in between."""
    from yolov3_amd import ComputeLoss
    from yolov3_amd.engine import plan_cache
    from yolov3_amd import train_engine

    hyp = dict(box=0.05, cls=0.5, cls_pw=1.0, obj=1.0, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0, label_smoothing=0.0)
    bs = 4
    tg = yo.synth_targets(bs, 80, seed=5).to(dev)
    out = {}
    for arm in ("0", "1"):
        monkeypatch.setenv("Y3_WGRAD_STREAM", arm)
        m, _ = build_pair(cfg, 80, 41, dev, torch.float32)
        m.train()
        m.hyp = hyp
        crit = ComputeLoss(m)
        steps = []
        for it in range(3):
            x = torch.rand(bs, 3, hw, hw + 32 * (it % 2), generator=torch.Generator().manual_seed(it)).to(dev)   # two shapes: two plans of one slot share the workspace
            m.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.float16):
                loss, _ = crit(m(x), tg)
            (loss * 64.0).backward()
            junk = torch.randn(1 << 22, device=dev).sin_().sum()   # the compute stream moves on while the side stream may still be writing gradients
            steps.append(torch.cat([p_.grad.flatten() for p_ in m.parameters()]).clone())
            del junk
        torch.cuda.synchronize()
        pl = [v for k, v in plan_cache(m).plans.items() if k[0] == "train"]
        assert pl and all((p_.wgrad_stream is not None) == (arm == "1") for p_ in pl), "the switch was not read"
        out[arm] = steps
    for it, (a, b) in enumerate(zip(out["0"], out["1"])):
        assert torch.isfinite(a).all() and a.abs().sum() > 0
        assert torch.equal(a, b), f"step {it}: max |d| {(a - b).abs().max().item():.3e}"


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(3, 70, 150), (2, 64, 64), (1, 5, 200)], ids=["ragged_tiles", "whole_tiles", "short_wide"])
def test_stem_layer0_by_recomputation_matches_the_stored_path(dev, dtype, shape):
    """Round 6: layer 0 of the training step without its pre-BatchNorm tensor (csrc/stem.hip: y3_stem_conv_stats_only + y3_stem_conv_fwd_bn; csrc/train.hip:
    y3_stem_bn_bwd_wgrad_recompute).  Forward: the statistics rows and the normalised activation are the SAME BITS as the stored path's (stem_conv_stats, then
    y3_bn_act_fwd on the stored u).  Backward: dW / dgamma / dbeta against the stored-u kernels (another order of the fp64 partial sums: 1e-6 level) and against
    fp32 autograd of conv -> affine -> SiLU on the same rounded operands."""
    _lib, ops = _ops()
    import ctypes as C

    L = _lib.lib()
    n, h, w = shape
    g = torch.Generator().manual_seed(17)
    x = torch.rand(n, 3, h, w, generator=g)
    wt = torch.randn(32, 3, 3, 3, generator=g) / math.sqrt(27)
    xd = x.to(dev)
    filt = ops.pack_filter_stem(wt.to(dev), 32, dtype)
    zb = torch.zeros(32, device=dev)
    rows_cap = ops.stem_conv_stats_rows(n, h, w)
    # stored path
    u = ops.View.alloc(n, h, w, 32, dtype, dev)
    r_ref = torch.zeros(rows_cap * 64, device=dev)
    nr = ops.stem_conv_stats(xd, filt, zb, u, r_ref, rows_cap)
    mean = (torch.randn(32, generator=g) * 0.2).to(dev)
    invstd = (torch.rand(32, generator=g) + 0.5).to(dev)
    gamma = (torch.rand(32, generator=g) + 0.5).to(dev)
    beta = (torch.randn(32, generator=g) * 0.3).to(dev)
    scale = (gamma * invstd).contiguous()
    shift = (beta - mean * gamma * invstd).contiguous()
    y_ref = ops.View.alloc(n, h, w, 32, dtype, dev)
    ut, yt = u.y3(), y_ref.y3()
    _lib.check(L.y3_bn_act_fwd(C.byref(ut), scale.data_ptr(), shift.data_ptr(), None, C.byref(yt), ops.dtype_code(dtype), _lib.Y3_ACT_SILU, ops.stream_ptr()), "y3_bn_act_fwd")
    # recomputation
    r_new = torch.zeros(rows_cap * 64, device=dev)
    like = ops.View(torch.empty(0, dtype=dtype, device=dev), n, h, w, 32, 32, 0)
    assert ops.stem_conv_stats_only(xd, filt, like, r_new, rows_cap) == nr
    y_new = ops.View.alloc(n, h, w, 32, dtype, dev)
    y_new.buf.fill_(float("nan"))
    ops.stem_conv_bn(xd, filt, scale, shift, _lib.Y3_ACT_SILU, y_new)
    torch.cuda.synchronize()
    assert torch.equal(r_new, r_ref), "statistics rows differ"
    assert torch.equal(y_new.buf.view(torch.int16), y_ref.buf.view(torch.int16)), "normalised activation differs"
    # backward
    gy = ops.View.alloc(n, h, w, 32, dtype, dev)
    gy.buf.copy_(torch.randn(n * h * w * 32, generator=g).to(dtype))
    ws = ops.stem_bwd_workspace(dev)
    outs = []
    for arm in ("stored", "recompute"):
        sums = ops.bn_scratch(32, dev)
        dg, db, dw = torch.zeros(32, device=dev), torch.zeros(32, device=dev), torch.zeros(32, 3, 3, 3, device=dev)
        if arm == "stored":
            ops.stem_bn_bwd_wgrad(xd, u, gy, scale, shift, mean, invstd, _lib.Y3_ACT_SILU, sums, dg, db, dw, ws)
        else:
            ops.stem_bn_bwd_wgrad_recompute(xd, filt, gy, scale, shift, mean, invstd, _lib.Y3_ACT_SILU, sums, dg, db, dw, ws)
            dw2 = torch.zeros_like(dw)
            ops.stem_bn_bwd_wgrad_recompute(xd, filt, gy, scale, shift, mean, invstd, _lib.Y3_ACT_SILU, ops.bn_scratch(32, dev), torch.zeros(32, device=dev), torch.zeros(32, device=dev), dw2, ws)
            torch.cuda.synchronize()
            assert torch.equal(dw, dw2), "not run-to-run deterministic"
        torch.cuda.synchronize()
        outs.append((dg.cpu(), db.cpu(), dw.cpu()))
    for a, b, what in zip(outs[0], outs[1], ("dgamma", "dbeta", "dw")):
        e = (a - b).abs().max().item() / max(a.abs().max().item(), 1e-30)
        assert e < 2e-5, f"{what}: stored vs recomputed {e:.2e}"
    # fp32 autograd on the same rounded operands: u as stored, the BatchNorm backward written out with the given (mean, invstd) as constants
    uf = u.as_nhwc().float().cpu()
    gf = gy.as_nhwc().float().cpu()
    z = uf * scale.cpu() + shift.cpu()
    sg = torch.sigmoid(z)
    dz = gf * (sg + z * sg * (1 - sg))
    xh = (uf - mean.cpu()) * invstd.cpu()
    M = n * h * w
    db_ref = dz.sum((0, 1, 2))
    dg_ref = (dz * xh).sum((0, 1, 2))
    du = (scale.cpu() * (dz - db_ref / M - xh * (dg_ref / M))).to(dtype).float()
    xr = x.to(dtype).float().requires_grad_(False)
    wz = torch.zeros(32, 3, 3, 3, requires_grad=True)
    F.conv2d(xr, wz, None, stride=1, padding=1).backward(du.permute(0, 3, 1, 2))
    dgn, dbn, dwn = outs[1]
    assert (dgn - dg_ref).abs().max().item() / dg_ref.abs().max().item() < 1e-4
    assert (dbn - db_ref).abs().max().item() / db_ref.abs().max().item() < 1e-4
    tol = 2e-3 if dtype == torch.float16 else 1.5e-2   # du is rounded to T in both; products of rounded operands, fp32 sums
    assert (dwn - wz.grad).abs().max().item() / wz.grad.abs().max().item() < tol


def test_train_step_with_and_without_layer0_recomputation(dev, monkeypatch):
    """the whole step with Y3_STEM_RECOMPUTE = 0 / 1 (yolov3, 160 x 128, autocast fp16; the switch is off by default -- profiles/r06_stem_recompute_ab.txt): layer 0's
    activation is bit-identical, so the loss and every gradient behind layer 0 are; layer 0's own three gradients differ by the order of its fp64 partial sums only"""
    from yolov3_amd import ComputeLoss

    hyp = dict(box=0.05, cls=0.5, cls_pw=1.0, obj=1.0, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0, label_smoothing=0.0)
    tg = yo.synth_targets(4, 80, seed=5).to(dev)
    x = torch.rand(4, 3, 160, 128, generator=torch.Generator().manual_seed(3)).to(dev)
    res = {}
    for arm in ("0", "1"):
        monkeypatch.setenv("Y3_STEM_RECOMPUTE", arm)
        m, _ = build_pair("yolov3", 80, 41, dev, torch.float32)
        m.train()
        m.hyp = hyp
        crit = ComputeLoss(m)
        with torch.autocast("cuda", dtype=torch.float16):
            loss, _ = crit(m(x), tg)
        (loss * 64.0).backward()
        torch.cuda.synchronize()
        res[arm] = (float(loss), {k: p_.grad.clone() for k, p_ in m.named_parameters()}, {k: v.clone() for k, v in m.state_dict().items() if "running" in k})
    assert res["0"][0] == res["1"][0], (res["0"][0], res["1"][0])
    for k, v in res["0"][2].items():
        assert torch.equal(v, res["1"][2][k]), f"running statistics {k}"
    for k, ga in res["0"][1].items():
        gb = res["1"][1][k]
        if k.startswith("model.0."):
            assert (ga - gb).abs().max().item() <= 2e-5 * ga.abs().max().item(), k
        else:
            assert torch.equal(ga, gb), k


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
@pytest.mark.parametrize("n,h,w,c,k,s,p,zr,zb", [(2, 20, 20, 64, 13, 1, 6, 0, 0), (2, 20, 20, 64, 9, 1, 4, 0, 0), (3, 11, 17, 40, 5, 1, 2, 0, 0), (2, 26, 26, 32, 2, 2, 0, 0, 0),
                                                  (2, 13, 13, 48, 2, 1, 0, 1, 1), (1, 9, 7, 16, 3, 2, 1, 0, 0)],
                         ids=["spp13", "spp9", "spp5_odd", "tiny_2x2s2", "tiny_zeropad_s1", "k3s2"])
def test_maxpool_backward_indexed_form(dev, dtype, n, h, w, c, k, s, p, zr, zb):
    """y3_maxpool2d_bwd_ws (round 6: first-maximum index per window + k^2 look-ups per element) against the gather form y3_maxpool2d_bwd it replaces in the training plans --
    the same bits, write and accumulate, with ties in the input (half the values are repeated) -- and against torch's max_pool2d autograd on tie-free inputs."""
    _lib, ops = _ops()
    import ctypes as C

    L = _lib.lib()
    g = torch.Generator().manual_seed(23)
    ho, wo = (h + zb + 2 * p - k) // s + 1, (w + zr + 2 * p - k) // s + 1
    for ties in (True, False):
        xt = torch.randn(n, c, h, w, generator=g)
        if ties:
            xt = (xt * 2).round() / 2          # many equal values per window: the FIRST maximum in row-major order takes the gradient
        xt = xt.to(dtype).float()
        gy = torch.randn(n, c, ho, wo, generator=g).to(dtype).float()
        xv = ops.View.alloc(n, h, w, c, dtype, dev)
        ops.nchw_to_nhwc(xt.to(dev), xv)
        gv = ops.View.alloc(n, ho, wo, c, dtype, dev)
        ops.nchw_to_nhwc(gy.to(dev), gv)
        outs = []
        for form in ("gather", "indexed"):
            dx = ops.View.alloc(n, h, w, c, dtype, dev)
            dx.buf.fill_(float("nan"))
            for acc in (False, True):
                if form == "gather":
                    a, b, d = xv.y3(), gv.y3(), dx.y3()
                    _lib.check(L.y3_maxpool2d_bwd(C.byref(a), C.byref(b), C.byref(d), ops.dtype_code(dtype), k, s, p, zr, zb, int(acc), ops.stream_ptr()), "y3_maxpool2d_bwd")
                else:
                    ops.maxpool2d_bwd(xv, gv, dx, k, s, p, zr, zb, accumulate=acc)
            torch.cuda.synchronize()
            outs.append(dx.as_nhwc().float().cpu().clone())
        assert torch.equal(outs[0], outs[1]), f"ties={ties}: indexed form differs from the gather form"
        if not ties and zr == 0 and zb == 0:
            xr = xt.clone().requires_grad_(True)
            F.max_pool2d(xr, k, s, p).backward(gy)
            ref2 = (2 * xr.grad).to(dtype).float() if dtype == torch.float32 else None   # (written once, accumulated once: 2 x the gradient; exact in fp32 only)
            if ref2 is not None:
                torch.testing.assert_close(outs[1].permute(0, 3, 1, 2), ref2, rtol=1e-6, atol=1e-6)


def test_loss_rejects_out_of_range_targets(dev):
    """a target with image index >= bs (or < 0) or class >= nc used to index out of bounds in the match kernels; the
    reference raises an IndexError.  Here the row is dropped on the device and the loss comes back NaN (no host sync to raise
    from), and so does every gradient of the backward: loud, and nothing is read or written out of bounds."""
    hyp = dict(box=0.05, cls=0.5, cls_pw=1.0, obj=1.0, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0, label_smoothing=0.0)
    m, crit = _loss_setup(dev, "yolov3-tiny", 5, 64, hyp)
    p = [(torch.rand(2, 3, 64 // s, 64 // s, 10, device=dev) * 4 - 2).requires_grad_(True) for s in (16, 32)]
    good = torch.tensor([[0, 1, 0.5, 0.5, 0.3, 0.3], [1, 4, 0.25, 0.75, 0.2, 0.4]], device=dev)
    loss, _ = crit(p, good)
    assert torch.isfinite(loss).all()
    for bad_row in ([2, 1, 0.5, 0.5, 0.3, 0.3], [-1, 1, 0.5, 0.5, 0.3, 0.3], [0, 5, 0.5, 0.5, 0.3, 0.3], [0, -2, 0.5, 0.5, 0.3, 0.3], [70000, 1, 0.5, 0.5, 0.3, 0.3]):
        tg = torch.cat([good, torch.tensor([bad_row], device=dev, dtype=torch.float32)])
        for q in p:
            q.grad = None
        loss, items = crit(p, tg)
        loss.sum().backward()
        torch.cuda.synchronize()
        assert torch.isnan(loss).all() and torch.isnan(items).all(), bad_row
        # the backward is poisoned too (round 3): the objectness gradient reaches every cell, so every level's gradient is NaN and a
        # GradScaler-driven loop skips the step instead of applying the valid rows' gradients
        assert all(torch.isnan(q.grad[..., 4]).all() for q in p), bad_row


def test_model_rejects_wrong_channel_count(dev):
    """the stem kernel is handed the raw image pointer, so a 1-channel batch would read past the buffer; the reference
    raises a shape error"""
    m, _ = build_pair("yolov3-tiny", 80, 3, dev, torch.float16)
    with pytest.raises(ValueError, match="input channels"):
        m(torch.rand(1, 1, 64, 64, device=dev).half())
    with pytest.raises(ValueError, match="input channels"):
        m(torch.rand(1, 4, 64, 64, device=dev).half())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape,sdt,div,act", [((2, 3, 40, 72), torch.float16, 1.0, True), ((1, 3, 37, 131), torch.float32, 1.0, True), ((3, 3, 24, 200), torch.uint8, 255.0, True),
                                               ((2, 1, 19, 64), torch.float16, 1.0, False), ((5, 3, 128, 192), torch.float16, 1.0, True)],
                         ids=["l0", "ragged_f32src", "u8_div255", "cin1_noact", "more_tiles_than_blocks"])
def test_stem_bn_bwd_wgrad_matches_unfused_backward(dev, dtype, shape, sdt, div, act):
    """y3_stem_bn_bwd_wgrad (layer 0 backward: BatchNorm + SiLU backward and the filter gradient in one pass, du never stored) against the
    path it replaces -- y3_bn_act_bwd (du stored in T) + y3_conv2d_wgrad on the NHWC image: dgamma / dbeta bit-identical (same reduction),
    dW equal up to fp32 summation order (both multiply the SAME T-rounded du and image values); and against torch autograd in fp32."""
    import ctypes as C

    _lib, ops = _ops()
    n, cin, h, w = shape
    cout = 32
    g = torch.Generator().manual_seed(17)
    if sdt == torch.uint8:
        x = torch.randint(0, 256, (n, cin, h, w), generator=g, dtype=torch.uint8)
    else:
        x = torch.rand(n, cin, h, w, generator=g).to(sdt)
    xd = x.to(dev)
    u = (torch.randn(n, cout, h, w, generator=g) * 1.5).to(dtype)
    dy = (torch.randn(n, cout, h, w, generator=g) * 0.1).to(dtype)
    gamma, beta = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    uv, gv = ops.View.alloc(n, h, w, cout, dtype, dev), ops.View.alloc(n, h, w, cout, dtype, dev)
    ops.nchw_to_nhwc(u.to(dev), uv)
    ops.nchw_to_nhwc(dy.to(dev), gv)
    cnt = n * h * w
    uf = u.double()
    mean = uf.mean((0, 2, 3))
    var = uf.var((0, 2, 3), unbiased=False)
    invstd = (1.0 / torch.sqrt(var + 1e-3))
    scale = (gamma.double() * invstd).float().to(dev)
    shift = (beta.double() - mean * gamma.double() * invstd).float().to(dev)
    mean_d, invstd_d = mean.float().to(dev), invstd.float().to(dev)
    a = _lib.Y3_ACT_SILU if act else _lib.Y3_ACT_NONE
    # unfused: du stored, generic filter gradient
    sums = ops.bn_scratch(cout, dev)
    duv = ops.View.alloc(n, h, w, cout, dtype, dev)
    dg0, db0 = torch.empty(cout, device=dev), torch.empty(cout, device=dev)
    ut, gt, dt_ = uv.y3(), gv.y3(), duv.y3()
    _lib.check(_lib.lib().y3_bn_act_bwd(C.byref(ut), C.byref(gt), scale.data_ptr(), shift.data_ptr(), mean_d.data_ptr(), invstd_d.data_ptr(), ops.dtype_code(dtype), a,
                                        sums.data_ptr(), C.byref(dt_), dg0.data_ptr(), db0.data_ptr(), ops.stream_ptr()), "y3_bn_act_bwd")
    xin = ops.View.alloc(n, h, w, 8, dtype, dev)
    ops.nchw_to_nhwc(xd, xin, div)
    dw0, _ = ops.conv2d_wgrad(xin, duv, 3, 1, cout, cin)
    # fused
    sums1 = ops.bn_scratch(cout, dev)
    dg1, db1 = torch.empty(cout, device=dev), torch.empty(cout, device=dev)
    dw1 = torch.full((cout, cin, 3, 3), float("nan"), device=dev)
    ops.stem_bn_bwd_wgrad(xd, uv, gv, scale, shift, mean_d, invstd_d, a, sums1, dg1, db1, dw1, ops.stem_bwd_workspace(dev), div)
    torch.cuda.synchronize()
    assert torch.equal(dg0, dg1) and torch.equal(db0, db1)
    assert torch.isfinite(dw1).all()
    ref = dw0.abs().max().item()
    assert (dw0 - dw1).abs().max().item() <= 2e-5 * ref + 1e-6, f"fused vs unfused dW: {(dw0 - dw1).abs().max().item():.3e} of {ref:.3e}"
    # fp32 autograd of the same function of the rounded operands
    xq = (x.float() / div).to(dtype).float().requires_grad_(False)
    wt = torch.zeros(cout, cin, 3, 3, requires_grad=True)
    # d loss / d W with loss = <dy, act(bn(conv(x, W) + (u - conv(x, W)).detach()))>: the gradient wrt the conv OUTPUT at value u
    uu = u.float().clone().requires_grad_(True)
    z = (uu - mean.float().view(1, -1, 1, 1)) * (gamma * invstd.float()).view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)
    # batch statistics depend on u: use torch's batch_norm for the exact backward
    zz = F.batch_norm(uu, None, None, gamma, beta, True, 0.0, 1e-3)
    out = F.silu(zz) if act else zz
    (out * dy.float()).sum().backward()
    du_ref = uu.grad
    dw_ref = torch.nn.grad.conv2d_weight(xq, (cout, cin, 3, 3), du_ref, stride=1, padding=1)
    err = (dw1.cpu() - dw_ref).abs().max().item() / dw_ref.abs().max().item()
    assert err < (4e-3 if dtype == torch.float16 else 3e-2), f"dW vs fp32 autograd: {err:.3e}"


BNECK_CASES = [
    ("square", (2, 64, 64), True),
    ("odd_rect", (1, 37, 53), True),
    ("many_tiles_per_block", (9, 96, 224), True),      # 9 * 12 * 7 = 756 tiles > 512 persistent blocks: the grid-stride loop and its DMA prefetch
    ("smaller_than_a_tile", (1, 5, 7), True),
    ("no_shortcut", (2, 40, 72), False),
]


@pytest.mark.parametrize("c", [64, 128])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("name,shape,add", BNECK_CASES, ids=[c[0] for c in BNECK_CASES])
def test_bneck_pair_vs_fp32_reference(dev, dtype, name, shape, add, c):
    """y3_bneck_pair_fwd (Bottleneck(64, 64) / Bottleneck(128, 128) in one kernel, the C/2-channel intermediate kept in LDS) against torch fp32 convolutions on
    the same rounded operands, the intermediate rounded to the storage dtype as the two-launch form stores it and the residual added
    to the ROUNDED cv2 output (what `x + cv2(cv1(x))` does on half tensors); borders (cv2's zero padding applies to cv1's OUTPUT), odd
    sizes, images smaller than a tile; and bit-compared with the two generic launches it replaces where those exist."""
    _lib, ops = _ops()
    n, h, w = shape
    g = torch.Generator().manual_seed(23)
    cm = c // 2
    x = torch.randn(n, c, h, w, generator=g).to(dtype)
    w1 = (torch.randn(cm, c, 1, 1, generator=g) / math.sqrt(c)).to(dtype).float()
    b1 = torch.randn(cm, generator=g) * 0.1
    w2 = (torch.randn(c, cm, 3, 3, generator=g) / math.sqrt(9 * cm)).to(dtype).float()
    b2 = torch.randn(c, generator=g) * 0.1
    t = F.silu(F.conv2d(x.float(), w1, b1)).to(dtype).float()
    ref = F.silu(F.conv2d(t, w2, b2, padding=1)).to(dtype).float()
    if add:
        ref = ref + x.float()
    xv = ops.View.alloc(n, h, w, c, dtype, dev)
    ops.nchw_to_nhwc(x.to(dev), xv)
    f1 = ops.pack_filter(w1.to(dev), cm, c, dtype)
    f2 = ops.pack_filter(w2.to(dev), c, cm, dtype)
    b1d, b2d = b1.to(dev), b2.to(dev)
    yv = ops.View.alloc(n, h, w, c, dtype, dev)
    yv.buf.fill_(float("nan"))
    ops.bneck_pair(xv, f1, b1d, True, f2, b2d, True, add, yv)
    torch.cuda.synchronize()
    got = yv.as_nhwc().float().cpu().permute(0, 3, 1, 2)
    assert torch.isfinite(got).all(), "an output pixel was not written"
    tol = 2.0**-8 if dtype == torch.float16 else 2.0**-5
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    assert err < tol, f"{name} {dtype}: {err:.2e}"
    # the two launches it replaces: same arithmetic up to fp32 summation order inside each conv
    tv, y2 = ops.View.alloc(n, h, w, cm, dtype, dev), ops.View.alloc(n, h, w, c, dtype, dev)
    ops.conv2d(xv, f1, b1d, tv, 1, 1, True)
    ops.conv2d(tv, f2, b2d, y2, 3, 1, True, residual=xv if add else None)
    torch.cuda.synchronize()
    two = y2.as_nhwc().float().cpu().permute(0, 3, 1, 2)
    assert (got - two).abs().max().item() <= 2 * tol * ref.abs().max().item()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_pack_filter_jobs_matches_per_layer_packing(dev, dtype):
    """y3_pack_filter_jobs (every layer's forward + data-gradient bank in ONE launch, the training step's packing) against y3_pack_filter_pair /
    y3_pack_filter layer by layer: bit-identical banks, including channel-padded heads, a forward-only job and a job whose weights moved
    (the device table is rebuilt from the new pointer)."""
    _lib, ops = _ops()
    g = torch.Generator().manual_seed(3)
    shapes = [(64, 32, 3), (32, 64, 1), (255, 1024, 1), (128, 64, 3), (1024, 512, 3), (21, 256, 1)]
    ws = [torch.randn(co, ci, k, k, generator=g).to(dev) for co, ci, k in shapes]
    jobs = ops.PackJobs(dtype, dev)
    banks = []
    for i, (w, (co, ci, k)) in enumerate(zip(ws, shapes)):
        banks.append(jobs.add(w, (co + 7) // 8 * 8, ci, True, i != 3))
    # the contract (round 5, source-indexed tiles): banks are zero-filled once by whoever allocates them (PackJobs.add does), the launch writes only the elements that
    # come from a weight.  Run 1 on the zero-filled banks: bit-identical to the per-layer packers.  Run 2 on banks full of a sentinel: every weight element (and its
    # fragment-ordered copy) is rewritten, the row / K padding keeps the sentinel -- exactly cout_src * cin_src * k * k (x 2 with the second copy) elements change
    for sentinel in (None, 7.0):
        if sentinel is not None:
            for b in banks:
                for t in b:
                    if t is not None:
                        t.fill_(sentinel)
        jobs.run()
        torch.cuda.synchronize()
        for i, (w, (co, ci, k)) in enumerate(zip(ws, shapes)):
            cop = (co + 7) // 8 * 8
            f_ref, d_ref = ops.pack_filter_pair(w, cop, ci, dtype)
            pairs = [(banks[i][0], f_ref, "forward", f_ref.numel() == 2 * ((cop + 127) // 128 * 128) * ((k * k * ci + 63) // 64 * 64))]   # (3x3 banks of 256-row / 32-channel multiples carry a second copy)
            if i != 3:
                pairs.append((banks[i][1], d_ref, "data-gradient", d_ref.numel() == 2 * ((ci + 127) // 128 * 128) * ((k * k * cop + 63) // 64 * 64)))
                assert torch.equal(d_ref.view(torch.int16), ops.pack_filter_dgrad(w, cop, ci, dtype).view(torch.int16))
            else:
                assert banks[i][1] is None
            for got, ref, what, two_copies in pairs:
                if sentinel is None:
                    assert torch.equal(got.view(torch.int16), ref.view(torch.int16)), f"{what} bank {i}"
                else:
                    kept = got.float() == sentinel
                    assert torch.equal(got[~kept].view(torch.int16), ref[~kept].view(torch.int16)), f"{what} bank {i}: a written element differs"
                    assert int((ref[kept].float() != 0).sum()) == 0, f"{what} bank {i}: a weight element was not written"
                    assert int((~kept).sum()) == co * ci * k * k * (2 if two_copies else 1), f"{what} bank {i}: padding was written"
    for b in banks:   # back to the contract's state for the rest of the test
        for t in b:
            if t is not None:
                t.zero_()
    jobs.run()
    # a weight tensor that moved: the job table follows
    jobs.jobs[1] = (ws[1].clone() * 2.0,) + jobs.jobs[1][1:]
    jobs.run()
    torch.cuda.synchronize()
    assert torch.equal(banks[1][0].view(torch.int16), ops.pack_filter(ws[1] * 2.0, 32, 64, dtype).view(torch.int16))


def test_map_parity_on_synthetic_scenes(dev):
    """BASELINE target "mAP@0.5 within 0.1 of reference", measured the only way that is possible offline (no coco128, no pretrained
    weights): a yolov3-tiny trained by the HIP training path on seeded synthetic scenes (tests/map_parity.py), then the SAME weights
    evaluated with val.py's procedure by the HIP path and by the CPU oracle.  The trained model must actually detect (mAP@0.5 > 0.25, else
    the comparison is vacuous); fp32 engine vs reference CPU path: |delta| <= 0.001 (0.1 mAP points) on mAP@0.5 and mAP@0.5:0.95;
    fp16 engine: <= 0.01 (1 point)."""
    import map_parity

    res = map_parity.run(steps=1200, dev=dev)
    print("[map parity]", {k: res[k] for k in ("reference_cpu_fp32", "hip_fp32", "hip_fp16", "abs_diff_fp32", "abs_diff_fp16", "train_seconds")})
    assert res["reference_cpu_fp32"]["mAP50"] > 0.25, res
    assert res["abs_diff_fp32"]["mAP50"] <= 1e-3 and res["abs_diff_fp32"]["mAP50-95"] <= 1e-3, res
    assert res["abs_diff_fp16"]["mAP50"] <= 1e-2 and res["abs_diff_fp16"]["mAP50-95"] <= 1e-2, res


# ------------------------------------------------------------------------------------------------ BASELINE configs[2] shapes: the training kernels at the sizes the bench runs
# (round-2 review: split-K slice counts, XCD-grouped grids, 16-byte non-temporal partial sums and the >= 128 MB BatchNorm forms ran in bench.py only)
BENCH_WGRAD_CASES = [
    # name, (n, h, w, cin, cout, k, s), (tile edge, xcd-grouped) the dispatcher must pick
    ("L6cv2_128_256_80", (64, 80, 80, 128, 256, 3, 1), (4, 1)),       # tile 4 = the padded-position kernel (wgrad_patch.h)
    ("L8cv2_256_512_40", (64, 40, 40, 256, 512, 3, 1), (4, 1)),
    ("L10cv2_512_1024_20", (64, 20, 20, 512, 1024, 3, 1), (4, 1)),
    ("L5_128_256_s2_160", (64, 160, 160, 128, 256, 3, 2), (256, 1)),
    ("L4cv2_64_128_160", (64, 160, 160, 64, 128, 3, 1), (3, 0)),      # tile 3 = the strip kernel (wgrad_strip.h)
    ("L3_64_128_s2_320", (64, 320, 320, 64, 128, 3, 2), (3, 0)),
    ("L2cv1_64_32_1x1_320", (64, 320, 320, 64, 32, 1, 1), (128, 0)),
    # the 64-filter tiles of the 640x640 / 320x320 maps: stride 1 / 2, the ragged third column tile of 288 columns, 6.5 M pixels
    ("L1_32_64_s2_640", (64, 640, 640, 32, 64, 3, 2), (3, 0)),
    ("L2cv2_32_64_320", (64, 320, 320, 32, 64, 3, 1), (3, 0)),
]


@pytest.mark.parametrize("name,shape,plan", BENCH_WGRAD_CASES, ids=[c[0] for c in BENCH_WGRAD_CASES])
def test_conv_wgrad_benchmark_shapes_fp16(dev, name, shape, plan):
    """y3_conv2d_wgrad at the batch-64 shapes of the benchmarked train step (409 600 / 102 400 / 25 600 / 1.6 M / 6.5 M pixels): the
    dispatcher's tile / slice / XCD decision is asserted through the dry-run query, the result is run-to-run deterministic, and 8 filter
    rows spread over the MFMA blocks and waves of the filter tile (every (tap, channel) column, i.e. every column tile and every pixel
    slice contributes to each) equal torch's fp32 autograd on the same rounded operands."""
    _lib, ops = _ops()
    n, h, w, cin, cout, k, s = shape
    dtype = torch.float16
    ho, wo = (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1
    g = torch.Generator(device=dev).manual_seed(11)
    xv = ops.View.alloc(n, h, w, cin, dtype, dev)
    xv.buf.normal_(generator=g)
    gv = ops.View.alloc(n, ho, wo, cout, dtype, dev)
    gv.buf.normal_(generator=g)
    tile, slices, xg = ops.conv2d_wgrad_plan(xv, cout, k, s)
    assert (tile, xg) == plan, f"{name}: dispatcher picked tile {tile}, xcd-grouped {xg}"
    assert slices >= 2, f"{name}: one pixel slice -- the split-K sum is not exercised"
    dw, _ = ops.conv2d_wgrad(xv, gv, k, s, cout, cin)
    dw2, _ = ops.conv2d_wgrad(xv, gv, k, s, cout, cin)
    torch.cuda.synchronize()
    assert torch.equal(dw, dw2), "filter gradient is not run-to-run deterministic"
    rows = sorted({0, 37 % cout, 70 % cout, 101 % cout, (cout // 2 + 2) % cout, (cout - 89) % cout, cout - 56 if cout > 56 else 5, cout - 1})
    # reference: conv2d restricted to those filters (cost ~ rows / cout of the full layer), image chunks to bound host memory
    wt = torch.zeros(len(rows), cin, k, k, requires_grad=True)
    xn, gn = xv.as_nhwc(), gv.as_nhwc()
    for i0 in range(0, n, 8):
        xc = xn[i0 : i0 + 8].permute(0, 3, 1, 2).float().cpu()
        gc = gn[i0 : i0 + 8][..., rows].permute(0, 3, 1, 2).float().cpu()
        F.conv2d(xc, wt, None, stride=s, padding=k // 2).backward(gc)
    ref = wt.grad
    got = dw.cpu()[rows]
    e_w = (got - ref).abs().max().item() / ref.abs().max().item()
    print(f"[wgrad {name}] tile {tile} slices {slices} xcd {xg}: max rel err {e_w:.2e}")
    # same rounded operands on both sides, fp32 accumulation over up to 6.5 M pixels in 51..1024 partial sums: measured 5e-7 .. 3e-6 (round 3)
    assert e_w < 2e-5, f"{name}: wgrad {e_w:.2e}"


BENCH_TRAIN_CONV_CASES = [
    # name, kind, (n, h, w, cin, cout, k, s) of the FORWARD layer, variant the dispatcher must pick for that launch.  kind: "fwd_stats" = the training
    # forward (no activation, BatchNorm statistics rows from the epilogue), "dgrad" = the data gradient through the forward kernel on the flipped bank
    # (accumulating into dx through the residual port), "dgrad_s2" = the four parity classes of a stride-2 data gradient
    ("L8cv2_fwd_stats", "fwd_stats", (64, 40, 40, 256, 512, 3, 1), "v10h"),
    ("L10cv2_fwd_stats", "fwd_stats", (64, 20, 20, 512, 1024, 3, 1), "v10"),
    ("L6cv2_fwd_stats", "fwd_stats", (64, 80, 80, 128, 256, 3, 1), "v10h"),
    ("L8cv2_dgrad", "dgrad", (64, 40, 40, 256, 512, 3, 1), "v10"),
    ("L10cv2_dgrad", "dgrad", (64, 20, 20, 512, 1024, 3, 1), "v10"),
    ("L4cv2_fwd_stats", "fwd_stats", (64, 160, 160, 64, 128, 3, 1), "strip"),
    ("L3_s2_fwd_stats", "fwd_stats", (64, 320, 320, 64, 128, 3, 2), "strip"),
    ("L2cv2_dgrad", "dgrad", (64, 320, 320, 32, 64, 3, 1), "strip"),
    ("L4cv2_dgrad", "dgrad", (64, 160, 160, 64, 128, 3, 1), "strip"),
    ("L1_dgrad_s2", "dgrad_s2", (64, 640, 640, 32, 64, 3, 2), "strip_quad"),      # dx = 64 x 640 x 640 x 32 = 1.68 GB: the tensor closest to a descriptor's 2 GiB reach
    ("L3_dgrad_s2", "dgrad_s2", (64, 320, 320, 64, 128, 3, 2), "strip_quad"),
    ("L5_dgrad_s2", "dgrad_s2", (64, 160, 160, 128, 256, 3, 2), "v3_quad"),
]


@pytest.mark.parametrize("name,kind,shape,variant", BENCH_TRAIN_CONV_CASES, ids=[c[0] for c in BENCH_TRAIN_CONV_CASES])
def test_conv_train_launches_benchmark_shapes_fp16(dev, name, kind, shape, variant):
    """The forward-with-statistics and data-gradient conv launches of the benchmarked train step at ITS shapes (batch 64; round-3 review: only the filter
    gradients were covered there): the dispatched kernel variant is asserted, a subset of output channels spread over the MFMA blocks / waves of a filter
    tile is compared with fp32 conv2d on the same rounded operands over ALL pixels (cost ~ subset / channels of the layer, image chunks bound the host
    memory), and the statistics rows summed in fp64 equal the statistics of the stored tensor."""
    _lib, ops = _ops()
    n, h, w, cin, cout, k, s = shape
    dtype = torch.float16
    ho, wo = (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1
    g = torch.Generator(device=dev).manual_seed(21)
    gc = torch.Generator().manual_seed(22)
    wt = (torch.randn(cout, cin, k, k, generator=gc) / math.sqrt(cin * k * k)).to(dtype).float()
    chunk = 4 if h >= 320 else 8

    def pick(c):
        return sorted({0, 37 % c, 70 % c, (c // 2 + 2) % c, (c - 89) % c, c - 1})

    if kind == "fwd_stats":
        xv = ops.View.alloc(n, h, w, cin, dtype, dev)
        xv.buf.normal_(generator=g)
        filt = ops.pack_filter(wt.to(dev), cout, cin, dtype)
        yv = ops.View.alloc(n, ho, wo, cout, dtype, dev)
        yv.buf.fill_(float("nan"))
        rows = ops.conv2d_stats_rows(xv, yv, k, s)
        buf = torch.full((rows * 2 * cout,), float("nan"), device=dev)
        assert ops.conv2d_stats(xv, filt, torch.zeros(cout, device=dev), yv, k, s, buf, rows) == rows
        assert ops.last_conv_variant() == variant, f"{name}: dispatcher picked {ops.last_conv_variant()}"
        torch.cuda.synchronize()
        sel = pick(cout)
        xn, yn = xv.as_nhwc(), yv.as_nhwc()
        bad, worst = 0, 0.0
        for i0 in range(0, n, chunk):
            ref = F.conv2d(xn[i0 : i0 + chunk].permute(0, 3, 1, 2).float().cpu(), wt[sel], None, stride=s, padding=k // 2)
            got = yn[i0 : i0 + chunk][..., sel].permute(0, 3, 1, 2).float().cpu()
            err = (got - ref).abs()
            bad += (err > 2.0**-10 * ref.abs() + 2e-3).sum().item()
            worst = max(worst, err.max().item())
        assert bad == 0, f"{name}: {bad} outputs outside tolerance, max abs err {worst:.3g}"
        y2 = yn.reshape(-1, cout)
        s0 = torch.zeros(cout, dtype=torch.float64, device=dev)
        s1 = torch.zeros(cout, dtype=torch.float64, device=dev)
        a0 = torch.zeros(cout, dtype=torch.float64, device=dev)
        step = 1 << 18
        for r0 in range(0, y2.shape[0], step):
            u = y2[r0 : r0 + step].double()
            s0 += u.sum(0); s1 += (u * u).sum(0); a0 += u.abs().sum(0)
        assert torch.isfinite(s0).all(), "the output holds a non-finite value (an unwritten pixel)"
        tot = buf.view(rows, cout, 2).double().sum(0)
        assert torch.isfinite(tot).all(), "a statistics row was not written"
        assert (tot[:, 0] - s0).abs().max().item() <= 1e-5 * a0.max().item()
        assert (tot[:, 1] - s1).abs().max().item() <= 1e-5 * s1.max().item()
        return

    # data gradients: du (n, ho, wo, cout) -> dx (n, h, w, cin) = conv_transpose(du, w)
    gv = ops.View.alloc(n, ho, wo, cout, dtype, dev)
    gv.buf.normal_(generator=g)
    gx = ops.View.alloc(n, h, w, cin, dtype, dev)
    sel = pick(cin)
    if kind == "dgrad":
        filt_d = ops.pack_filter_dgrad(wt.to(dev), cout, cin, dtype)
        if variant == "strip":   # cv2 of a Bottleneck: its input has one consumer, the gradient is written (the strip kernel has no residual port)
            gx.buf.fill_(float("nan"))
            base = None
            ops.conv2d(gv, filt_d, torch.zeros(cin, device=dev), gx, k, 1, act=False, in_dilation=s)
        else:
            gx.buf.normal_(generator=g)   # what a fan-out already accumulated: the launch adds to it through the residual port
            base = gx.as_nhwc()[..., sel].float().cpu()
            ops.conv2d(gv, filt_d, torch.zeros(cin, device=dev), gx, k, 1, act=False, residual=gx, in_dilation=s)
    else:
        gx.buf.fill_(float("nan"))
        base = None
        ops.conv2d_dgrad_s2(wt.to(dev), gv, gx, accumulate=False)
    assert ops.last_conv_variant() == variant, f"{name}: dispatcher picked {ops.last_conv_variant()}"
    torch.cuda.synchronize()
    gn, xn = gv.as_nhwc(), gx.as_nhwc()
    bad, worst = 0, 0.0
    for i0 in range(0, n, chunk):
        du = gn[i0 : i0 + chunk].permute(0, 3, 1, 2).float().cpu()
        ref = F.conv_transpose2d(du, wt[:, sel], None, stride=s, padding=k // 2, output_padding=(h - ((ho - 1) * s - 2 * (k // 2) + k), w - ((wo - 1) * s - 2 * (k // 2) + k)))
        if base is not None:
            ref = ref + base[i0 : i0 + chunk].permute(0, 3, 1, 2)
        got = xn[i0 : i0 + chunk][..., sel].permute(0, 3, 1, 2).float().cpu()
        err = (got - ref).abs()
        bad += (~(err <= 2.0**-10 * ref.abs() + 4e-3)).sum().item()   # (NaN = an unwritten pixel counts as bad)
        worst = max(worst, err.max().item())
    assert bad == 0, f"{name}: {bad} gradient elements outside tolerance, max abs err {worst:.3g}"


def _bn_reference(u, dy, gamma, beta, eps, act, res):
    """fp64 torch reference of train-mode act(bn(u)) (+ res) and its backward (du, dgamma, dbeta), NHWC (M, C) operands"""
    u = u.double().requires_grad_(True)
    gm, bt = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    mean, var = u.mean(0), u.var(0, unbiased=False)
    z = (u - mean) / torch.sqrt(var + eps) * gm + bt
    y = z * torch.sigmoid(z) if act else z
    if res is not None:
        y = y + res.double()
    y.backward(dy.double())
    return y.detach(), u.grad, gm.grad, bt.grad, mean.detach(), var.detach()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("nt", [0, 1], ids=["plain", "nontemporal"])
@pytest.mark.parametrize("shape,act,residual", [((2, 40, 40, 128), True, True), ((3, 23, 19, 64), True, False), ((1, 64, 64, 256), False, False), ((2, 16, 16, 1024), True, True)],
                         ids=["c128_res", "c64_ragged", "c256_linear", "c1024_res"])
def test_bn_passes_plain_and_nontemporal_forms(dev, tune, dtype, nt, shape, act, residual):
    """The elementwise / reduction passes of train-mode BatchNorm + SiLU (y3_bn_stats_finalize, y3_bn_act_fwd, y3_bn_act_bwd(_res)) against
    an fp64 torch reference, in their plain form and -- knob bn_nt_bytes = 0 -- in the non-temporal form the library takes for tensors of
    >= 128 MB (the batch-64 activations of the 640 / 320 / 160-pixel layers); both forms must agree bit for bit."""
    import ctypes as C

    _lib, ops = _ops()
    L = _lib.lib()
    n, h, w, c = shape
    M = n * h * w
    g = torch.Generator(device=dev).manual_seed(3)
    outs = {}
    for form in ([0, 1] if nt else [0]):
        tune("bn_nt_bytes", 0 if form else 1 << 62)
        g.manual_seed(3)
        uv, yv, dyv, duv = (ops.View.alloc(n, h, w, c, dtype, dev) for _ in range(4))
        uv.buf.normal_(generator=g).mul_(1.5).add_(0.25)
        dyv.buf.normal_(generator=g)
        rv = gr = None
        if residual:
            rv, gr = ops.View.alloc(n, h, w, c, dtype, dev), ops.View.alloc(n, h, w, c, dtype, dev)
            rv.buf.normal_(generator=g)
            gr.buf.fill_(0.5)
        gamma = torch.rand(c, device=dev, generator=g) + 0.5
        beta = torch.randn(c, device=dev, generator=g) * 0.3
        sums = ops.bn_scratch(c, dev)
        scale, shift, mean, invstd, dgamma, dbeta = (torch.empty(c, device=dev) for _ in range(6))
        rmean, rvar = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        ut, yt, dyt, dut = uv.y3(), yv.y3(), dyv.y3(), duv.y3()
        dc, a, st = ops.dtype_code(dtype), (_lib.Y3_ACT_SILU if act else _lib.Y3_ACT_NONE), ops.stream_ptr()
        _lib.check(L.y3_bn_stats_finalize(C.byref(ut), dc, sums.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-3, 0.03, rmean.data_ptr(), rvar.data_ptr(), scale.data_ptr(),
                                          shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(), st), "y3_bn_stats_finalize")
        rt = rv.y3() if rv is not None else None
        _lib.check(L.y3_bn_act_fwd(C.byref(ut), scale.data_ptr(), shift.data_ptr(), C.byref(rt) if rt is not None else None, C.byref(yt), dc, a, st), "y3_bn_act_fwd")
        if residual:
            grt = gr.y3()
            _lib.check(L.y3_bn_act_bwd_res(C.byref(ut), C.byref(dyt), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(), dc, a, sums.data_ptr(), C.byref(dut),
                                           dgamma.data_ptr(), dbeta.data_ptr(), C.byref(grt), 1, st), "y3_bn_act_bwd_res")
        else:
            _lib.check(L.y3_bn_act_bwd(C.byref(ut), C.byref(dyt), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(), dc, a, sums.data_ptr(), C.byref(dut),
                                       dgamma.data_ptr(), dbeta.data_ptr(), st), "y3_bn_act_bwd")
        torch.cuda.synchronize()
        outs[form] = dict(y=yv.buf.clone(), du=duv.buf.clone(), dgamma=dgamma.clone(), dbeta=dbeta.clone(), mean=mean.clone(), invstd=invstd.clone(), rmean=rmean.clone(),
                          rvar=rvar.clone(), gres=gr.buf.clone() if gr is not None else None)
        if form == 0:
            yr, dur, dgr, dbr, mr, vr = _bn_reference(uv.buf.view(M, c), dyv.buf.view(M, c), gamma, beta, 1e-3, act, rv.buf.view(M, c) if rv is not None else None)
            ulp = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
            o = outs[0]
            assert (o["mean"].double() - mr).abs().max().item() <= 1e-5 * max(1.0, mr.abs().max().item())
            assert (o["invstd"].double() - 1 / torch.sqrt(vr + 1e-3)).abs().max().item() <= 1e-5 * o["invstd"].max().item()
            assert (o["rmean"].double() - 0.03 * mr).abs().max().item() <= 1e-6 + 1e-5 * mr.abs().max().item()
            assert (o["rvar"].double() - (0.97 + 0.03 * vr * M / (M - 1))).abs().max().item() <= 1e-5
            assert (o["y"].view(M, c).double() - yr).abs().max().item() <= 1.01 * ulp * yr.abs().max().item()
            assert (o["du"].view(M, c).double() - dur).abs().max().item() <= 1.5 * ulp * dur.abs().max().item() + 1e-6
            assert (o["dgamma"].double() - dgr).abs().max().item() <= 2e-5 * dyv.buf.float().abs().sum().item() / c
            assert (o["dbeta"].double() - dbr).abs().max().item() <= 2e-5 * dyv.buf.float().abs().sum().item() / c
            if gr is not None:
                assert torch.equal(o["gres"].float(), (0.5 + dyv.buf.float()).to(dtype).float()), "residual gradient accumulation"
    if nt:
        for kname in outs[0]:
            if outs[0][kname] is not None:
                assert torch.equal(outs[0][kname], outs[1][kname]), f"non-temporal form differs from the plain form in {kname}"


def test_bn_passes_on_a_tensor_beyond_the_nontemporal_threshold(dev):
    """default knobs on the batch-64 activation of the 160-pixel layers (64 x 160 x 160 x 128 fp16 = 419 MB >= bn_nt_bytes): statistics,
    normalisation + SiLU and the backward reduce / apply run in the forms the benchmarked train step runs them in (non-temporal loads and
    stores, uncapped grids, two-level partial sums) and match a chunked fp32 torch evaluation."""
    import ctypes as C

    _lib, ops = _ops()
    L = _lib.lib()
    assert ops.tune_get("bn_nt_bytes") == 64 << 20   # (round 6: 64 MiB; the tensor below is beyond either value)
    n, h, w, c = 64, 160, 160, 128
    dtype = torch.float16
    M = n * h * w
    g = torch.Generator(device=dev).manual_seed(9)
    uv, yv, dyv, duv = (ops.View.alloc(n, h, w, c, dtype, dev) for _ in range(4))
    assert uv.buf.numel() * 2 >= 128 << 20
    uv.buf.normal_(generator=g).add_(0.1)
    dyv.buf.normal_(generator=g)
    gamma = torch.rand(c, device=dev, generator=g) + 0.5
    beta = torch.randn(c, device=dev, generator=g) * 0.3
    sums = ops.bn_scratch(c, dev)
    scale, shift, mean, invstd, dgamma, dbeta = (torch.empty(c, device=dev) for _ in range(6))
    ut, yt, dyt, dut = uv.y3(), yv.y3(), dyv.y3(), duv.y3()
    dc, st = ops.dtype_code(dtype), ops.stream_ptr()
    _lib.check(L.y3_bn_stats_finalize(C.byref(ut), dc, sums.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-3, 0.03, None, None, scale.data_ptr(), shift.data_ptr(),
                                      mean.data_ptr(), invstd.data_ptr(), st), "y3_bn_stats_finalize")
    _lib.check(L.y3_bn_act_fwd(C.byref(ut), scale.data_ptr(), shift.data_ptr(), None, C.byref(yt), dc, _lib.Y3_ACT_SILU, st), "y3_bn_act_fwd")
    _lib.check(L.y3_bn_act_bwd(C.byref(ut), C.byref(dyt), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(), dc, _lib.Y3_ACT_SILU, sums.data_ptr(), C.byref(dut),
                               dgamma.data_ptr(), dbeta.data_ptr(), st), "y3_bn_act_bwd")
    torch.cuda.synchronize()
    U, DY = uv.buf.view(M, c), dyv.buf.view(M, c)
    s0 = torch.zeros(c, dtype=torch.float64, device=dev)
    s1 = torch.zeros(c, dtype=torch.float64, device=dev)
    CH = 1 << 18
    for i in range(0, M, CH):
        b = U[i : i + CH].double()
        s0 += b.sum(0)
        s1 += (b * b).sum(0)
    mr = s0 / M
    vr = s1 / M - mr * mr
    isr = 1 / torch.sqrt(vr + 1e-3)
    assert (mean.double() - mr).abs().max().item() <= 1e-6 and (invstd.double() - isr).abs().max().item() <= 1e-5 * isr.max().item()
    sg = torch.zeros(c, dtype=torch.float64, device=dev)
    sgx = torch.zeros(c, dtype=torch.float64, device=dev)
    ymax = 0.0
    for i in range(0, M, CH):
        xh = (U[i : i + CH].float() - mr.float()) * isr.float()
        z = xh * gamma + beta
        sgm = torch.sigmoid(z)
        yref = z * sgm
        ymax = max(ymax, (yv.buf.view(M, c)[i : i + CH].float() - yref).abs().max().item() / max(1.0, yref.abs().max().item()))
        gz = DY[i : i + CH].float() * (sgm + z * sgm * (1 - sgm))
        sg += gz.double().sum(0)
        sgx += (gz * xh).double().sum(0)
    assert ymax <= 2.0 ** -10, f"bn_act_fwd: {ymax:.3e}"
    assert (dbeta.double() - sg).abs().max().item() <= 1e-4 * sg.abs().max().item() + 1e-2
    assert (dgamma.double() - sgx).abs().max().item() <= 1e-4 * sgx.abs().max().item() + 1e-2
    dmax = 0.0
    for i in range(0, M, CH):
        xh = (U[i : i + CH].float() - mr.float()) * isr.float()
        z = xh * gamma + beta
        sgm = torch.sigmoid(z)
        gz = DY[i : i + CH].float() * (sgm + z * sgm * (1 - sgm))
        dref = (gamma * isr.float()) * (gz - (sg / M).float() - xh * (sgx / M).float())
        dmax = max(dmax, (duv.buf.view(M, c)[i : i + CH].float() - dref).abs().max().item() / dref.abs().max().item())
    assert dmax <= 1.5 * 2.0 ** -10, f"bn_act_bwd: {dmax:.3e}"


@pytest.mark.parametrize("adt,force_v10", [(torch.float16, 0), (torch.bfloat16, 0), (torch.float16, 1)], ids=["fp16", "bf16", "fp16_every_3x3_on_v10"])
def test_train_step_640_autocast_vs_oracle_autograd(dev, tune, monkeypatch, adt, force_v10):
    """BASELINE configs[2] resolution: one autocast training step of yolov3 at 640 x 640, batch 4 -- 1.6 M / 409 600 / ... / 1 600 pixel maps,
    i.e. the fused stem backward, the 256-tile filter gradients with many pixel slices, the K-split form with statistics rows, the one-launch stride-2 data
    gradients and the two-level statistics sums at real map sizes -- against torch autograd over the fp32 CPU oracle: loss, and the
    direction and norm of every large parameter gradient."""
    from yolov3_amd import ComputeLoss

    nc, bs, hw = 80, 4, 640
    hyp = dict(box=0.05, cls=0.5, cls_pw=1.0, obj=1.0, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0, label_smoothing=0.0)
    m, (layers, save, sd, strides) = build_pair("yolov3", nc, 23, dev, torch.float32)
    m.train()
    m.hyp = hyp
    x = torch.rand(bs, 3, hw, hw, generator=torch.Generator().manual_seed(12))
    tg = yo.synth_targets(bs, nc, seed=13)
    sdg = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items()}
    raws_ref = yo.forward(layers, save, sdg, x, strides, training=True)
    loss_ref, _, _ = yo.compute_loss(raws_ref, tg, sd[[k for k in sd if k.endswith("anchors")][0]], hyp, nc)
    loss_ref.backward()
    # which conv kernels the step ran: at batch 4 the dispatcher gives the 80 x 80 maps to v10h and the 40 x 40 / 20 x 20 maps (below a quarter round of tiles) to
    # the K-split form v10k; the third parametrisation forces every eligible 3 x 3 launch -- forward with statistics AND data gradient -- onto v10 / v10h, the kernels the
    # batch-64 benchmark runs there
    _lib, ops = _ops()
    if force_v10:
        tune("conv_v10", 2)
    seen = set()
    for fn in ("conv2d", "conv2d_stats"):
        orig = getattr(ops, fn)

        def wrapped(*a, __orig=orig, **kw):
            r = __orig(*a, **kw)
            seen.add(ops.last_conv_variant())
            return r

        monkeypatch.setattr(ops, fn, wrapped)
    import copy

    m_off = copy.deepcopy(m) if adt == torch.bfloat16 else None   # the same weights for the A/B arm without conv_v10.h (its own plan: the statistics rows differ by variant)

    def run_step(model):
        with torch.autocast("cuda", dtype=adt):
            raws = model(x.to(dev))
            loss_, _ = ComputeLoss(model)(raws, tg.to(dev))
        (loss_ * 128.0).backward()
        torch.cuda.synchronize()
        cmin, wk, nworst = 1.0, None, (0.0, None)
        dot = n_hip = n_ref = 0.0   # the whole gradient as one vector
        for k, p_ in model.named_parameters():
            ref = sdg[k].grad
            assert p_.grad is not None and torch.isfinite(p_.grad).all(), k
            if ref is not None:
                gd, rd = p_.grad.double().cpu().flatten() / 128.0, ref.double().flatten()
                dot += float(gd @ rd); n_hip += float(gd @ gd); n_ref += float(rd @ rd)
            if ref is None or ref.numel() < 4096:
                continue
            gq = p_.grad.float().cpu() / 128.0
            cq = torch.nn.functional.cosine_similarity(gq.flatten(), ref.flatten(), dim=0).item()
            nr = abs(gq.norm().item() / ref.norm().item() - 1.0)
            if cq < cmin:
                cmin, wk = cq, k
            if nr > nworst[0]:
                nworst = (nr, k)
        return loss_, cmin, wk, nworst, dot / math.sqrt(n_hip * n_ref)

    loss, cos_min, worst, norm_worst, cos_all = run_step(m)
    assert "v10h" in seen and ("v10" in seen) == bool(force_v10) and ("v10k" in seen) != bool(force_v10), seen
    rel = abs(loss.item() - loss_ref.item()) / loss_ref.item()
    if m_off is not None:
        # same box, same weights, same batch: the step with every launch of conv_v10.h handed back to the tile kernels it replaced.  The absolute bf16 bounds below sit
        # near the floor of the model; THIS is the assert that tracks the kernel: v10 / v10h / v10k may not be further from the fp32 gradient than the v6 / v3 path
        # by more than the summation-order noise (round-4 advisor finding)
        m_off.hyp = hyp
        m_off.train()
        tune("conv_v10", 0)
        seen.clear()
        _, cos_min_off, worst_off, _, cos_all_off = run_step(m_off)
        assert not (seen & {"v10", "v10h", "v10k"}), seen
        print(f"[train 640 {adt}] without conv_v10: whole-gradient cosine {cos_all_off:.5f}, min per-tensor cosine {cos_min_off:.4f} at {worst_off}")
        assert cos_all >= cos_all_off - 0.01, f"whole-gradient cosine {cos_all:.5f} with conv_v10, {cos_all_off:.5f} without"
        assert cos_min >= cos_min_off - 0.03, f"min per-tensor cosine {cos_min:.4f} with conv_v10, {cos_min_off:.4f} without"
    print(f"[train 640 {adt}] loss rel err {rel:.2e}, whole-gradient cosine {cos_all:.5f}, min per-tensor cosine {cos_min:.4f} at {worst}, worst norm ratio error {norm_worst[0]:.3f} at {norm_worst[1]}")
    # measured (MI355X, rounds 3 / 4): fp16 loss 1e-7 .. 6e-6, whole-gradient cosine 0.9973, min per-tensor cosine 0.9947 .. 0.9951, worst norm error 0.5 %;
    # bf16 loss 1e-5 .. 5e-5, whole-gradient cosine 0.941, min per-tensor cosine 0.922 .. 0.938 (0.967 at 128 x 128), norm 2.1 .. 3.3 %.
    # Where the bf16 bound comes from: 1 - cos = noise^2 / (2 signal^2).  If EVERY rounding error of the step scaled with the mantissa (2^-8 against 2^-11: x 64
    # in noise^2), the fp16 figure 1 - 0.9951 would put bf16 at 1 / sqrt(1 + 64 * 2 * 0.0049) = 0.78: that is the floor of the model.  The hardware sits above
    # it (0.92 .. 0.97: the fp32 accumulations and the fp64 statistics do not scale), and moves by ~0.01-0.02 between builds that differ only in summation order
    # (the round-4 kernel moved the minimum from 0.938 to 0.922).  Asserted: per tensor 0.88 (between the derived floor and the lowest value seen), and the
    # whole gradient -- what the optimizer step follows -- 0.92 / 0.995
    assert rel < (1e-4 if adt == torch.float16 else 1e-3)
    assert cos_min > (0.99 if adt == torch.float16 else 0.88), f"gradient direction: cosine {cos_min:.4f} at {worst}"
    assert cos_all > (0.995 if adt == torch.float16 else 0.92), f"whole-gradient cosine {cos_all:.5f}"
    assert norm_worst[0] < (0.02 if adt == torch.float16 else 0.06), norm_worst


# ------------------------------------------------------------------------------------------------ conv v10 (persistent, register-resident filter fragments)
V10_CASES = [
    # name, (n,h,w,cin,cout,k,s), kwargs, knobs (v10_mp cap, v10_blocks per filter tile; 0 = the host's plan)
    ("plan_20x20_res", (8, 20, 20, 256, 512, 3, 1), {"residual": True}, (0, 0)),                  # 100 column blocks: 16 blocks per filter tile, runs of 6 / 7
    ("two_blocks_walk_5_tiles", (6, 20, 20, 64, 256, 3, 1), {}, (0, 2)),                          # runs of 38 / 37: bodies 8 / 7, tiles cross rows and images, 2 channel blocks
    ("tiny_images_one_cb", (40, 7, 5, 32, 256, 3, 1), {"residual": True}, (7, 3)),               # 35-pixel images, ncb = 1 (every channel block is a tile's last), ragged tail
    ("odd_cb_parity", (3, 21, 19, 96, 256, 3, 1), {"sliced": True}, (6, 2)),                      # 3 channel blocks: tiles start in alternating patch buffers
    ("w80_two_requests", (2, 80, 80, 64, 256, 3, 1), {}, (7, 0)),                                 # 36+ patch pieces: two request slots per tap
    ("one_column_block_tiles", (1, 40, 40, 128, 512, 3, 1), {"act": False}, (0, 50)),             # 32 valid pixels in a 192-pixel body
    ("one_block_walks_all", (2, 13, 26, 160, 256, 3, 1), {}, (0, 1)),                             # 22 column blocks (the last with 4 pixels) in one block: 8 + 7 + 7, 5 channel blocks
    ("w160_mp6", (1, 160, 160, 32, 256, 3, 1), {"residual": True}, (6, 0)),
    ("four_filter_tiles", (4, 20, 20, 128, 1024, 3, 1), {"residual": True}, (0, 0)),
    ("mixed_7_6_runs", (13, 20, 20, 64, 512, 3, 1), {}, (0, 25)),                                 # 163 column blocks over 25 blocks: runs of 7 / 6 -> both bodies in one launch
]


@pytest.mark.parametrize("half", [0, 1], ids=["one_block_per_cu", "two_half_blocks_per_cu"])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("name,shape,kw,knobs", V10_CASES, ids=[c[0] for c in V10_CASES])
def test_conv_v10_vs_fp32_reference(dev, tune, dtype, name, shape, kw, knobs, half):
    """conv_v10.h (persistent blocks over 32-pixel column blocks, bodies of 6 / 7 / 8 column blocks, filter fragments by register loads from the
    fragment-ordered copy of the bank, the next tile's first patch requested under the last channel block) against fp32 conv2d on the same rounded
    operands: tiles that cross rows and image boundaries, edge taps, 1 .. 8 channel blocks (odd counts flip the patch-buffer parity from tile to
    tile), one and two request slots per tap, single-column-block tiles, ragged tails, residual / sliced outputs; repeated launches bit-identical."""
    mp, blocks = knobs
    tune("conv_v10", 2)
    tune("v10_half", half)   # 1: two blocks per CU with bodies of 3 / 4 column blocks and 32-pixel epilogue passes (the same kernel source, other geometry)
    tune("v10_mp", {0: 0, 6: 3, 7: 4, 8: 4}[mp] if half else mp)
    tune("v10_blocks", blocks)
    want = "v10h" if half and not (name == "w160_mp6") else "v10"   # (160-pixel rows: the halo patch of even 3 column blocks exceeds a half block's 31 KiB buffer)
    out, ref = run_conv(dev, dtype, *shape, algo=1, ws=True, expect=want, repeat=2, **kw)
    _conv_tol_check(name, dtype, out, ref)


@pytest.mark.parametrize("half", [0, 1])
def test_conv_v10_statistics_rows(dev, tune, half):
    """BatchNorm statistics rows from the v10 epilogue (four rows per tile: one per 64-pixel pass, zero rows for the passes a narrower body does not
    have; only valid pixels counted): their fp64 sum equals the statistics of the stored tensor."""
    _lib, ops = _ops()
    tune("conv_v10", 2)
    tune("v10_blocks", 3)
    tune("v10_half", half)
    n, h, w, cin, cout, k, s = 5, 20, 20, 64, 256, 3, 1
    dtype = torch.float16
    g = torch.Generator().manual_seed(4)
    x = torch.randn(n, cin, h, w, generator=g).to(dtype)
    wt = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
    xv = ops.View.alloc(n, h, w, cin, dtype, dev)
    ops.nchw_to_nhwc(x.to(dev), xv)
    filt = ops.pack_filter(wt.to(dev), cout, cin, dtype)
    zb = torch.zeros(cout, device=dev)
    y1 = ops.View.alloc(n, h, w, cout, dtype, dev)
    rows = ops.conv2d_stats_rows(xv, y1, k, s)
    assert rows == (18 if half else 9) * 4, rows   # 63 column blocks over 3 blocks: runs of 21 = 3 tiles of 7 (half-size blocks: 6 tiles of 4 / 3)
    buf = torch.full((rows * 2 * cout,), float("nan"), device=dev)
    assert ops.conv2d_stats(xv, filt, zb, y1, k, s, buf, rows) == rows and ops.last_conv_variant() == ("v10h" if half else "v10")
    torch.cuda.synchronize()
    u = y1.as_nhwc().double().cpu().reshape(-1, cout)
    tot = buf.view(rows, cout, 2).double().sum(0).cpu()
    assert torch.isfinite(tot).all(), "a statistics row was not written"
    assert (tot[:, 0] - u.sum(0)).abs().max().item() <= 1e-5 * u.abs().sum(0).max().item()
    assert (tot[:, 1] - (u * u).sum(0)).abs().max().item() <= 1e-5 * (u * u).sum(0).max().item()
    xr = x.float()
    ref = F.conv2d(xr, wt.to(dtype).float(), None, padding=1)
    assert (y1.as_nhwc().float().cpu().permute(0, 3, 1, 2) - ref).abs().max().item() < 2e-2


def test_pack_filter_fragment_copy(dev):
    """Every packer writes the fragment-ordered copy conv_v10.h reads behind the row-major bank of a 3x3 layer with rows % 256 == 0 and channels % 32 == 0
    (y3_frag_index): element (row, tap, channel) of the bank sits at ((((row / 64) nk + 9 cb + tap) 4 + 2 kk + a) 64 + 32 fk + row % 32) 8 + e."""
    _lib, ops = _ops()
    cout, cin = 256, 64
    g = torch.Generator().manual_seed(3)
    wt = torch.randn(cout, cin, 3, 3, generator=g)
    for dtype in (torch.float16, torch.bfloat16):
        rows_kpad = cout * 9 * cin
        assert ops.packed_filter_elems(cout, cin, 3) == 2 * rows_kpad and ops.packed_filter_elems(cout, cin, 1) == cout * cin and ops.packed_filter_elems(128, cin, 3) == 128 * 9 * cin
        bank = ops.pack_filter(wt.to(dev), cout, cin, dtype)
        std = bank[:rows_kpad].view(cout, 9, cin).cpu()
        assert torch.equal(std, wt.permute(0, 2, 3, 1).reshape(cout, 9, cin).to(dtype))
        nk = 9 * (cin // 32)
        frag = bank[rows_kpad:].view(cout // 64, nk, 2, 2, 2, 32, 8).cpu()   # [tile-wave][K-step = 9 cb + tap][kk][a][fk][row][e]
        want = std.view(cout // 64, 2, 32, 9, cin // 32, 2, 2, 8)            # [tw][a][row][tap][cb][kk][fk][e]
        want = want.permute(0, 4, 3, 5, 1, 6, 2, 7).reshape(cout // 64, nk, 2, 2, 2, 32, 8)
        assert torch.equal(frag, want)
        f2, d2 = ops.pack_filter_pair(wt.to(dev), cout, cin, dtype)
        assert torch.equal(f2.view(torch.int16), bank.view(torch.int16))
        assert torch.equal(d2.view(torch.int16), ops.pack_filter_dgrad(wt.to(dev), cout, cin, dtype).view(torch.int16))
        # the data-gradient bank of a 256 -> 256 layer carries the copy too (rows = cin)
        w2 = torch.randn(256, 256, 3, 3, generator=g)
        dg = ops.pack_filter_dgrad(w2.to(dev), 256, 256, dtype)
        assert dg.numel() == 2 * 256 * 9 * 256
        dstd = dg[: 256 * 9 * 256].view(256, 9, 256).cpu()
        assert torch.equal(dstd, w2.flip(2, 3).permute(1, 2, 3, 0).reshape(256, 9, 256).to(dtype))
        dfrag = dg[256 * 9 * 256 :].view(4, 72, 2, 2, 2, 32, 8).cpu()
        dwant = dstd.view(4, 2, 32, 9, 8, 2, 2, 8).permute(0, 4, 3, 5, 1, 6, 2, 7).reshape(4, 72, 2, 2, 2, 32, 8)
        assert torch.equal(dfrag, dwant)


# ------------------------------------------------------------------------------------------------ conv strip (register-resident filters, row ring)
STRIP_CONV_CASES = [
    # name, (n,h,w,cin,cout,k,s), kwargs, output rows per block (knob conv_strip; 2 = the host's plan)
    ("c64_32_one_strip", (2, 20, 64, 64, 32, 3, 1), {}, 2),
    ("c64_32_ragged_walk7", (3, 19, 150, 64, 32, 3, 1), {"act": False}, 7),           # 3 strips, the last 22 pixels wide; blocks cross strips and images
    ("c64_32_dgrad_shape_walk5", (2, 33, 100, 64, 32, 3, 1), {"act": False}, 5),      # the data gradient of a 32 -> 64 layer
    ("c64_32_sliced", (2, 17, 70, 64, 32, 3, 1), {"sliced": True}, 4),
    ("c64_128_walk9", (2, 24, 96, 64, 128, 3, 1), {}, 9),
    ("c64_128_tall_one_block", (1, 70, 64, 64, 128, 3, 1), {"act": False}, 70),       # one block walks a whole strip: the row ring wraps many times
    ("c64_32_narrow_map", (4, 40, 13, 64, 32, 3, 1), {}, 3),                          # map narrower than a strip
    ("c128_64_ksplit_walk6", (2, 30, 100, 128, 64, 3, 1), {"act": False}, 6),         # the data gradient of a 64 -> 128 layer: two waves split the 72 reduction steps
    ("c128_64_ksplit_odd_rows", (3, 9, 64, 128, 64, 3, 1), {}, 5),                    # odd rows per block: the giver / taker roles of a pair differ from block to block
    ("c128_64_ksplit_sliced_one_row", (1, 11, 70, 128, 64, 3, 1), {"sliced": True}, 3),
    ("c64_128_s2_walk5", (2, 40, 200, 64, 128, 3, 2), {}, 5),                         # stride 2: two new input rows per output row, even / odd pixel halves
    ("c64_128_s2_odd_sizes", (3, 37, 131, 64, 128, 3, 2), {"act": False}, 4),         # odd input sizes: the last input row / column only under some taps
    ("c64_128_s2_tall_sliced", (1, 150, 128, 64, 128, 3, 2), {"sliced": True}, 75),   # one block walks a whole strip: the five-slot ring wraps many times
]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("name,shape,kw,per", STRIP_CONV_CASES, ids=[c[0] for c in STRIP_CONV_CASES])
def test_conv_strip_vs_fp32_reference(dev, tune, dtype, name, shape, kw, per):
    """conv_strip.h (filters in registers, one new input row per output row of a 64-pixel column strip, nine taps as views of three resident rows, two
    waves splitting the reduction at Cin = 128) against fp32 conv2d on the same rounded operands -- strips and images crossed inside a block, ragged
    last strips, edge taps, bias + SiLU, sliced outputs; repeated launches bit-identical; and against the tile kernels it replaces."""
    tune("conv_strip", per)
    out, ref = run_conv(dev, dtype, *shape, algo=1, ws=True, expect="strip", repeat=2, **kw)
    _conv_tol_check(name, dtype, out, ref)
    tune("conv_strip", 0)
    old, _ = run_conv(dev, dtype, *shape, algo=1, ws=True, **kw)
    assert (out - old).abs().max().item() <= 2e-2 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("cin,cout", [(64, 32), (128, 64)])
def test_conv_strip_statistics_rows(dev, tune, cin, cout):
    """BatchNorm statistics from the strip kernel: one row per block, pixel tile and K-split wave (accumulated in registers over the block's output rows),
    their fp64 sum equals the statistics of the stored tensor; pixels beyond the ragged last strip are not counted"""
    _lib, ops = _ops()
    tune("conv_strip", 6)
    n, h, w, k, s = 3, 21, 150, 3, 1
    dtype = torch.float16
    g = torch.Generator().manual_seed(4)
    x = torch.randn(n, cin, h, w, generator=g).to(dtype)
    wt = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
    xv = ops.View.alloc(n, h, w, cin, dtype, dev)
    ops.nchw_to_nhwc(x.to(dev), xv)
    filt = ops.pack_filter(wt.to(dev), cout, cin, dtype)
    zb = torch.zeros(cout, device=dev)
    y1 = ops.View.alloc(n, h, w, cout, dtype, dev)
    rows = ops.conv2d_stats_rows(xv, y1, k, s)
    blocks = -(-(n * 3 * h) // 6)
    assert rows == blocks * 2 * (2 if cin == 128 else 1), (rows, blocks)
    buf = torch.full((rows * 2 * cout,), float("nan"), device=dev)
    assert ops.conv2d_stats(xv, filt, zb, y1, k, s, buf, rows) == rows and ops.last_conv_variant() == "strip"
    torch.cuda.synchronize()
    u = y1.as_nhwc().double().cpu().reshape(-1, cout)
    tot = buf.view(rows, cout, 2).double().sum(0).cpu()
    assert torch.isfinite(tot).all(), "a statistics row was not written"
    assert (tot[:, 0] - u.sum(0)).abs().max().item() <= 1e-5 * u.abs().sum(0).max().item()
    assert (tot[:, 1] - (u * u).sum(0)).abs().max().item() <= 1e-5 * (u * u).sum(0).max().item()


@pytest.mark.parametrize("name,dtype", [("yolov3", torch.float32), ("yolov3-tiny", torch.float16)])
def test_loss_autobalance_vs_oracle(dev, name, dtype):
    """ComputeLoss(autobalance=True) (reference utils/loss.py:121, :171-175; the oracle's restatement is pinned to the unmodified reference by
    tests/test_oracle_vs_reference_live.py): three consecutive calls -- every loss formed with the weights the previous calls left, the weights
    after each call equal to the oracle's (one read-back of nl floats per call: y3_loss_level_obj)."""
    hyp = dict(box=0.05, cls=0.5, cls_pw=1.0, obj=1.0, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0, label_smoothing=0.0)
    from yolov3_amd import ComputeLoss

    nc, hw, bs = 20, 128, 3
    m, (layers, save, sd, strides) = build_pair(name, nc, 5, dev, torch.float32)
    m.hyp = hyp
    crit = ComputeLoss(m, autobalance=True)
    ssi = [int(s) for s in strides].index(16)
    assert crit.ssi == ssi
    balance = {3: [4.0, 1.0, 0.4]}.get(len(strides), [4.0, 1.0, 0.25, 0.06, 0.02])[: len(strides)]
    balance = list(balance)
    anchors = sd[[k for k in sd if k.endswith("anchors")][0]]
    shapes = [(bs, 3, hw // int(s), hw // int(s), nc + 5) for s in strides]
    tol = 1e-4 if dtype == torch.float32 else 3e-3
    for step in range(3):
        tg = yo.synth_targets(bs, nc, seed=90 + step)
        p = yo.synth_raw_predictions(shapes, seed=60 + step)
        pq = [t.to(dtype) for t in p]
        loss_ref, _, _ = yo.compute_loss([t.float() for t in pq], tg, anchors, hyp, nc, balance=balance, autobalance_ssi=ssi)
        loss, _ = crit([t.to(dev) for t in pq], tg.to(dev))
        assert abs(loss.item() - loss_ref.item()) <= tol * abs(loss_ref.item()), (step, loss.item(), loss_ref.item())
        for a, b in zip(crit.balance, balance):
            assert abs(a - b) <= tol * abs(b), (step, crit.balance, balance)
    assert abs(crit.balance[ssi] - 1.0) < 1e-6


# ---------------------------------------------------------------------------------------------- test-time augmentation (models/yolo.py:239-276)
TTA_KEYS = ["yolov3-tiny-nc20-96x160-bs2", "yolov3-nc7-128x96-bs1"]


def _tta_case(key):
    name, nc, hw, bs = key.rsplit("-", 3)
    h, w = (int(v) for v in hw.split("x"))
    return name, int(nc[2:]), h, w, int(bs[2:])


@pytest.mark.parametrize("key", TTA_KEYS)
def test_scale_img_vs_reference_golden(dev, golden_dir, key):
    """y3_scale_img (mirror + bilinear resize + 0.447 padding in one kernel) against the tensors the reference's scale_img call produced on the CPU
    (F.interpolate align_corners=False: 2 ulp of fp32), then in fp16 / bf16 against the rounded fp32 result."""
    from yolov3_amd import ops
    from oracle import upstream

    gold = torch.load(golden_dir / "tta.pt")[key]
    name, nc, h, w, bs = _tta_case(key)
    x = torch.rand(bs, 3, h, w, generator=torch.Generator().manual_seed(8))
    assert checksum(x) == gold["x_sum"]
    a = ops.scale_img(x.to(dev), 0.83, gs=32, flip_lr=True)
    b = ops.scale_img(x.to(dev), 0.67, gs=32)
    assert a.shape == gold["x_083_flip"].shape and b.shape == gold["x_067"].shape
    torch.testing.assert_close(a.cpu(), gold["x_083_flip"], rtol=0, atol=2.5e-7)
    torch.testing.assert_close(b.cpu(), gold["x_067"], rtol=0, atol=2.5e-7)
    xd = x.to(dev)
    assert ops.scale_img(xd, 1.0) is xd
    torch.testing.assert_close(ops.scale_img(xd, 1.0, flip_lr=True).cpu(), x.flip(3), rtol=0, atol=0)   # mirror only: exact
    for dt, ulp in ((torch.float16, 2.0**-10), (torch.bfloat16, 2.0**-7)):
        xh = x.to(dt)
        ref = upstream.scale_img(xh.float().flip(3), 0.83, gs=32)
        got = ops.scale_img(xh.to(dev), 0.83, gs=32, flip_lr=True)
        assert got.dtype == dt
        err = (got.float().cpu() - ref).abs().max().item()
        assert err <= ulp, (dt, err)   # values in [0, 1): at most one unit in the last place of T


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_descale_pred_bit_exact(dev, dtype):
    """y3_descale_pred against the reference's in-place tensor ops (models/yolo.py:253-261: `p[..., :4] /= scale`, `p[..., 0] = w - p[..., 0]`) run by
    torch on the CPU in the same dtype, row windows included: bit-exact"""
    from yolov3_amd import ops

    g = torch.Generator().manual_seed(77)
    bs, rows, no = 3, 1000, 25
    p = (torch.rand(bs, rows, no, generator=g) * 640).to(dtype)
    for scale, flip, lo, n, off in [(0.83, 3, 0, rows, 5), (0.67, None, 160, rows - 160, 0), (1, None, 0, rows - 40, 0), (0.83, 2, 7, 100, 3)]:
        ref = p.clone()
        ref[..., :4] /= scale
        if flip == 2:
            ref[..., 1] = 480 - ref[..., 1]
        elif flip == 3:
            ref[..., 0] = 640 - ref[..., 0]
        out = torch.full((bs, off + n + 2, no), -1.0, dtype=dtype, device=dev)
        ops.descale_pred_into(p.to(dev), lo, n, scale, flip, (480, 640), out, off)
        got = out.cpu()
        assert torch.equal(got[:, off : off + n].view(torch.int16 if dtype != torch.float32 else torch.int32), ref[:, lo : lo + n].contiguous().view(torch.int16 if dtype != torch.float32 else torch.int32))
        assert (got[:, :off] == -1).all() and (got[:, off + n :] == -1).all()   # nothing outside the window


@pytest.mark.parametrize("key", TTA_KEYS)
def test_augmented_forward_vs_reference_golden(dev, golden_dir, key):
    """model(x, augment=True) (fp32 engine) against the UNMODIFIED reference's augmented prediction (tests/golden/tta.pt): shape (the clipped row
    windows) and values at the whole-model tolerance; the call returns (pred, None) like the reference"""
    gold = torch.load(golden_dir / "tta.pt")[key]
    name, nc, h, w, bs = _tta_case(key)
    m, _ = build_pair(name, nc, 13, dev, torch.float32)
    x = torch.rand(bs, 3, h, w, generator=torch.Generator().manual_seed(8))
    assert checksum(x) == gold["x_sum"]
    pred, none = m(x.to(dev), augment=True)
    assert none is None and pred.shape == gold["pred"].shape
    torch.testing.assert_close(pred.cpu(), gold["pred"], rtol=1e-4, atol=2e-4)
    m.train()
    with pytest.raises(RuntimeError):
        m(x.to(dev), augment=True)


def test_augmented_forward_half_vs_oracle(dev):
    """fp16 engine, yolov3 at 256 x 320, batch 2: the augmented prediction against the fp32 oracle's (same weights) -- confidences and classes
    absolutely, boxes relative to the image size (half-precision storage through 75 layers, as test_model_half_vs_fp32_oracle)"""
    m, (layers, save, sd, strides) = build_pair("yolov3", 80, 21, dev, torch.float16)
    x = torch.rand(2, 3, 256, 320, generator=torch.Generator().manual_seed(9)).half()
    pred, _ = m(x.to(dev), augment=True)
    with torch.no_grad():
        ref = yo.forward_augment(layers, save, sd, x.float(), strides)
    assert pred.shape == ref.shape and pred.dtype == torch.float16
    d = (pred.float().cpu() - ref).abs()
    assert d[..., 4:].max().item() < 2e-2
    assert (d[..., :4] / (ref[..., :4].abs() + 32.0)).max().item() < 3e-2


@pytest.mark.parametrize("backend", ["nccl", "gloo"])
def test_sync_batchnorm_two_ranks(tmp_path, backend):
    """--sync-bn (reference train.py:270-272: torch.nn.SyncBatchNorm.convert_sync_batchnorm(model) before DDP).  Two ranks, half of a batch each, against
    ONE process running the whole batch through plain BatchNorm with the same weights: with statistics taken over both ranks the activations of a
    rank's half, the running statistics and -- for a loss that is a sum over images -- world x the averaged parameter gradients are those of the
    whole-batch run.  (Per-rank statistics would differ in the second digit: the check at the end.)"""
    import subprocess
    import sys

    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (the round-end 8-GPU node; the 1-GPU test box runs the gloo form)")
    script = tmp_path / "sync_bn.py"
    script.write_text(f"""
import copy, sys, torch
sys.path.insert(0, {str(ROOT)!r})
from yolov3_amd import DetectionModel, parallel
rank, local_rank, world = parallel.init({backend!r})
dev = parallel.local_device(local_rank)
torch.cuda.set_device(dev)
torch.manual_seed(0)
full = DetectionModel("yolov3-tiny.yaml", nc=20).to(dev).train()
parallel.broadcast_parameters(full)
sync = torch.nn.SyncBatchNorm.convert_sync_batchnorm(copy.deepcopy(full)).to(dev).train()
assert sum(isinstance(q, torch.nn.SyncBatchNorm) for q in sync.modules()) == 11
g = torch.Generator().manual_seed(3)
x = torch.rand(8, 3, 96, 128, generator=g).to(dev)
wts = [torch.randn(8, 3, 96 // s, 128 // s, 25, generator=g).to(dev) for s in (16, 32)]
lo, hi = rank * 4, rank * 4 + 4
def run(model, xs, ws, exchange):
    model.grad_sync = parallel.GradBuckets(bucket_bytes=4 << 20) if exchange else None
    model.zero_grad(set_to_none=True)
    out = model(xs)
    sum((o * w).sum() for o, w in zip(out, ws)).backward()
    torch.cuda.synchronize()
    return [o.detach() for o in out], torch.cat([q.grad.flatten() for q in model.parameters()])
out_f, g_f = run(full, x, wts, False)
out_s, g_s = run(sync, x[lo:hi], [w[lo:hi] for w in wts], True)
def close(a, b, tol, what):
    err, ref = float((a - b).abs().max()), float(b.abs().max())
    assert err <= tol * ref, (what, err, ref)
for a, b in zip(out_s, out_f):
    close(a, b[lo:hi], 2e-5, "activations")
close(g_s * world, g_f, 2e-4, "gradients")
bufs_f, bufs_s = dict(full.named_buffers()), dict(sync.named_buffers())
for k, v in bufs_f.items():
    if k.endswith("running_mean") or k.endswith("running_var"):
        close(bufs_s[k], v, 1e-5, k)
    if k.endswith("num_batches_tracked"):
        assert int(bufs_s[k]) == int(v) == 1
# and the statistics really were global: the same half batch through per-rank BatchNorm gives something else
own = copy.deepcopy(full)
own.load_state_dict(sync.state_dict(), strict=False)
out_o, _ = run(own, x[lo:hi], [w[lo:hi] for w in wts], False)
assert float((out_o[0] - out_f[0][lo:hi]).abs().max()) > 1e-3 * float(out_f[0].abs().max())
print("rank", rank, "ok")
parallel.finalize()
""")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29537", str(script)],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.count("ok") == 2, out.stdout + out.stderr
