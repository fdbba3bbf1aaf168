"""bench.py -- images/sec of the YOLOv3 inference hot path (forward + Detect decode + batched NMS) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload = BASELINE.json configs[1]: yolov3, 640x640, batch 32 per GPU, fp16, + NMS with val.py's settings
(conf 0.001, iou 0.6, multi_label, max_det 300 -- reference val.py:374-376).  One "step" = one pass of the hot
path over one batch resident in HBM: DetectionModel.forward (75 fused conv launches + decode) on synthetic
images, then non_max_suppression on THAT forward's prediction tensor -- one data-dependent pipeline.  The model has
random weights; its Detect head is calibrated once, before any timing, so that its own output is the NMS load of
SURVEY.md 8(d) (~5 k of the 25 200 rows per image above conf 0.001, ~12 k (row, class) candidates under multi_label:
calibrate_detect_head -- a random-weight head has objectness ~0.003 everywhere, i.e. no candidates at all).  The same
schedule with the NMS leg on the seeded synthetic prediction tensor of SURVEY 8(d) (rounds 1-4's headline) is kept as
the secondary key `synthetic_nms_tensor`.

Multi-GPU: inference does not exchange data -> N independent replicas, one process per GPU,
"weak" scaling; the only collective is the timing barrier / max-reduce.

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline` and `cpu_baseline`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

MFMA_PEAK_TFLOPS = 2500.0  # dense fp16/bf16, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


# launch geometries of ONE kernel source are one roofline group: conv_v10.h runs as "v10" (one block per CU), "v10h" (two half-size blocks per CU) and "v10k" (K split
# for small launches); `roofline.forms` keeps them apart
KERNEL_FAMILY = {"v10h": "v10", "v10k": "v10"}


def per_kernel_times(plan, reps=5, forms=None):
    """HIP-event timing of every launch of the compiled plan on the stream the kernels run on (torch's current
    stream).  Returns {kernel group: [flops, bytes, seconds, launches]} averaged over `reps` passes; `forms` (a dict) receives the same per dispatched variant name."""
    from yolov3_amd import ops

    stream = ops.stream_ptr()
    acc = {}
    for _ in range(reps):
        evs = []
        for ln in plan.launches:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ln.fn(*plan.resolved_args(ln), stream)
            e1.record()
            evs.append((ln, e0, e1))
        torch.cuda.synchronize()
        for ln, e0, e1 in evs:
            if getattr(ln, "kernel", ""):
                key = ln.kernel + "/3x3"
            elif ln.flops:
                w = ln.keep[4]
                var = plan.conv_variant(ln)   # the variant name comes from the library's own dispatcher (y3_conv2d_fwd_variant)
                key = f"conv_igemm_{KERNEL_FAMILY.get(var, var)}/{w.k}x{w.k}"
                if forms is not None and (var in KERNEL_FAMILY or var in KERNEL_FAMILY.values()):
                    if True:
                        f = forms.setdefault(f"{var}/{w.k}x{w.k}", [0.0, 0.0, 0.0, 0])
                        f[0] += ln.flops; f[1] += ln.bytes; f[2] += e0.elapsed_time(e1) * 1e-3; f[3] += 1
            else:
                key = ln.label.split(".")[-1]
            a = acc.setdefault(key, [0.0, 0.0, 0.0, 0])
            a[0] += ln.flops
            a[1] += ln.bytes
            a[2] += e0.elapsed_time(e1) * 1e-3
            a[3] += 1
    return acc


def host_threads(cap=32):
    """Threads for the CPU leg: the cores this process may really use (affinity mask and cgroup quota, not
    os.cpu_count(), which reports the whole host), capped -- oversubscribed oneDNN convolutions get slower."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


def cpu_baseline(bs_sample=4):
    """The oracle (torch-CPU fp32 restatement of the reference, fused eval forward + NMS) on the host cores, on a
    bounded sample of the same workload.  kind="port": the reference itself needs /root/reference + stubs and
    cannot travel to the GPU box."""
    import yaml

    from oracle import yolo_oracle as yo

    torch.set_num_threads(host_threads())
    d = yaml.safe_load(open(ROOT / "yolov3_amd" / "cfg" / "yolov3.yaml"))
    layers, save, anchors, nc = yo.parse_cfg(d)
    strides = yo.model_strides(layers)
    sd = yo.fuse_state_dict(yo.seeded_state_dict(layers, nc, anchors, strides, seed=0))
    x = torch.rand(bs_sample, 3, 640, 640, generator=torch.Generator().manual_seed(0))
    pred_s = yo.synth_predictions(bs=bs_sample, n_rows=25200, nc=80, seed=2)
    with torch.inference_mode():
        yo.forward(layers, save, sd, x[:1], strides)  # warm-up
        best_f, best_n = 1e9, 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            yo.forward(layers, save, sd, x, strides)
            best_f = min(best_f, time.perf_counter() - t0)
            t0 = time.perf_counter()
            yo.non_max_suppression(pred_s, 0.001, 0.6, multi_label=True, max_det=300)
            best_n = min(best_n, time.perf_counter() - t0)
    return {
        "value": round(bs_sample / (best_f + best_n), 3),
        "unit": "images/sec",
        "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": f"{bs_sample} images 640x640 fp32 fused-eval forward ({best_f:.2f}s) + NMS val settings ({best_n:.2f}s), best of 2, oracle/yolo_oracle.py",
    }


def train_flops_per_image(model, hw):
    """conv FLOPs of one training step per image: forward + data gradient + filter gradient = 3 x the forward's
    2*Ho*Wo*Cout*Cin*k*k per layer, minus the data gradient of the first layer (the image needs none)."""
    from yolov3_amd.engine import graph_hw, _sources

    total, first = 0.0, None
    sizes = graph_hw(model, hw, hw)
    for i, m in enumerate(model.model):
        src = _sources(i, m.f)
        for mod in m.modules():
            if isinstance(mod, torch.nn.Conv2d):
                ho, wo = sizes[i] if not type(m).__name__ == "Detect" else (None, None)
                if ho is None:   # Detect: one 1x1 conv per level on that level's map
                    continue
                f = 2.0 * ho * wo * mod.out_channels * mod.in_channels * mod.kernel_size[0] * mod.kernel_size[1]
                total += f
                if first is None:
                    first = f
    det = model.model[-1]
    for lvl, conv in enumerate(det.m):
        ho, wo = sizes[det.f[lvl]]
        total += 2.0 * ho * wo * conv.out_channels * conv.in_channels
    return 3.0 * total - (first or 0.0)


def run_train(args, rank, world, dev, parallel, yo, batch, steps, warmup):
    """Data-parallel training step (BASELINE configs[2] shape per GPU): forward (batch-stat BN) + ComputeLoss + backward with
    the gradient all-reduce overlapped (parallel.GradBuckets; RCCL over xGMI) + the fused optimizer step.  Weak scaling.
    Returns the result record (all ranks run it; rank 0 reports)."""
    from yolov3_amd import ComputeLoss, DetectionModel

    bs, hw = batch, args.imgsz
    torch.manual_seed(0)
    model = DetectionModel(f"{args.model}.yaml", nc=args.nc).to(dev).train()
    model.hyp = dict(box=0.05, cls=0.5, cls_pw=1.0, obj=1.0, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0, label_smoothing=0.0)
    parallel.broadcast_parameters(model)
    if world > 1:
        model.grad_sync = parallel.GradBuckets()   # the collective of a bucket: Y3_GRAD_EXCHANGE = all_reduce (default) | direct (parallel.GradBuckets)
    sync_form = model.grad_sync.exchange if world > 1 else None
    crit = ComputeLoss(model)
    # the reference's optimizer step (train.py:414-422): unscale_ + clip_grad_norm_(10) + SGD(nesterov, 3 param groups) + EMA (rank 0),
    # here the fused 3-launch kernel; weight decay scaled by total batch / 64 (train.py:236-237)
    from yolov3_amd.optim import FusedSGD, GradScaler, ModelEMA, smart_param_groups

    opt = FusedSGD(smart_param_groups(model, 0.01, 5e-4 * bs * world / 64), momentum=0.937, nesterov=True)
    ema = ModelEMA(model) if rank == 0 else None
    scaler = GradScaler(init_scale=1024.0)   # train.py:345; dynamic scale, growth counter and found-inf flag stay on the device
    x = torch.rand(bs, 3, hw, hw, generator=torch.Generator().manual_seed(rank)).to(dev)
    tg = yo.synth_targets(bs, args.nc, seed=1 + rank).to(dev)

    def step():   # train.py:402-422
        with torch.autocast("cuda", dtype=torch.float16 if args.dtype == "fp16" else torch.bfloat16):
            loss, _ = crit(model(x), tg)
        scaler.scale(loss * world).backward()  # loss *= WORLD_SIZE (train.py:406): DDP averages, the reference wants the sum
        scaler.unscale_(opt)
        scaler.step(opt, max_norm=10.0, ema=ema)   # unscale + inf check + clip_grad_norm_(10) + SGD(nesterov) + EMA, fused
        scaler.update()
        opt.zero_grad(set_to_none=True)
        return loss

    for _ in range(warmup):
        step()
    parallel.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    parallel.barrier()
    dt = parallel.max_over_ranks(time.perf_counter() - t0, dev)
    group = parallel.describe(dev)   # (collective: every rank calls it) who took part, as the communicator sees it
    flops_img = train_flops_per_image(model, hw)
    tflops = flops_img * bs * steps / dt / 1e12   # per GPU
    # N = 1: what the gradient exchange costs the step when it runs beside the backward -- the same bucket / side-stream / event machinery as
    # N ranks on a ONE-rank RCCL communicator (the collective itself moves no bytes over xGMI at one rank: this prices the plumbing and shows
    # the side stream does not stall the compute stream; the wire time of N = 8 is modelled in DESIGN.md section 7)
    exchange = None
    if world == 1 and not os.environ.get("Y3_NO_EXCHANGE_LEG"):
        try:
            parallel.init(force=True)
            model.grad_sync = parallel.GradBuckets(force=True)
            step()
            parallel.barrier()
            t1 = time.perf_counter()
            for _ in range(steps):
                step()
            parallel.barrier()
            dt1 = time.perf_counter() - t1
            exchange = {"ms_per_step_with_exchange": round(dt1 / steps * 1e3, 3), "ms_per_step_without": round(dt / steps * 1e3, 3),
                        "gradient_bytes": int(sum(p.numel() for p in model.parameters()) * 4), "bucket_bytes": model.grad_sync.bucket_bytes,
                        "communicator": "RCCL, 1 rank (in-place ReduceOp.AVG on arena ranges, side stream, event-ordered against the producing kernels)"}
        except Exception as e:  # noqa: BLE001
            exchange = {"error": f"{type(e).__name__}: {e}"[:300]}
        finally:
            model.grad_sync = None
    # kernel families of ONE more step, measured here: HIP events around every C-ABI call of the step on the stream it runs on (_lib.CallTimer), after the timed steps
    groups, timed_step_ms = None, None
    from yolov3_amd import _lib as y3lib

    side_streams = []
    try:   # (every rank runs the extra step -- it contains the gradient collectives -- rank 0's record is the one reported)
        # the instrumented step runs on ONE stream: with the filter gradients on their side stream (the default since round 6) two full-chip kernels are time-sliced and
        # every call's event pair would measure the other stream's kernels too (the families then sum to more than the step)
        from yolov3_amd.engine import plan_cache

        for key, pl in plan_cache(model).plans.items():
            if key[0] == "train" and getattr(pl, "wgrad_stream", None) is not None:
                side_streams.append((pl, pl.wgrad_stream))
                pl.wgrad_stream = None
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        with y3lib.CallTimer() as ct:
            step()
        e1.record()
        by_fn = ct.by_function()
        timed_step_ms = round(e0.elapsed_time(e1), 3)
        groups = train_call_families(by_fn)
        lib_ms = sum(g["ms_per_step"] for g in groups.values())
        groups["outside the library (torch glue, gaps)"] = {"ms_per_step": round(timed_step_ms - lib_ms, 3), "calls_per_step": 0}
    except Exception as e:  # noqa: BLE001
        groups = {"error": f"{type(e).__name__}: {e}"[:300]}
    finally:
        for pl, st_ in side_streams:
            pl.wgrad_stream = st_
    parallel.barrier()
    rec = {
        "metric": "images/sec (640x640) train step", "value": round(world * bs * steps / dt, 2), "unit": "images/sec", "n_gpus": world,
        "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f16" if args.dtype == "fp16" else "bf16", "data": "synthetic (seeded uniform images, Poisson(7) targets/img; random-init weights)",
        "config": {"workload": f"{args.model} train step {hw}x{hw} batch={bs}/GPU autocast {args.dtype}: fwd (batch-stat BN) + ComputeLoss + bwd + grad all-reduce + fused unscale/clip/SGD-nesterov/EMA [BASELINE configs[2]]",
                   "global_batch": world * bs, "parallelism": f"dp{world} (bucketed gradient average overlapped with backward)", "gradient_exchange": sync_form},
        "final_loss": float(loss.detach()), "loss_scale": scaler.get_scale(), "gradient_exchange_1rank": exchange, "process_group": group,
        "roofline": {"bound": "mfma", "achieved": round(tflops, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tflops / MFMA_PEAK_TFLOPS, 4),
                     "whole_step_frac": round(tflops / MFMA_PEAK_TFLOPS, 4), "gflop_per_image": round(flops_img / 1e9, 2),
                     "note": "whole step (fwd + dgrad + wgrad conv FLOPs per GPU / step time)",
                     "kernel_groups": groups, "kernel_groups_step_ms": timed_step_ms,
                     "kernel_groups_source": "measured in this run: HIP events around every C-ABI call of one more step (yolov3_amd._lib.CallTimer), after the timed steps; that step runs on ONE "
                                             "stream (the timed steps run the filter gradients on a second one: their families would overlap), so kernel_groups_step_ms is the one-stream step"},
    }
    del model, opt, ema, crit
    torch.cuda.empty_cache()
    return rec


def train_main(args, rank, local_rank, world, dev, parallel, yo, print_record):
    rec = run_train(args, rank, world, dev, parallel, yo, args.batch, args.steps, args.warmup)
    if rank == 0:
        print_record(rec)
    parallel.finalize()


def clocks_under_load(fn, seconds=2.5, per_call=1):
    """Shader clock and socket power while `fn` (one forward) runs back to back, sampled with rocm-smi from a thread.  Outside the timed
    region.  MI355X is power-capped under MFMA load: the sustained clock -- not the 2.4 GHz the 2.5 PFLOP/s peak is quoted at -- is what a
    kernel's MFMA rate can be compared with (profiles/r02_clocks.md).  `per_call` = pipeline steps one call of `fn` runs (the sustained-throughput leg calls
    it with the two-stream schedule: calls, steps and seconds are returned too).  Returns None when rocm-smi is not usable (the sustained leg still gets its counts)."""
    import re
    import subprocess
    import threading

    stop, rows = threading.Event(), []

    def sampler():
        while not stop.is_set():
            try:
                txt = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            except Exception:  # noqa: BLE001
                return
            c = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", txt)
            w = re.search(r"Power \(W\): ([\d.]+)", txt)
            if c:
                rows.append((int(c.group(1)), float(w.group(1)) if w else None))

    th = threading.Thread(target=sampler, daemon=True)
    for _ in range(20 // per_call if per_call < 20 else 1):
        fn()
    torch.cuda.synchronize()
    th.start()
    calls = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(max(1, 10 // per_call)):
            fn()
            calls += 1
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    stop.set()
    th.join(timeout=10)
    rows = [r for r in rows if r[0] > 500]   # a sample that caught the queue empty reads the idle clock
    res = {"calls": calls, "steps": calls * per_call, "seconds": round(elapsed, 3)}
    if len(rows) < 3:
        return res if per_call > 1 else None
    clk = sorted(r[0] for r in rows)
    pw = sorted(r[1] for r in rows if r[1])
    res.update({"sclk_mhz_median": clk[len(clk) // 2], "sclk_mhz_min": clk[0], "sclk_mhz_max": clk[-1], "socket_power_w_median": pw[len(pw) // 2] if pw else None,
                "samples": len(rows), "source": "rocm-smi --showclocks --showpower sampled while the load runs back to back (after the timed region)"})
    return res


def calibrate_detect_head(model, x, conf_thres=0.001, row_frac=0.20, strong_frac=0.03, cands_per_image=12000.0):
    """Give the random-weight bench model a Detect head whose OWN output is the NMS load of SURVEY 8(d) -- about a fifth of the rows above conf 0.001 (~5 k of 25 200
    per image), a few percent above 0.25, ~12 k (row, class) candidates per image under multi_label -- so that the timed step is ONE pipeline: forward -> decode ->
    NMS of what the forward produced.  (A random-weight head has objectness ~0.003 everywhere: 0 candidates.)  Only the last 1x1 convs are touched: the objectness
    filters get one gain and one bias shift (two quantiles of the objectness logits over the bench batch are put on logit(0.001) and logit(0.25)), the class filters
    one bias shift found by bisection on the candidate count.  Setup code outside every timed region; plain torch on the head's raw outputs."""
    import math

    det = model.model[-1]
    no = det.no
    with torch.no_grad():
        _, raws = model(x)
        lo = torch.cat([r[..., 4].float().flatten() for r in raws])
        k_hi, k_lo = max(1, int(lo.numel() * strong_frac)), max(2, int(lo.numel() * row_frac))
        top = torch.topk(lo, k_lo).values
        q_strong, q_row = float(top[k_hi - 1]), float(top[-1])
        t_row, t_strong = math.log(conf_thres / (1 - conf_thres)), math.log(0.25 / 0.75)
        gain = (t_strong - t_row) / max(q_strong - q_row, 1e-6)
        shift = t_row - gain * q_row
        obj = [torch.sigmoid(r[..., 4].float() * gain + shift) for r in raws]
        cls = [r[..., 5:].float() for r in raws]
        bs = raws[0].shape[0]

        def cands(dc):
            n = 0
            for o, c in zip(obj, cls):
                n += int(((o[..., None] * torch.sigmoid(c + dc) > conf_thres) & (o[..., None] > conf_thres)).sum())
            return n / bs

        a, b = -20.0, 20.0
        for _ in range(24):
            mid = 0.5 * (a + b)
            if cands(mid) < cands_per_image:
                a = mid
            else:
                b = mid
        dc = 0.5 * (a + b)
        for conv in det.m:   # channel a * no + 4 = objectness of anchor a, a * no + 5 .. = its classes (reference models/yolo.py:96-98)
            w, bia = conv.weight.data, conv.bias.data
            for an in range(det.na):
                w[an * no + 4] *= gain
                bia[an * no + 4] = bia[an * no + 4].float() * gain + shift
                bia[an * no + 5:(an + 1) * no] += dc
            conv.weight.add_(0)   # (bumps the version counter: the engine re-packs its filter banks)
            conv.bias.add_(0)
    return {"objectness_gain": round(gain, 4), "objectness_shift": round(shift, 4), "class_shift": round(dc, 4),
            "targets": {"rows_above_conf_frac": row_frac, "rows_above_0.25_frac": strong_frac, "candidates_per_image": cands_per_image}}


def train_call_families(by_fn):
    """C-ABI calls of one training step (yolov3_amd._lib.CallTimer) grouped into the families of tools/kgroups.py"""
    import re

    fams = [("conv forward + data gradient (implicit GEMM)", r"conv2d_fwd|conv2d_dgrad|stem_conv_fwd|stem_pair|bneck_pair"), ("wgrad (filter gradients)", r"conv2d_wgrad"),
            ("layer 0 backward (BatchNorm backward + filter gradient in one pass)", r"stem_bn_bwd_wgrad"), ("bn / activation passes", r"y3_bn_"), ("filter packing", r"pack_filter"),
            ("loss", r"y3_loss_(fwd|bwd|level)"), ("optimizer (fused SGD / clip / EMA / loss scale)", r"sgd_step|loss_scale_update"),
            ("pool / upsample / layout / decode", r"maxpool|spp|upsample|nchw|nhwc|decode|detect_raw|copy_slice")]
    out = {}
    for fn, (ms, calls) in by_fn.items():
        fam = next((name for name, pat in fams if re.search(pat, fn)), "other library calls")
        g = out.setdefault(fam, {"ms_per_step": 0.0, "calls_per_step": 0})
        g["ms_per_step"] += ms
        g["calls_per_step"] += calls
    for g in out.values():
        g["ms_per_step"] = round(g["ms_per_step"], 3)
    return dict(sorted(out.items(), key=lambda kv: -kv[1]["ms_per_step"]))


def self_launch_needed(gpus: int, env) -> bool:
    """--gpus N > 1 and no torch.distributed.run environment around this process: bench.py has to start the ranks itself"""
    return gpus > 1 and "WORLD_SIZE" not in env and "RANK" not in env


def self_launch_command(gpus: int, argv, port: int):
    """the command the driver would have typed: one rank per GPU of ONE node, rendezvous on 127.0.0.1 (the container hostname may not resolve)"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
            str(Path(__file__).resolve()), *argv]


def self_launch(gpus: int, argv) -> int:
    import socket
    import subprocess

    with socket.socket() as s:   # a free rendezvous port (two bench.py launchers on one node must not meet)
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # RCCL / device-tensor sharing between the ranks needs dmabuf IPC on this host driver
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(self_launch_command(gpus, argv, port), env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--imgsz", type=int, default=640)
    ap.add_argument("--model", default="yolov3")
    ap.add_argument("--nc", type=int, default=80, help="classes (365 = the Objects365 head of BASELINE configs[4])")
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-layers", action="store_true", help="print the per-launch table (rank 0)")
    ap.add_argument("--no-overlap", action="store_true", help="sequential forward -> NMS per step (no second stream)")
    ap.add_argument("--mode", default="infer", choices=["infer", "train"], help="train = BASELINE configs[2] per-GPU shape (data-parallel train step)")
    ap.add_argument("--no-clocks", action="store_true", help="skip the rocm-smi clock / power sampling leg (after the timed region)")
    ap.add_argument("--no-train", action="store_true", help="infer mode: skip the appended train-step leg (BASELINE metric part ii, batch 64, after the timed inference region)")
    ap.add_argument("--train-batch", type=int, default=64)
    ap.add_argument("--train-steps", type=int, default=6)
    ap.add_argument("--train-timeout", type=float, default=240.0, help="N > 1: seconds the appended train leg may take before the inference line is printed without it")
    ap.add_argument("--dist-backend", default=None, choices=["nccl", "gloo"],
                    help="process-group backend (default nccl = RCCL; gloo: plumbing smoke with several ranks on fewer GPUs, tools/gpu_dist_smoke.sh)")
    args = ap.parse_args()

    # `python bench.py --gpus N` on its own (no torch.distributed.run around it): become the launcher -- one rank per GPU on this node, rendezvous on
    # 127.0.0.1 (the reference's train.py:672-683 is launched the same way).  The ranks' stdout is ours: rank 0's single JSON line passes through.
    if self_launch_needed(args.gpus, os.environ):
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))

    # stdout carries exactly ONE line, the JSON record: libraries that chat on fd 1 (RCCL prints a version banner when a communicator is
    # created) go to stderr for the rest of the process, the record is written to the saved descriptor
    sys.stdout.flush()
    _json_fd = os.dup(1)
    os.dup2(2, 1)

    def print_record(obj):
        os.write(_json_fd, (json.dumps(obj) + "\n").encode())

    from yolov3_amd import parallel

    rank, local_rank, world = parallel.init(args.dist_backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    dev = parallel.local_device(local_rank)
    torch.cuda.set_device(dev)

    from oracle import yolo_oracle as yo  # only for the seeded synthetic inputs + cpu_baseline leg
    from yolov3_amd import DetectionModel, non_max_suppression

    if args.mode == "train":
        return train_main(args, rank, local_rank, world, dev, parallel, yo, print_record)

    dtype = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    bs, hw = args.batch, args.imgsz
    torch.manual_seed(0)
    model = DetectionModel(f"{args.model}.yaml", nc=args.nc)
    for m in model.modules():  # well-conditioned BN statistics so activations stay O(1) through 75 layers
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.1)
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    model = model.to(dev).to(dtype).eval()
    x = torch.rand(bs, 3, hw, hw, generator=torch.Generator().manual_seed(rank)).to(dev).to(dtype)
    calib = calibrate_detect_head(model, x)   # the head's own output becomes the NMS load of SURVEY 8(d); before every timed region
    n_rows = sum(3 * (hw // s) ** 2 for s in (8, 16, 32)) if args.model != "yolov3-tiny" else sum(3 * (hw // s) ** 2 for s in (16, 32))
    pred_synth = yo.synth_predictions(bs=bs, n_rows=n_rows, nc=args.nc, seed=2 + rank, img=hw).to(dev).to(dtype)
    nms_kw = dict(conf_thres=0.001, iou_thres=0.6, multi_label=True, max_det=300)

    def step(synthetic=False):
        pred, _ = model(x)
        return pred, non_max_suppression(pred_synth if synthetic else pred, **nms_kw)

    # Throughput form of the same steps (default): the NMS of batch i runs on a second HIP stream while the forward of batch i+1
    # occupies the main stream (NMS is a chain of small latency-bound launches ending in the one device->host copy of the counts;
    # the forward is what fills the CUs).  NMS(i) waits for forward(i)'s completion event, every batch gets its forward, decode
    # and NMS inside the timed region, and the region ends with both streams drained.  --no-overlap times the sequential loop.
    side = torch.cuda.Stream(device=dev)

    def run_steps(n, synthetic=False):
        pred = dets = None
        prev = None
        for _ in range(n):
            pred, _ = model(x)                      # forward(i) enqueued on the main stream; `pred` is a fresh tensor per forward
            ev = torch.cuda.Event()
            ev.record()
            if prev is not None:                    # NMS(i-1), of forward(i-1)'s predictions, next to forward(i)
                with torch.cuda.stream(side):
                    side.wait_event(prev[1])
                    prev[0].record_stream(side)
                    dets = non_max_suppression(pred_synth if synthetic else prev[0], **nms_kw)
            prev = (pred, ev)
        if prev is not None:
            with torch.cuda.stream(side):
                side.wait_event(prev[1])
                prev[0].record_stream(side)
                dets = non_max_suppression(pred_synth if synthetic else prev[0], **nms_kw)
        torch.cuda.current_stream().wait_stream(side)
        return pred, dets

    barrier = parallel.barrier

    if args.no_overlap:
        for _ in range(args.warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            pred, dets = step()
    else:
        run_steps(args.warmup)
        barrier()
        t0 = time.perf_counter()
        pred, dets = run_steps(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    dt = parallel.max_over_ranks(dt, dev)
    group = parallel.describe(dev)   # (collective: every rank calls it) who took part, as the communicator sees it

    if rank == 0:
        # ---- leg split + per-kernel roofline (outside the timed region) ----
        def timed(fn, n=5):
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / n

        from yolov3_amd import ops as y3ops

        # the same steps strictly sequentially (forward -> NMS on one stream), outside the timed region: `value` depends on the two-stream
        # schedule, this is the figure without it (same process, same box)
        seq = None
        if not args.no_overlap:
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            ts = time.perf_counter()
            for _ in range(max(5, args.steps // 2)):
                step()
            torch.cuda.synchronize()
            seq = round(bs * max(5, args.steps // 2) / (time.perf_counter() - ts), 2)
        t_fwd = timed(lambda: model(x))
        t_nms = timed(lambda: non_max_suppression(pred_synth, **nms_kw))
        cand_synth = y3ops.nms_raw.last_candidates / bs
        t_nms_own = timed(lambda: non_max_suppression(pred, **nms_kw))
        cand_own = y3ops.nms_raw.last_candidates / bs
        rows_own = float((pred[..., 4] > nms_kw["conf_thres"]).sum()) / bs
        kept_own = sum(int(d.shape[0]) for d in dets) / bs
        # the secondary figure: the same schedule with the NMS leg on the seeded synthetic prediction tensor (what rounds 1-4 reported as `value`)
        synth_rate = None
        if not args.no_overlap:
            run_steps(2, synthetic=True)
            torch.cuda.synchronize()
            ts = time.perf_counter()
            run_steps(args.steps, synthetic=True)
            torch.cuda.synchronize()
            synth_rate = round(bs * args.steps / (time.perf_counter() - ts), 2)
        plan = next(iter(model._plans.values()))
        forms = {}
        groups = per_kernel_times(plan, forms=forms)
        if args.profile_layers:
            model(x, profile=True)
        dom = max((k for k in groups if groups[k][0] > 0), key=lambda k: groups[k][2])
        fl, by, sec, nl = groups[dom]
        total_conv_flops = sum(g[0] for g in groups.values()) / 5
        total_kernel_s = sum(g[2] for g in groups.values()) / 5
        pmc, pmc_file = {}, None
        for cand in ("r06_pmc_summary.json", "r05_pmc_summary.json", "r04_pmc_summary.json", "r03_pmc_summary.json", "r02_pmc_summary.json", "r01_pmc_summary.json"):   # HBM traffic per launch comes from the committed rocprofv3 --pmc passes (bench.py cannot run the profiler)
            try:
                pmc = json.load(open(ROOT / "profiles" / cand)).get(dom.rsplit("/", 1)[0], {})  # PMC averages are per kernel symbol
            except OSError:
                continue
            if pmc:
                pmc_file = cand
                break
        # the PMC average is over EVERY launch of the kernel symbol (e.g. v6 serves 3x3 stride-2 and 1x1 layers): the algorithmic bytes
        # it is compared with are summed over the same launch set
        sym = dom.rsplit("/", 1)[0]
        sym_groups = [g for k, g in groups.items() if k.rsplit("/", 1)[0] == sym]
        sym_bytes, sym_launches = sum(g[1] for g in sym_groups), sum(g[3] for g in sym_groups)
        # 3x3 launches with Cin >= 128 are MFMA-bound, the 1x1 / small-channel launches of the same template HBM-bound
        # (SURVEY Appendix B): groups are split by filter size so that the dominant group has ONE bounding roofline
        bound = "mfma" if fl / max(by, 1.0) > 312.0 else "hbm"
        roofline = {
            "kernel": dom,
            "bound": bound,
            "achieved": round(fl / sec / 1e12, 2) if bound == "mfma" else round(by / sec / 1e9, 1),
            "peak": MFMA_PEAK_TFLOPS if bound == "mfma" else HBM_PEAK_GBS,
            "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
            "frac": round(fl / sec / 1e12 / MFMA_PEAK_TFLOPS, 4) if bound == "mfma" else round(by / sec / 1e9 / HBM_PEAK_GBS, 4),
            "traffic": round(pmc["hbm_bytes_per_launch"]) if "hbm_bytes_per_launch" in pmc else None,
            "traffic_source": f"profiles/{pmc_file} (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, avg per launch of this kernel symbol)" if pmc else None,
            "traffic_launch_set": f"all {sym_launches // 5} launches per forward of kernel source {sym} (every instantiation)",
            "algorithmic_bytes_per_launch_same_set": round(sym_bytes / max(sym_launches, 1)),
            "traffic_over_algorithmic": round(pmc["hbm_bytes_per_launch"] / (sym_bytes / max(sym_launches, 1)), 3) if "hbm_bytes_per_launch" in pmc else None,
            "mfma_busy_frac_pmc": round(pmc["mfma_busy_frac_of_simd_cycles"], 4) if "mfma_busy_frac_of_simd_cycles" in pmc else None,
            "forms": {k: {"ms": round(g[2] / 5 * 1e3, 3), "launches": g[3] // 5, "tflops": round(g[0] / g[2] / 1e12, 1), "frac": round(g[0] / g[2] / 1e12 / MFMA_PEAK_TFLOPS, 4)}
                      for k, g in sorted(forms.items(), key=lambda kv: -kv[1][2])},   # the launch geometries of the dominant kernel source, apart
            "algorithmic_bytes_per_launch": round(by / nl),
            "launches_per_forward": nl // 5,
            "avg_launch_us": round(sec / nl * 1e6, 2),
            "algorithmic_gflop_per_launch": round(fl / nl / 1e9, 3),
            "hbm_frac_of_same_kernel": round(by / sec / 1e9 / HBM_PEAK_GBS, 4),
            "whole_forward": {
                "conv_tflops": round(total_conv_flops / total_kernel_s / 1e12, 2),
                "gflop_per_image": round(total_conv_flops / bs / 1e9, 2),
                "kernel_ms": round(total_kernel_s * 1e3, 3),
                "by_kernel": {k: {"ms": round(g[2] / 5 * 1e3, 3), "launches": g[3] // 5, "tflops": round(g[0] / g[2] / 1e12, 1), "gbs": round(g[1] / g[2] / 1e9, 1)}
                              for k, g in sorted(groups.items(), key=lambda kv: -kv[1][2])},
            },
        }
        # context for `frac`: what the vendor GEMM (torch.matmul -> hipBLASLt) reaches on the plain GEMM the dominant layer group is equivalent
        # to -- same FLOPs, no halo gather, no bias / SiLU / residual epilogue -- measured here, after the timed region, rank 0 at N = 1 only
        if world == 1 and bound == "mfma" and dom.endswith("/3x3"):
            try:
                # every distinct GEMM shape of the dominant group, with how many of its launches have that shape: the launch-weighted vendor rate is what the group's
                # average rate stands beside
                shapes = {}
                for l in plan.launches:
                    if l.flops and not getattr(l, "kernel", ""):
                        var = plan.conv_variant(l)
                        if f"conv_igemm_{KERNEL_FAMILY.get(var, var)}/3x3" == dom:
                            d_, yt_ = l.keep[0], l.keep[2]
                            key = (yt_.n * yt_.h * yt_.w, d_.cout, 9 * d_.cin)
                            shapes[key] = shapes.get(key, 0) + 1
                per_shape, t_all, f_all = [], 0.0, 0.0
                for (Mg, Ng, Kg), cnt in sorted(shapes.items(), key=lambda kv: -kv[1])[:4]:
                    ga = torch.randn(Mg, Kg, device=dev, dtype=dtype)
                    gb = torch.randn(Ng, Kg, device=dev, dtype=dtype)
                    for _ in range(3):
                        torch.matmul(ga, gb.t())
                    t_g = timed(lambda: torch.matmul(ga, gb.t()), n=10)
                    fl_g = 2.0 * Mg * Ng * Kg
                    per_shape.append({"M": Mg, "N": Ng, "K": Kg, "launches": cnt, "us": round(t_g * 1e6, 1), "tflops": round(fl_g / t_g / 1e12, 1)})
                    t_all += cnt * t_g
                    f_all += cnt * fl_g
                    del ga, gb
                roofline["vendor_gemm_same_shape"] = {"tflops": round(f_all / t_all / 1e12, 1), "frac": round(f_all / t_all / 1e12 / MFMA_PEAK_TFLOPS, 4), "per_shape": per_shape,
                                                      "note": "torch.matmul(A, B^T) fp16/bf16 random operands: hipBLASLt on the plain GEMMs the launches of the dominant group are equivalent to "
                                                              "(no im2col / halo, no fused epilogue), launch-weighted like `achieved`"}
            except Exception as e:  # noqa: BLE001
                roofline["vendor_gemm_same_shape"] = {"error": f"{type(e).__name__}: {e}"[:200]}
        cpu = None if (args.no_cpu_baseline or world > 1) else cpu_baseline()   # rank 0 at N = 1 only
        value = world * bs * args.steps / dt
        roofline["whole_step_frac"] = round(total_conv_flops / (dt / args.steps) / 1e12 / MFMA_PEAK_TFLOPS, 4)   # conv FLOPs of one step / wall time of one step / 2.5 PF
        clk = None if args.no_clocks else clocks_under_load(lambda: model(x))
        roofline["clocks_under_load"] = clk
        # the timed window is steps x ~6 ms on a chip whose clock follows a power budget: the same two-stream pipeline for >= 2 s, outside `value`
        sustained = None
        if not args.no_clocks and not args.no_overlap:
            sus = clocks_under_load(lambda: run_steps(8), seconds=2.0, per_call=8)
            if sus:
                sustained = {"images_per_sec": round(bs * sus["steps"] / sus["seconds"], 2), "steps": sus["steps"], "seconds": sus["seconds"],
                             "sclk_mhz_median": sus.get("sclk_mhz_median"), "socket_power_w_median": sus.get("socket_power_w_median"),
                             "note": "forward(i + 1) beside NMS(i) of the model's own output, back to back for >= 2 s after the timed region (rank 0)"}
        if clk and bound == "mfma":   # the same MFMA peak at the clock the chip actually sustains under this load (power cap)
            peak_at_clk = MFMA_PEAK_TFLOPS * clk["sclk_mhz_median"] / 2400.0
            roofline["peak_at_sustained_clock"] = round(peak_at_clk, 1)
            roofline["frac_of_peak_at_sustained_clock"] = round(fl / sec / 1e12 / peak_at_clk, 4)
        out = {
            "metric": "images/sec (640x640) inference+NMS",
            "value": round(value, 2),
            "unit": "images/sec",
            "n_gpus": world,
            "process_group": group,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16" if dtype == torch.float16 else "bf16",
            "data": "synthetic (seeded uniform images; random-init weights with conditioned BN stats; Detect head calibrated to the NMS load of SURVEY 8d; NMS runs on the model's own output)",
            "config": {
                "workload": f"{args.model} inference {hw}x{hw} batch={bs}/GPU {args.dtype} nc={args.nc} + NMS(conf 0.001, iou 0.6, multi_label, max_det 300)" + (" [BASELINE configs[1]]" if (args.model, hw, bs, args.dtype, args.nc) == ("yolov3", 640, 32, "fp16", 80) else ""),
                "global_batch": world * bs,
                "parallelism": f"replicas x{world} (no data-path collective)",
                "schedule": "sequential forward -> NMS(pred) per batch" if args.no_overlap else "NMS of batch i's predictions on a second HIP stream beside the forward of batch i+1 (every batch completes inside the timed region)",
            },
            "sustained_images_per_sec": sustained["images_per_sec"] if sustained else None,
            "sustained": sustained,
            "sequential_images_per_sec_per_gpu": seq,
            "legs_ms": {"forward+decode": round(t_fwd * 1e3, 3), "nms_on_model_output": round(t_nms_own * 1e3, 3), "nms_synthetic_pred": round(t_nms * 1e3, 3)},
            "nms_candidates_per_image": {"model_output": round(cand_own, 1), "synthetic_pred": round(cand_synth, 1)},
            "nms_load_model_output": {"rows_above_conf_per_image": round(rows_own, 1), "candidates_per_image": round(cand_own, 1), "detections_kept_per_image": round(kept_own, 1),
                                      "head_calibration": calib},
            "synthetic_nms_tensor": {"images_per_sec": synth_rate, "note": "same two-stream schedule, NMS leg on the seeded synthetic (bs, 25200, 85) tensor instead of the forward's output (rounds 1-4's headline form)"},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        try:   # the UNMODIFIED reference timed where it can run (the build container; /root/reference does not exist on the GPU box): committed record
            ref = json.load(open(ROOT / "profiles" / "r03_cpu_reference.json"))
            out["cpu_reference_recorded"] = {"kind": ref["kind"], "host": ref["host"], "inference": ref["inference"], "train_step": ref["train_step"],
                                             "source": "profiles/r03_cpu_reference.json (tools/cpu_reference.py, not measured in this run)"}
        except (OSError, ValueError, KeyError):
            pass
    # BASELINE metric part (ii): the train step, measured in the same run AFTER the timed inference region.  At N > 1 it is the
    # data-parallel step of BASELINE configs[2] (batch 64 per GPU, RCCL gradient all-reduce overlapped with the backward; `value` = images/s
    # of the whole job), so the driver's N = 1, 2, 4, 8 runs carry the training scaling curve next to the inference replicas.  The
    # inference line must survive whatever happens in that leg: an exception is recorded in `train`, and a watchdog prints the line
    # and ends the process if a collective hangs.
    import threading

    state = {"printed": False}

    def emit(train):
        if rank == 0 and not state["printed"]:
            state["printed"] = True
            out["train"] = train
            print_record(out)

    def on_timeout():
        emit({"error": f"train leg did not finish within {args.train_timeout:.0f} s (world {world})"})
        os._exit(0)

    train, wd = None, None
    if not args.no_train:
        if world > 1:
            wd = threading.Timer(args.train_timeout, on_timeout)
            wd.daemon = True
            wd.start()
        try:
            del model
            torch.cuda.empty_cache()
            train = run_train(args, rank, world, dev, parallel, yo, args.train_batch, args.train_steps, 2)
        except Exception as e:  # noqa: BLE001
            if world == 1:
                raise
            train = {"error": f"{type(e).__name__}: {e}"[:400]}
    emit(train)
    parallel.finalize()
    if wd is not None:
        wd.cancel()


if __name__ == "__main__":
    main()
