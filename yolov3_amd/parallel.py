"""One-process-per-GPU helpers (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU).

Inference/validation does not shard a batch across ranks in the reference (val.py is single-process; train-time
val runs on rank 0 only, train.py:441-459), so multi-GPU inference here is N independent replicas on disjoint image
shards with NO data-path collective; the only collectives are the timing barrier and a max-reduce of the elapsed
time.  Training's gradient all-reduce (reference utils/torch_utils.py:60-72) lands with the backward kernels.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_rank():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init(backend: str | None = None):
    """Initialise the default process group from the torch.distributed.run environment (no-op for world 1)."""
    rank, local_rank, world = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, **kw)
    return rank, local_rank, world


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def max_over_ranks(value: float, device=None) -> float:
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device or ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shard_range(total: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of `total` images for this rank (ragged tail goes to the low ranks), the
    replica analogue of the reference's DistributedSampler split (utils/dataloaders.py:115)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def finalize():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ data-parallel training
class GradBuckets:
    """Gradient averaging across ranks, overlapped with the backward pass.

    Reference behaviour: DistributedDataParallel (utils/torch_utils.py:60-72) all-reduces (average) the 222 fp32
    gradient tensors (247.8 MB for yolov3) in 25 MiB buckets while autograd runs.  The MI355X training engine
    produces gradients layer by layer in REVERSE layer order inside one autograd node, so it feeds them to this
    object as they appear: a bucket is flattened and its all-reduce (RCCL over xGMI, or gloo on CPU) is launched
    asynchronously on a side stream as soon as it is full, while the remaining backward kernels keep the compute
    stream busy.  `finish()` waits for the collectives and hands back the averaged gradients.

    xGMI is point-to-point (7 links x ~153 GB/s): ring collectives are per-link bound, so buckets are fewer and larger
    than DDP's default (64 MiB: 4 collectives for yolov3) to amortise launch/latency; `wire_dtype=torch.bfloat16`
    halves the bytes on the wire (changes rounding: off by default).
    """

    def __init__(self, bucket_bytes: int = 64 << 20, wire_dtype: torch.dtype | None = None, group=None):
        self.bucket_bytes, self.wire_dtype, self.group = bucket_bytes, wire_dtype, group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self._pending: list = []   # (keys, tensors) of the bucket being filled
        self._bytes = 0
        self._inflight: list = []  # (work, flat, keys, shapes, dtypes, event)
        self._out: dict = {}
        self._side = None

    def _stream(self, device):
        if device.type != "cuda":
            return None
        if self._side is None:
            self._side = torch.cuda.Stream(device=device)
        return self._side

    def add(self, key, grad: torch.Tensor):
        """Hand over one finished gradient (called in reverse layer order by the backward plan)."""
        if self.world == 1:
            self._out[key] = grad
            return
        ev = None
        if grad.is_cuda:   # producers may sit on different streams (filter gradients come from the engine's side stream)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(grad.device))
        self._pending.append((key, grad, ev))
        self._bytes += grad.numel() * grad.element_size()
        if self._bytes >= self.bucket_bytes:
            self._launch()

    def _launch(self):
        if not self._pending:
            return
        keys = [k for k, _, _ in self._pending]
        tensors = [g for _, g, _ in self._pending]
        events = [e for _, _, e in self._pending if e is not None]
        self._pending, self._bytes = [], 0
        dev = tensors[0].device
        side = self._stream(dev)
        wire = self.wire_dtype or torch.float32
        if side is not None:
            for e in events:   # every gradient of the bucket has been produced (whatever stream issued it)
                side.wait_event(e)
            for t in tensors:
                t.record_stream(side)
            with torch.cuda.stream(side):
                flat = torch.cat([t.reshape(-1).to(wire) for t in tensors])
                work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            flat = torch.cat([t.reshape(-1).to(wire) for t in tensors])
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._inflight.append((work, flat, keys, [t.shape for t in tensors], [t.dtype for t in tensors], side))

    def finish(self) -> dict:
        """Flush the last bucket, wait for every collective, return {key: averaged gradient}."""
        if self.world > 1:
            self._launch()
            for work, flat, keys, shapes, dtypes, side in self._inflight:
                work.wait()
                if side is not None:
                    torch.cuda.current_stream(flat.device).wait_stream(side)
                flat = flat / self.world
                off = 0
                for k, shp, dt in zip(keys, shapes, dtypes):
                    n = 1
                    for d in shp:
                        n *= d
                    self._out[k] = flat[off : off + n].view(shp).to(dt)
                    off += n
            self._inflight = []
        out, self._out = self._out, {}
        return out


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None):
    """Start every rank from rank `src`'s parameters and buffers (what DDP's constructor does, train.py:323)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)
