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


def init(backend: str | None = None, force: bool = False):
    """Initialise the default process group from the torch.distributed.run environment (no-op for world 1 unless `force`:
    a one-rank RCCL communicator exercises the same device collectives, streams and event ordering as N ranks)."""
    rank, local_rank, world = env_rank()
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = backend or os.environ.get("Y3_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, **kw)
    return rank, local_rank, world


def local_device(local_rank: int) -> torch.device:
    """cuda:<local_rank>; on a box with fewer GPUs than local ranks (the 2-rank gloo smoke of the multi-process plumbing on a 1-GPU box) the
    ranks share the devices round-robin -- RCCL itself refuses two ranks on one device"""
    n = torch.cuda.device_count()
    return torch.device("cuda", local_rank % n if n else 0)


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


def describe(device=None) -> dict:
    """Self-evidence of the process group for a benchmark line: backend as torch reports it ("nccl" is RCCL on ROCm), the world size the
    COMMUNICATOR sees (not the environment variable), and every rank's device (index, name, PCI bus id) gathered over the group -- an N-GPU
    run proves that N ranks on N distinct devices took part."""
    dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu"))
    me = {"rank": env_rank()[0], "pid": os.getpid(), "device": str(dev)}
    if dev.type == "cuda":
        pr = torch.cuda.get_device_properties(dev)
        me["name"] = pr.name
        me["pci_bus_id"] = getattr(pr, "pci_bus_id", None)
        me["uuid"] = str(getattr(pr, "uuid", ""))
    if not (dist.is_available() and dist.is_initialized()):
        return {"backend": None, "world_size": 1, "ranks": [me]}
    ranks = [None] * dist.get_world_size()
    dist.all_gather_object(ranks, me)
    backend = dist.get_backend()
    return {"backend": backend + (" (RCCL)" if backend == "nccl" else ""), "world_size": dist.get_world_size(), "rccl_ranks": dist.get_world_size() if backend == "nccl" else 0,
            "distinct_devices": len({(r.get("pci_bus_id"), r.get("uuid"), r["device"]) for r in ranks}), "ranks": ranks}


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

    No extra passes over the gradients: the training plan writes every gradient into one flat fp32 arena in the order the
    backward produces them (train_engine.TrainPlan.grad_alloc), so a bucket is a contiguous range of that arena and is
    all-reduced IN PLACE with ReduceOp.AVG (RCCL divides on the wire); the tensors autograd receives are views of the
    arena.  Gradients that are not arena views (or a bf16 wire) take the flatten / copy-back path.

    `exchange` (or the environment variable Y3_GRAD_EXCHANGE) picks the collective of a bucket:
      "all_reduce"  one RCCL all-reduce(AVG) per bucket (the default: RCCL chooses its rings / channels over the xGMI mesh);
      "direct"      the two-phase exchange SURVEY 8(e) prices for the fully connected mesh: every rank owns 1/P of the bucket,
                    phase 1 all-to-all sends each peer its shard (P-1 point-to-point transfers per rank, one per link, at once), the owner sums the
                    P contributions in rank order and scales by 1/P, phase 2 all-gathers the averaged shards back in place.  Wire bytes per link and
                    phase: bucket / P (31 MB for yolov3's 247.8 MB at P = 8) instead of a ring's 2 (P-1)/P x bucket over one link.  Every rank receives
                    the owner's sum, so the replicas stay bit-identical whatever P is.  Needs one receive buffer of the bucket's size.
                    Buckets whose length P does not divide take the all-reduce.  NOT measured on an 8-GPU node (the build box has one GPU): opt-in.
    """

    EXCHANGES = ("all_reduce", "direct")

    def __init__(self, bucket_bytes: int = 64 << 20, wire_dtype: torch.dtype | None = None, group=None, force: bool = False, exchange: str | None = None):
        self.bucket_bytes, self.wire_dtype, self.group = bucket_bytes, wire_dtype, group
        self.exchange = exchange or os.environ.get("Y3_GRAD_EXCHANGE", "all_reduce")
        if self.exchange not in self.EXCHANGES:
            raise ValueError(f"GradBuckets: exchange {self.exchange!r} is not one of {self.EXCHANGES}")
        self.collectives = {"all_reduce": 0, "direct": 0}   # buckets sent by each form since construction (tests, bench line)
        self._warned_fallback = False
        self._recv = None          # the direct form's receive buffer (grown to the largest bucket, reused: collectives of one side stream run in order)
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.force = force   # run the bucket / side-stream / collective machinery even at world size 1 (single-GPU tests of the exchange step)
        self._pending: list = []   # (keys, tensors) of the bucket being filled
        self._bytes = 0
        self._inflight: list = []  # (work, flat, keys, shapes, dtypes, event)
        self._out: dict = {}
        self._side = None
        self._ranges: list = []    # (arena, lo, hi) of the in-place buckets in flight

    def _stream(self, device):
        if device.type != "cuda":
            return None
        if self._side is None:
            self._side = torch.cuda.Stream(device=device)
        return self._side

    def add(self, key, grad: torch.Tensor):
        """Hand over one finished gradient (called in reverse layer order by the backward plan)."""
        if self.world == 1 and not self.force:
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

    def flush(self):
        """launch the bucket that is being filled now (the training plan calls this when only a few MB of gradients are still to come, so
        that the collective left for finish() is small)"""
        self._launch()

    @staticmethod
    def _arena_range(tensors):
        """(base, lo, hi) when every tensor is a contiguous fp32 view of ONE flat base tensor (the training plan's gradient arena):
        the bucket is then the element range [lo, hi) of the base and is reduced in place.  None otherwise."""
        base = tensors[0]._base
        if base is None or base.dim() != 1 or base.dtype != torch.float32:
            return None
        lo, hi, covered = None, None, 0
        for t in tensors:
            if t._base is not base or not t.is_contiguous() or t.dtype != torch.float32:
                return None
            o = t.storage_offset() - base.storage_offset()
            lo = o if lo is None else min(lo, o)
            hi = o + t.numel() if hi is None else max(hi, o + t.numel())
            covered += (t.numel() + 63) // 64 * 64   # the arena hands out 64-element-aligned slices (TrainPlan.grad_alloc)
        # the members must TILE [lo, hi): a range that merely spans them could swallow an arena slice that has not been handed over yet (a kernel
        # may still be writing it, and its own bucket would average it a second time) -- only the last member's pad may be missing
        if not (hi - lo <= covered < hi - lo + 64):
            return None
        return base, lo, hi

    def _reduce(self, flat):
        """average `flat` over the ranks, in place; returns the async work handle"""
        if not (dist.is_available() and dist.is_initialized()):
            return None
        if self.exchange == "direct" and self.world > 1 and flat.numel() > 0:
            if flat.numel() % self.world == 0:
                return self._reduce_direct(flat)
            if not self._warned_fallback:
                self._warned_fallback = True
                import warnings

                warnings.warn(f"GradBuckets(exchange='direct'): a bucket of {flat.numel()} elements is not divisible by the world size {self.world}; it takes the all-reduce")
        self.collectives["all_reduce"] += 1
        if dist.get_backend(self.group) == "nccl":
            return dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group, async_op=True)   # RCCL averages on the wire: no divide pass
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return (work, flat)   # gloo has no AVG: divide after the wait

    def _reduce_direct(self, flat):
        """all-to-all of the shards, the owner's rank-ordered sum x 1/P, all-gather in place (class docstring); returns the all-gather's handle.
        Called on the side stream: wait() of the first phase orders that stream (not the host, with RCCL) behind the transfer."""
        P, n = self.world, flat.numel() // self.world
        rank = dist.get_rank(self.group)
        if self._recv is None or self._recv.numel() < flat.numel() or self._recv.device != flat.device or self._recv.dtype != flat.dtype:
            self._recv = torch.empty(flat.numel(), dtype=flat.dtype, device=flat.device)
        recv = self._recv[: flat.numel()]
        dist.all_to_all_single(recv, flat, group=self.group, async_op=True).wait()   # recv.view(P, n)[r] = rank r's copy of MY shard
        mine = flat[rank * n : (rank + 1) * n]
        if flat.is_cuda and flat.dtype == torch.float32:
            # the owner's sum: rank order 0 .. P-1, times 1 / P, one launch of the library (csrc/optim.hip::shard_mean_kernel) on the current (side) stream
            from . import _lib, ops

            _lib.check(_lib.lib().y3_shard_mean(recv.data_ptr(), P, n, 1.0 / P, mine.data_ptr(), ops.stream_ptr()), "y3_shard_mean")
        else:                        # host tensors (the gloo tests) and reduced wire dtypes: the same sum in the same order with torch ops
            parts = recv.view(P, n)
            acc = parts[0].clone() if P > 1 else parts[0]
            for r in range(1, P):
                acc.add_(parts[r])
            torch.mul(acc, 1.0 / P, out=mine)
        self.collectives["direct"] += 1
        nccl = dist.get_backend(self.group) == "nccl"
        return dist.all_gather_into_tensor(flat, mine if nccl else mine.clone(), group=self.group, async_op=True)   # RCCL gathers in place (send == recv + rank x n)

    def _launch(self):
        if not self._pending:
            return
        keys = [k for k, _, _ in self._pending]
        tensors = [g for _, g, _ in self._pending]
        events = [e for _, _, e in self._pending if e is not None]
        self._pending, self._bytes = [], 0
        dev = tensors[0].device
        side = self._stream(dev)
        rng = self._arena_range(tensors) if self.wire_dtype in (None, torch.float32) else None
        if rng is not None:
            # in-place ranges of one arena must be disjoint: a range that reaches into one that is already being reduced would be averaged twice
            # (harmless with AVG over identical values, a race on the SUM + divide path) -- such a bucket takes the flatten / copy-back path
            base, lo, hi = rng
            if any(b is base and lo < h and l < hi for b, l, h in self._ranges):
                rng = None
            else:
                self._ranges.append((base, lo, hi))

        def issue():
            direct = self.exchange == "direct" and self.world > 1
            if rng is not None:
                base, lo, hi = rng
                if direct:                  # the last member's own pad (the arena hands out 64-element slices) makes the length divisible by P = 2 .. 64; another P
                    hi = min(base.numel(), (hi + 63) // 64 * 64)   # (3, 6, ...) takes the all-reduce for this bucket (one warning): the elements behind `hi` are other
                                                                   # tensors' slices, which the backward may be writing while this range is on the wire
                flat = base[lo:hi]          # padding between slices rides along (<= 252 B per tensor); the slices ARE the results
                return flat, self._reduce(flat), None
            wire = self.wire_dtype or torch.float32
            parts = [t.reshape(-1).to(wire) for t in tensors]
            n = sum(p.numel() for p in parts)
            if direct and n % self.world:
                parts.append(torch.zeros(self.world - n % self.world, dtype=wire, device=dev))
            flat = torch.cat(parts)
            return flat, self._reduce(flat), [(t.shape, t.dtype) for t in tensors]

        if side is not None:
            for e in events:   # every gradient of the bucket has been produced (whatever stream issued it)
                side.wait_event(e)
            for t in tensors:
                t.record_stream(side)
            with torch.cuda.stream(side):
                flat, work, meta = issue()
        else:
            flat, work, meta = issue()
        self._inflight.append((work, flat, keys, tensors, meta, side))

    def finish(self) -> dict:
        """Flush the last bucket, wait for every collective, return {key: averaged gradient}."""
        self._launch()
        for work, flat, keys, tensors, meta, side in self._inflight:
            if isinstance(work, tuple):      # gloo all-reduce: SUM + divide
                work[0].wait()
                flat.div_(self.world)
            elif work is not None:
                work.wait()                  # nccl: makes the current stream wait for the collective's stream
            if side is not None:
                torch.cuda.current_stream(flat.device).wait_stream(side)
            if meta is None:                 # reduced in place inside the arena: the views are the results
                for k, t in zip(keys, tensors):
                    self._out[k] = t
            else:
                off = 0
                for k, (shp, dt) in zip(keys, meta):
                    n = 1
                    for d in shp:
                        n *= d
                    self._out[k] = flat[off : off + n].view(shp).to(dt)
                    off += n
        self._inflight = []
        self._ranges = []
        out, self._out = self._out, {}
        return out


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None):
    """Start every rank from rank `src`'s parameters and buffers (what DDP's constructor does, train.py:323)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)
