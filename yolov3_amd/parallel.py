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
