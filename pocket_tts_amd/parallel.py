"""Multi-GPU plumbing: utterances are independent, so the node runs one replica per GPU (one process per
GPU under `torch.distributed.run`) and shards the utterance list; there is NO collective on the data
path.  torch.distributed (RCCL on GPUs, gloo in the CPU tests) is only used to
line up the start/end of a measurement and to reduce scalars."""

from __future__ import annotations

import os


def env_ranks():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_distributed(backend: str, device=None):
    """-> torch.distributed module (initialised) or None when WORLD_SIZE == 1."""
    rank, local, world = env_ranks()
    if world == 1:
        return None
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def shard(n_items: int, rank: int, world: int) -> list:
    """utterance i -> rank i mod world"""
    return list(range(rank, n_items, world))


def reduce_scalar(value: float, dist, op: str = "max", device="cpu") -> float:
    if dist is None:
        return float(value)
    import torch

    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
    return float(t.item())


def job_throughput(units_per_rank: float, wall_s: float, dist, device="cpu"):
    """whole-job value = units processed by ALL ranks / max-over-ranks wall time"""
    total = reduce_scalar(units_per_rank, dist, "sum", device)
    wall = reduce_scalar(wall_s, dist, "max", device)
    return total / wall, wall
