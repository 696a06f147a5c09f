"""Data-parallel plumbing for the RBVAE trainer: one process per GPU, items sharded
along the batch axis, ONE all-reduce of the flat gradient buffer per step.

The reference has no distributed path; every loss term is a mean over
items, so with equal per-rank batches  global_grad = mean_r(local_grad_r)  exactly
(up to summation order).  backend "nccl" is RCCL on ROCm; tests use gloo on the CPU."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun). Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_items(n_items: int, rank: int, world: int):
    """Indices of the global batch that rank `rank` trains on: items rank, rank+world, ... (equal counts)."""
    if n_items % world:
        raise ValueError(f"global batch {n_items} is not divisible by world size {world}")
    return list(range(rank, n_items, world))


def allreduce_mean_(flat: torch.Tensor, group=None):
    """In-place mean of a flat gradient buffer over the ranks of `group`."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(world)
    return flat


def broadcast_(flat: torch.Tensor, src=0, group=None):
    """Make every rank start from rank `src`'s parameters."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat, src=src, group=group)
    return flat
