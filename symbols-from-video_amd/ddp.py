"""Data-parallel plumbing for the RBVAE trainer: one process per GPU, items sharded
along the batch axis, ONE all-reduce of the flat gradient buffer per step.

The reference has no distributed path; every loss term is a mean over
items, so with equal per-rank batches  global_grad = mean_r(local_grad_r)  exactly
(up to summation order).  backend "nccl" is RCCL on ROCm; tests use gloo on the CPU."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, timeout_s=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun). Returns (rank, world, local_rank).
    timeout_s (default RBVAE_DIST_TIMEOUT or 180 s): rendezvous and collective timeout -- a rank that never arrives or a
    collective that never completes becomes an error on the surviving ranks instead of an indefinite wait (the
    reference has no failure handling at all; SURVEY.md 5 asks the data-parallel layer to time out at least)."""
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
        import datetime
        if timeout_s is None:
            timeout_s = float(os.environ.get("RBVAE_DIST_TIMEOUT", "180"))
        dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=float(timeout_s)))
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


class GradReducer:
    """The data-parallel exchange step of the fused trainer: the flat f32 gradient buffer summed over the ranks (the
    1/world factor is folded into the Adam kernel), as ONE all-reduce, or as two buckets cut where the backward pass
    finishes them:

      tail = flat[cut:]   decoder CNN + both LSTM stacks -- final when the encoder CNN's backward pass starts
      head = flat[:cut]   encoder CNN                     -- final at the end of the pass

    start_tail() launches the tail's all-reduce asynchronously (RCCL runs it on its own stream, beside the HIP graph
    that computes the head's gradients); finish() all-reduces the head and makes the current stream wait for both.
    Bucket order = reverse layer order, as SURVEY.md 8e asks; the cut sits where the side stream's work is done
    when the main chain arrives (engine.backward(cut=...)), so it costs no idle time."""

    def __init__(self, gflat: torch.Tensor, cut: int, group=None, force: bool = False, wire_dtype=None):
        """force: issue the collectives even in a one-rank group (rehearsals of the multi-rank schedule on one GPU).
        wire_dtype: torch.float32 (default) or torch.bfloat16 (RBVAE_DDP_BF16=1): the buckets cross the links as bf16 --
        half the bytes of the exchange (SURVEY.md 8e allows either); every rank rounds its gradient to bf16, the collective
        sums in bf16, the sum is widened back into the f32 buffer (relative error of the summed gradient ~2^-9 per rank
        added: tests/test_ddp_cpu.py bounds it against the f32 exchange)."""
        self.gflat, self.cut, self.group = gflat, int(cut), group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        if force and dist.is_available() and dist.is_initialized():
            self.world = max(self.world, 2)
        if not 0 < self.cut < gflat.numel() or self.cut % 4:
            raise ValueError(f"bucket cut {cut} outside the buffer or not 16-byte aligned")
        if wire_dtype is None:
            wire_dtype = torch.bfloat16 if os.environ.get("RBVAE_DDP_BF16", "0") == "1" else torch.float32
        if wire_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("wire_dtype must be torch.float32 or torch.bfloat16")
        self.wire_dtype = wire_dtype
        # persistent staging buffer of the bf16 exchange (same address every step: captured graphs replay the casts)
        self._wire = torch.empty(gflat.numel(), dtype=torch.bfloat16, device=gflat.device) if wire_dtype == torch.bfloat16 else None

    @property
    def tail(self) -> torch.Tensor:
        return self.gflat[self.cut:]

    @property
    def head(self) -> torch.Tensor:
        return self.gflat[:self.cut]

    class _Pending:
        """an asynchronous bucket exchange: wait() orders the current stream behind it (and widens a bf16 bucket back)"""

        def __init__(self, work, part=None, wire=None):
            self.work, self.part, self.wire = work, part, wire

        def wait(self):
            if self.work is not None:
                self.work.wait()
            if self.wire is not None:
                self.part.copy_(self.wire)

    def _sum(self, part: torch.Tensor, async_op: bool = False):
        """all-reduce (sum) of one view of the gradient buffer; None in a one-rank job"""
        if self.world <= 1:
            return None
        if self._wire is None:
            work = dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
            return self._Pending(work) if async_op else None
        off = part.storage_offset() - self.gflat.storage_offset()
        wire = self._wire[off:off + part.numel()]
        wire.copy_(part)                                   # f32 -> bf16 (round to nearest even) on the current stream
        work = dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        pend = self._Pending(work if async_op else None, part, wire)
        if async_op:
            return pend
        pend.wait()
        return None

    # synchronous forms: on return the CURRENT stream is ordered behind the collective (what the 2-graph schedule issues
    # between its graphs and what an in-graph schedule captures on its communication stream)
    def reduce_all(self):
        self._sum(self.gflat)

    def reduce_tail(self):
        self._sum(self.tail)

    def reduce_head(self):
        self._sum(self.head)

    # asynchronous forms: RCCL runs the collective on its own stream; wait() orders the current stream behind it (no host block)
    def start_tail(self):
        return self._sum(self.tail, async_op=True)

    def start_head(self):
        return self._sum(self.head, async_op=True)

    @staticmethod
    def wait(work):
        if work is not None:
            work.wait()

    def finish(self, work):
        """head bucket now, then the current stream waits for the tail's asynchronous all-reduce"""
        self._sum(self.head)
        self.wait(work)


def broadcast_(flat: torch.Tensor, src=0, group=None):
    """Make every rank start from rank `src`'s parameters."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat, src=src, group=group)
    return flat
