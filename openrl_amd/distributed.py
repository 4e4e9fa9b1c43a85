"""Multi-GPU plumbing: one process per GPU, ``torch.distributed`` (backend "nccl" == RCCL over xGMI).

The PPO path shards naturally: each rank owns ``N`` env lanes, its buffer shard
and its rollout/GAE; the only exchanges are SUM all-reduces of small vectors:

* once per iteration: the 8 advantage/return statistics (global nan-mean/std, ppo.py:405-409);
* once per minibatch: 3 ValueNorm batch moments (valuenorm.py:64-77) and ONE flat vector holding both
  towers' raw gradient sums + masked-mean denominators + logging sums (38.8 KB at the CartPole shape),
  so "G GPUs == 1 GPU with the concatenated batch" up to fp32 summation order.

On a CPU process group (gloo) the same functions work on CPU tensors - that is what the world_size=2
tests exercise.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def is_initialized() -> bool:
    return dist.is_available() and dist.is_initialized()


def world_size() -> int:
    return dist.get_world_size() if is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if is_initialized() else 0


def init_from_env(backend: str = None) -> int:
    """Initialise from torchrun's env (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*); returns local rank."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    lr = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        lr = lr % torch.cuda.device_count()  # several ranks may share one GPU in tests (gloo backend)
    if ws > 1 and not is_initialized():
        if backend is None:
            backend = os.environ.get("ORL_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(lr)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend)
    return lr


def allreduce_(t: torch.Tensor) -> torch.Tensor:
    """In-place SUM over ranks (no-op for a single process)."""
    if is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def allreduce_stat_rows(rows: torch.Tensor) -> torch.Tensor:
    """Column-sum ``rows`` [n, 8] locally, SUM over ranks, return one [1, 8] row (float64)."""
    row = rows.sum(dim=0, keepdim=True).contiguous()
    return allreduce_(row)


def shard_range(n_total: int, rank_: int, world: int):
    """Contiguous env shard [lo, hi) of rank ``rank_`` (N/G envs per GPU, remainder to the low ranks)."""
    base, rem = divmod(n_total, world)
    lo = rank_ * base + min(rank_, rem)
    return lo, lo + base + (1 if rank_ < rem else 0)


def broadcast_(t: torch.Tensor, src: int = 0) -> torch.Tensor:
    if is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(t, src=src)
    return t
