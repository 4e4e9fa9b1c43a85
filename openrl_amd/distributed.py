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


class SmallAllreduce:
    """One-shot small-message SUM all-reduce over xGMI peer memory (``orl_comm_*`` / ``orl_allreduce_small`` of
    include/orl_hip.h): every rank maps its peers' inboxes through hipIpc handles exchanged
    with ``torch.distributed.all_gather_object`` (any backend), then each collective is ONE kernel per rank - or no
    extra kernel at all when fused into the optimiser step (``ops.ppo_reduce_pair(comm=...)`` /
    ``ops.ppo_apply(comm=...)``).  All ranks of the comm must live on one node.  Results are bit-identical on every
    rank (rank-ordered summation)."""

    def __init__(self, capacity_floats: int, device) -> None:
        import ctypes as C

        from . import _native as nat

        self.device = nat.require_gpu(device)
        self.rank, self.world = rank(), world_size()
        self.capacity = int(capacity_floats)
        self._lib = nat.load()
        self.handle = C.c_void_p()
        buf = (C.c_ubyte * nat.ORL_IPC_HANDLE_BYTES)()
        with torch.cuda.device(self.device):
            nat.check(self._lib.orl_comm_create(self.rank, self.world, self.capacity, C.byref(self.handle), buf),
                      "orl_comm_create")
            if self.world > 1:
                mine = bytes(buf)
                gathered = [None] * self.world
                dist.all_gather_object(gathered, mine)
                blob = b"".join(gathered)
                all_h = (C.c_ubyte * len(blob)).from_buffer_copy(blob)
                try:
                    nat.check(self._lib.orl_comm_connect(self.handle, all_h), "orl_comm_connect")
                except Exception:
                    self.close()
                    raise
                dist.barrier()  # every inbox is mapped before anybody pushes

    def allreduce_(self, t: torch.Tensor) -> torch.Tensor:
        from . import _native as nat

        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() <= self.capacity
        if self.world > 1:
            nat.check(self._lib.orl_allreduce_small(self.handle, t.data_ptr(), t.numel(), nat.stream_ptr(t.device)),
                      "orl_allreduce_small")
        return t

    def check(self) -> None:
        """Synchronises the stream; raises if a peer never arrived (10 s device-side timeout)."""
        from . import _native as nat

        nat.check(self._lib.orl_comm_error(self.handle, nat.stream_ptr(self.device)), "orl_comm_error")

    def close(self) -> None:
        if getattr(self, "handle", None) is not None and self.handle.value:
            self._lib.orl_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_small_allreduce(capacity_floats: int, device, mode: str = "p2p"):
    """A ``SmallAllreduce`` for this process group, or None (-> torch.distributed / RCCL all-reduces) when the run is
    single-process, ``mode == "rccl"``, or peer memory cannot be mapped.  The decision is made collectively: if ANY
    rank fails to set the comm up, every rank falls back."""
    if not (is_initialized() and dist.get_world_size() > 1) or mode != "p2p":
        return None
    comm, ok = None, 1
    try:
        comm = SmallAllreduce(capacity_floats, device)
    except Exception as e:  # IPC not permitted / peers on another node / more than 8 ranks
        ok = 0
        import warnings

        warnings.warn("orl_comm unavailable (%s): falling back to torch.distributed all-reduces" % (e,))
    if ok:  # self-test on THIS node's links before anything depends on it: a known sum, both inbox parities
        try:
            W, r = dist.get_world_size(), dist.get_rank()
            pat = (1 + torch.arange(comm.capacity, device=comm.device) % 7).to(torch.float32)
            for _ in range(2):
                t = pat * float(r + 1)
                comm.allreduce_(t)
                comm.check()
                if not torch.equal(t, pat * float(W * (W + 1) // 2)):
                    raise RuntimeError("orl_allreduce_small self-test: wrong sum on rank %d" % r)
        except Exception as e:
            ok = 0
            import warnings

            warnings.warn("orl_comm self-test failed (%s): falling back to torch.distributed all-reduces" % (e,))
    flags = [None] * dist.get_world_size()
    dist.all_gather_object(flags, ok)
    if not all(flags):
        if comm is not None:
            comm.close()
        return None
    return comm


def broadcast_(t: torch.Tensor, src: int = 0) -> torch.Tensor:
    if is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(t, src=src)
    return t
