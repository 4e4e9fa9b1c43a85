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
    rank (rank-ordered summation).

    The constructor only does the LOCAL half (allocate + export the inbox) and issues no collective;
    ``make_small_allreduce`` drives the exchange so that every rank runs the same sequence of collectives whether or
    not its local steps succeeded."""

    def __init__(self, capacity_floats: int, device) -> None:
        import ctypes as C

        from . import _native as nat

        self.device = nat.require_gpu(device)
        self.rank, self.world = rank(), world_size()
        self.capacity = int(capacity_floats)
        self._lib = nat.load()
        self.handle = C.c_void_p()
        self._err_dev = torch.zeros(1, dtype=torch.int32, device=self.device)
        buf = (C.c_ubyte * nat.ORL_IPC_HANDLE_BYTES)()
        with torch.cuda.device(self.device):
            nat.check(self._lib.orl_comm_create(self.rank, self.world, self.capacity, C.byref(self.handle), buf),
                      "orl_comm_create")
        self.ipc_handle = bytes(buf)

    def connect(self, handles) -> None:
        """Map the peers' inboxes (``handles`` = every rank's ``ipc_handle`` in rank order).  Local, no collective."""
        import ctypes as C

        from . import _native as nat

        blob = b"".join(handles)
        all_h = (C.c_ubyte * len(blob)).from_buffer_copy(blob)
        with torch.cuda.device(self.device):
            nat.check(self._lib.orl_comm_connect(self.handle, all_h), "orl_comm_connect")

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

    def error_flag(self) -> torch.Tensor:
        """The comm's error word copied (asynchronously, on the current stream) into an int32 device tensor: callers fold
        it into a read-back they do anyway (``DeviceTrainInfo``) so a peer that timed out mid-update is never silent."""
        from . import _native as nat

        nat.check(self._lib.orl_comm_error_copy(self.handle, self._err_dev.data_ptr(), nat.stream_ptr(self.device)),
                  "orl_comm_error_copy")
        return self._err_dev

    def close(self) -> None:
        if getattr(self, "handle", None) is not None and self.handle.value:
            self._lib.orl_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


#: test hook (tests/multirank_equiv.py): {"rank": r, "stage": "create" | "connect" | "selftest"} makes that stage fail on
#: rank r, to exercise the collective fallback.  Never set by product code.
_FAULT_INJECT = None


def _fault(stage: str) -> None:
    f = _FAULT_INJECT
    if f and f.get("stage") == stage and f.get("rank") == rank():
        raise RuntimeError("injected %s failure on rank %d" % (stage, rank()))


#: outcome of the most recent make_small_allreduce() on this rank, for run reports (bench.py's multi_gpu object):
#: {"requested": mode, "used": "p2p" | "torch.distributed", "selftest": "passed" | "skipped" | "<why it failed>"}
LAST_SETUP = {"requested": None, "used": "none (single process)", "selftest": "skipped"}


def make_small_allreduce(capacity_floats: int, device, mode: str = "p2p"):
    """A ``SmallAllreduce`` for this process group, or None (-> torch.distributed / RCCL all-reduces) when the run is
    single-process, ``mode == "rccl"``, or peer memory cannot be mapped.  The decision is made collectively: if ANY
    rank fails to set the comm up, every rank falls back.  Every rank executes the SAME three collectives (handle
    gather, connect-status gather, self-test-status gather) whatever happened locally - a rank never raises between
    them, so the group cannot fall out of step."""
    if not (is_initialized() and dist.get_world_size() > 1) or mode != "p2p":
        if is_initialized() and dist.get_world_size() > 1:
            LAST_SETUP.update(requested=mode, used="torch.distributed", selftest="skipped (amd_collective=%s)" % mode)
        return None
    import warnings

    W, r = dist.get_world_size(), dist.get_rank()

    def agree(local_ok: bool, payload=None):
        got = [None] * W
        dist.all_gather_object(got, (bool(local_ok), payload))
        return all(g[0] for g in got), [g[1] for g in got]

    comm, why = None, None
    try:  # 1. local create + export
        _fault("create")
        comm = SmallAllreduce(capacity_floats, device)
    except Exception as e:  # IPC not permitted / more than 8 ranks
        why = "create: %s" % (e,)
    ok, handles = agree(comm is not None, comm.ipc_handle if comm is not None else None)
    if ok:
        try:  # 2. map the peers' inboxes
            _fault("connect")
            comm.connect(handles)
            local = True
        except Exception as e:  # peers on another node / no peer access
            local, why = False, "connect: %s" % (e,)
        ok, _ = agree(local)  # also the barrier: every inbox is mapped before anybody pushes
    if ok:
        try:  # 3. self-test on THIS node's links before anything depends on it: a known sum, both inbox parities
            _fault("selftest")
            pat = (1 + torch.arange(comm.capacity, device=comm.device) % 7).to(torch.float32)
            for _ in range(2):
                t = pat * float(r + 1)
                comm.allreduce_(t)
                comm.check()
                if not torch.equal(t, pat * float(W * (W + 1) // 2)):
                    raise RuntimeError("wrong sum on rank %d" % r)
            local = True
        except Exception as e:
            local, why = False, "self-test: %s" % (e,)
        ok, _ = agree(local)
    if not ok:
        warnings.warn("orl_comm unavailable on rank %d (%s): every rank falls back to torch.distributed all-reduces"
                      % (r, why or "a peer failed"))
        if comm is not None:
            comm.close()
        LAST_SETUP.update(requested=mode, used="torch.distributed", selftest="failed: %s" % (why or "a peer failed"))
        return None
    LAST_SETUP.update(requested=mode, used="p2p", selftest="passed (known sum, both inbox parities, %d floats)" % comm.capacity)
    return comm


def broadcast_(t: torch.Tensor, src: int = 0) -> torch.Tensor:
    if is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(t, src=src)
    return t
