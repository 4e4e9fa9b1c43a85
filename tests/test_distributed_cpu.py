"""N>1 path on CPU: two gloo processes exercise the same helpers the multi-GPU run uses
(openrl_amd/distributed.py): rendezvous from torchrun-style env, SUM all-reduce of the flat raw-gradient
vector / advantage statistics / ValueNorm moments, weight broadcast, env sharding."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from openrl_amd import distributed as du

    du.init_from_env(backend="gloo")
    assert du.world_size() == world and du.rank() == rank
    # 1. env shards
    lo, hi = du.shard_range(4096, rank, world)
    # 2. flat raw-gradient vector: each rank contributes its shard's sums
    rs = np.random.RandomState(100 + rank)
    local = torch.tensor(rs.randn(9670 + 32).astype(np.float32))
    total = du.allreduce_(local.clone())
    # 3. advantage statistics rows -> one global row (float64)
    rows = torch.tensor(rs.rand(5, 8))
    row = du.allreduce_stat_rows(rows)
    # 4. ValueNorm moments
    mom = du.allreduce_(torch.tensor([1.0 + rank, 2.0 + rank, 10.0], dtype=torch.float64))
    # 5. replicas start from rank 0's weights
    theta = torch.full((7,), float(rank + 1))
    du.broadcast_(theta, 0)
    q.put((rank, lo, hi, local.numpy(), total.numpy(), rows.numpy(), row.numpy(), mom.numpy(), theta.numpy()))
    torch.distributed.destroy_process_group()


def test_two_rank_gloo_allreduce_and_sharding():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, l0, t0, rows0, row0, m0, th0), (r1, lo1, hi1, l1, t1, rows1, row1, m1, th1) = out
    assert (lo0, hi0, lo1, hi1) == (0, 2048, 2048, 4096)
    np.testing.assert_allclose(t0, l0 + l1, rtol=1e-6)
    assert np.array_equal(t0, t1), "all ranks must hold the identical reduced vector (identical Adam steps)"
    np.testing.assert_allclose(row0, (rows0.sum(0) + rows1.sum(0))[None], rtol=1e-12)
    assert np.array_equal(m0, np.array([3.0, 5.0, 20.0])) and np.array_equal(m0, m1)
    assert np.array_equal(th0, np.ones(7)) and np.array_equal(th1, np.ones(7))


def test_single_process_helpers_are_noops():
    from openrl_amd import distributed as du

    t = torch.arange(4.0)
    assert du.world_size() == 1 and du.rank() == 0
    assert torch.equal(du.allreduce_(t.clone()), t)
    assert du.shard_range(10, 0, 1) == (0, 10)
