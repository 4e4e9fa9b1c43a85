"""Multi-rank path on ONE GPU: two processes (gloo backend, both on cuda:0) run the sharded PPO update with the
real all-reduces and must reproduce the single-process update on the concatenated batch."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("mode", ["mlp", "rnn", "gen", "genrnn"])
def test_two_ranks_equal_one_rank_on_the_concatenated_batch(mode):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "multirank_equiv.py"), mode]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    out = res.stdout + res.stderr
    assert "MULTIRANK_EQUIV_OK" in out, out[-3000:]
    assert "REPLICAS_IDENTICAL" in out, out[-3000:]
    assert "COLLECTIVES_BITWISE_EQUAL" in out, out[-3000:]
    assert res.returncode == 0, out[-3000:]


def _run(args, timeout=600, env=None):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port())] + args
    e = dict(os.environ)
    e.update(env or {})
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=e)
    return res, res.stdout + res.stderr


def test_one_shot_p2p_allreduce_equals_torch_distributed_bitwise():
    """orl_allreduce_small (hipIpc-mapped peer inboxes, 8-byte {value, sequence} granules, rank-ordered sum) vs
    torch.distributed.all_reduce over 10 back-to-back collectives of sizes 1 .. 20 000 floats; 2 processes on cuda:0."""
    res, out = _run([os.path.join(ROOT, "tests", "multirank_equiv.py"), "comm"])
    assert "COMM_OK" in out, out[-3000:]
    assert res.returncode == 0, out[-3000:]


def test_bench_strong_scaling_two_ranks_on_one_gpu():
    """bench.py --gpus 2 (the metric's strong scaling: 4096 global envs, 2048 per rank) end to end with the fused
    orl_comm collective; both ranks share cuda:0, rendezvous over gloo."""
    import json

    res, out = _run([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                     "--no-cpu-baseline"], env={"ORL_DIST_BACKEND": "gloo"})
    assert res.returncode == 0, out[-3000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["scaling"] == "strong"
    assert rec["config"]["global_envs"] == 4096 and rec["config"]["envs_per_gpu"] == 2048
    assert rec["config"]["collective"].startswith("orl_comm")
    assert rec["value"] > 0 and rec["roofline"]["frac"] > 0
