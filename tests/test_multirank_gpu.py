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


@pytest.mark.parametrize("mode", ["mlp", "rnn"])
def test_two_ranks_equal_one_rank_on_the_concatenated_batch(mode):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "multirank_equiv.py"), mode]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    out = res.stdout + res.stderr
    assert "MULTIRANK_EQUIV_OK" in out, out[-3000:]
    assert "REPLICAS_IDENTICAL" in out, out[-3000:]
    assert res.returncode == 0, out[-3000:]
