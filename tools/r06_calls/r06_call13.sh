#!/usr/bin/env bash
# round 6, call 13: the new tic-tac-toe chain-vs-lockstep tests; a library built with --offload-compress (2.75 MB instead of
# 10.3 MB): does it load, what does the first launch cost, does anything run differently
set -u
export ORL_KEEP_BUILD=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
first_launch() {  # wall time from process start of the library's load + first kernel
  python - <<'PY'
import time
t0 = time.perf_counter()
import torch
t1 = time.perf_counter()
from openrl_amd import _native as nat
lib = nat.load()
t2 = time.perf_counter()
x = torch.zeros(16, device="cuda:0")
torch.cuda.synchronize()
t3 = time.perf_counter()
import __graft_entry__ as g
g.smoke()
torch.cuda.synchronize()
t4 = time.perf_counter()
print("import torch %.2f s, load library %.3f s, first torch kernel %.2f s, smoke() (first library kernels) %.3f s" % (t1 - t0, t2 - t1, t3 - t2, t4 - t3))
PY
}
for v in default compress default compress; do
  cp variants/$v.so openrl_amd/csrc/liborl_hip.so
  echo "== $v ($(stat -c %s openrl_amd/csrc/liborl_hip.so) B)"; first_launch 2>&1 | tail -1
done
cp variants/default.so openrl_amd/csrc/liborl_hip.so
timeout 900 python -m pytest tests/test_ttt_gpu.py -m gpu -q 2>&1 | tail -5 | cut -c1-300
cp variants/compress.so openrl_amd/csrc/liborl_hip.so
timeout 1500 python -m pytest tests/test_ttt_gpu.py tests/test_ppo_update_gpu.py tests/test_rollout_gpu.py -m gpu -q 2>&1 | tail -5 | cut -c1-300
for v in default compress default compress; do
  cp variants/$v.so openrl_amd/csrc/liborl_hip.so
  echo "== $v"; timeout 300 python bench.py --no-cpu-baseline --no-other-configs 2>&1 | tail -1 | cut -c1-330
done
