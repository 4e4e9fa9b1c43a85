#!/usr/bin/env bash
# round 6, call 12: tic-tac-toe (random opponent) on the chain rollout kernel - parity tests, then cfg5 A/B with kernel times
set -u
export ORL_KEEP_BUILD=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp variants/default.so openrl_amd/csrc/liborl_hip.so
timeout 900 python -m pytest tests/test_ttt_gpu.py tests/test_rollout_gpu.py -m gpu -q -x 2>&1 | tail -15 | cut -c1-300
for arm in chain lockstep chain lockstep; do
  timeout 300 python benchmarks/cfg5_ttt_bench.py --steps 10 --warmup 3 --rollout-kernel $arm 2>&1 | tail -1 | cut -c1-400
done
for arm in chain lockstep; do
  rm -rf gpurun_out/c12_$arm
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c12_$arm -- python benchmarks/cfg5_ttt_bench.py --steps 10 --warmup 3 --rollout-kernel $arm > /dev/null 2>&1
  f=$(find gpurun_out/c12_$arm -name "*kernel_stats.csv" | head -1)
  echo "== $arm"; head -6 "$f" | sed 's/(.*)//' | cut -c1-200
done
