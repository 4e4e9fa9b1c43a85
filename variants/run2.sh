#!/usr/bin/env bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export ORL_KEEP_BUILD=1
V=$1; shift
cp variants/liborl_$V.so openrl_amd/csrc/liborl_hip.so
for E in "$@"; do
  rm -rf /tmp/ks_$E
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$E -- python benchmarks/rnn_update_bench.py --iters 3 --warmup 1 --envs $E > /tmp/ks_$E.log 2>&1
  echo "== $V envs $E"
  python tools/kstats.py "$(find /tmp/ks_$E -name '*kernel_stats.csv' | head -1)" | head -2
done
