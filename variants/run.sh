#!/usr/bin/env bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export ORL_KEEP_BUILD=1
for V in "$@"; do
  cp variants/liborl_$V.so openrl_amd/csrc/liborl_hip.so
  rm -rf /tmp/ks_$V
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$V -- python benchmarks/rnn_update_bench.py --iters 3 --warmup 1 > /tmp/ks_$V.log 2>&1
  echo "== $V: $(tail -1 /tmp/ks_$V.log | cut -c1-200)"
  python tools/kstats.py "$(find /tmp/ks_$V -name '*kernel_stats.csv' | head -1)" | head -4
done
