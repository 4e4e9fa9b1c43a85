#!/usr/bin/env bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export ORL_KEEP_BUILD=1
for V in "$@"; do
  cp variants/liborl_$V.so openrl_amd/csrc/liborl_hip.so
  for E in 512 1024 4096; do
    echo "== $V envs $E: $(python bench.py --no-cpu-baseline --steps 20 --warmup 5 --envs $E 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["launch_ms"])')"
  done
done
