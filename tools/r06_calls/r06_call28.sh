#!/usr/bin/env bash
# round 6, call 28: ORL_USE_SETPRIO = 2 (shipped now) against 1 at the wide shapes and at a 512-env shard
set -u
export ORL_KEEP_BUILD=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in default prio1 default prio1; do
  cp variants/$v.so openrl_amd/csrc/liborl_hip.so
  echo "== $v"; timeout 300 python benchmarks/shape_sweep.py --steps 10 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['bench'], d['ms_per_iteration'], d['tower_pair_ms'])"
  timeout 300 python bench.py --no-cpu-baseline --no-other-configs --envs 512 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('512 envs', d['ms_per_step'], d['ms_per_step_min'], d['roofline']['launch_ms'])"
done
cp variants/default.so openrl_amd/csrc/liborl_hip.so
