#!/usr/bin/env bash
# round 6, call 27: s_setprio around the tower pair's MFMA bursts, re-measured after the fp16 split (0 = none, 1 = fc2 only: shipped,
# 2 = dgrad and wgrad too)
set -u
export ORL_KEEP_BUILD=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in default prio0 prio2 default prio0 prio2; do
  cp variants/$v.so openrl_amd/csrc/liborl_hip.so
  echo "== $v"; timeout 300 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_min'], d['roofline']['launch_ms'])"
done
cp variants/default.so openrl_amd/csrc/liborl_hip.so
