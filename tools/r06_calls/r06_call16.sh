#!/usr/bin/env bash
# round 6, call 16: timing probe - the tower pair with three of the six split products and no third term (WRONG numerics,
# variants/probe3.so = -DORL_SPLIT_PROBE): the ceiling of what a two-term split could buy
set -u
export ORL_KEEP_BUILD=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in default probe3 default probe3; do
  cp variants/$v.so openrl_amd/csrc/liborl_hip.so
  echo "== $v"; timeout 300 python bench.py --no-cpu-baseline --no-other-configs 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'] if 'kernel_ms' in d['roofline'] else d['roofline'])"
done
