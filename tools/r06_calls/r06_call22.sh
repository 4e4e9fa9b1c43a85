#!/usr/bin/env bash
# round 6, call 22: timing probe - the tower pair without the wait for the record DMA at the top of a tile (reads stale ring data:
# WRONG results) - is the DMA's latency exposed?
set -u
export ORL_KEEP_BUILD=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in default dmadep default dmadep; do
  cp variants/$v.so openrl_amd/csrc/liborl_hip.so
  echo "== $v"; timeout 300 python bench.py --no-cpu-baseline --no-other-configs 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_min'], d['roofline']['frac'], d['roofline']['launch_ms'])"
done
cp variants/default.so openrl_amd/csrc/liborl_hip.so
