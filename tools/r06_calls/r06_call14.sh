#!/usr/bin/env bash
# round 6, call 14: phase profiles of the chain rollout at the wide shapes (cfg3: obs 17 / Box(6); cfg5: obs 18 / Discrete(9))
set -u
export ORL_KEEP_BUILD=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp variants/prof.so openrl_amd/csrc/liborl_hip.so
for s in cfg3 cfg5; do
  echo "== $s"; timeout 300 python tools/rollout2_phase_prof.py --shape $s 2>&1 | grep -v "^{" | tail -12
done
