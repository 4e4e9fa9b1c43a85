#!/usr/bin/env bash
# round 6, call 19: phase profile of the fp16-split tower pair (timing build)
set -u
export ORL_KEEP_BUILD=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp variants/prof.so openrl_amd/csrc/liborl_hip.so
python tools/tower_phase_prof.py 2>/dev/null | tail -16
python tools/tower_phase_prof.py --obs 18 --act 9 --T 200 2>/dev/null | tail -16
