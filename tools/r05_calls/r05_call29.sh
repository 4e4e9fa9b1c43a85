#!/usr/bin/env bash
# hidden-state / reward stores behind the chase publication (coop) against in front of it (late0): MPE tests on coop, per-launch medians of both
set -u
export ORL_KEEP_BUILD=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp variants/coop.so openrl_amd/csrc/liborl_hip.so
timeout 300 python -m pytest tests/test_mpe_gpu.py -m gpu -x -q 2>&1 | tail -2
bash tools/r05_calls/r05_call23.sh late0 coop late0 coop 2>&1 | tail -4
