#!/usr/bin/env bash
# round 6, call 9: degenerate rollout sizes, smoke(), the default bench line once more
set -u
export ORL_KEEP_BUILD=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp variants/default.so openrl_amd/csrc/liborl_hip.so
timeout 600 python -m pytest tests/test_rollout_gpu.py -m gpu -q -k "chain_rollout" 2>&1 | tail -8 | cut -c1-250
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
