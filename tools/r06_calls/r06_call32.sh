#!/usr/bin/env bash
# round 6, call 32: the whole GPU suite on the tree with the recurrent fp16 images, then cfg4 end to end
set -u
export ORL_KEEP_BUILD=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp variants/default.so openrl_amd/csrc/liborl_hip.so
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -8 | cut -c1-300
timeout 600 python benchmarks/cfg4_mpe_bench.py 2>/dev/null | tail -1 | cut -c1-420
timeout 600 python benchmarks/cfg4_mpe_bench.py 2>/dev/null | tail -1 | cut -c1-420
