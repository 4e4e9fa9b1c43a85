#!/usr/bin/env bash
# round 6, call 3: adversarial LayerNorm tests, details of the failing cases
set -u
export ORL_KEEP_BUILD=1
OUT=gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp variants/default.so openrl_amd/csrc/liborl_hip.so
timeout 900 python -m pytest tests/test_layernorm_adversarial_gpu.py -m gpu -q 2>&1 | grep -E "^E  +Assertion|passed|failed|^FAILED" | cut -c1-330 > $OUT/r06_ln_guarded.log
cat $OUT/r06_ln_guarded.log
