#!/usr/bin/env bash
set -u
export ORL_KEEP_BUILD=1
OUT=gpurun_out/r05c9
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp variants/prof_rnn.so openrl_amd/csrc/liborl_hip.so
timeout 120 python tools/rnn_phase_prof.py fp32 2>&1 | grep -v "^{" | tail -12 | cut -c1-200 | tee $OUT/rnn_l2_phase_prof.txt
cp variants/default.so openrl_amd/csrc/liborl_hip.so
timeout 600 python -m pytest tests/test_rnn_train_gpu.py -m gpu -x -q -k "full_size or recurrent_train_matches" 2>&1 | tail -3
