#!/usr/bin/env bash
set -u
export ORL_KEEP_BUILD=1
OUT=gpurun_out/r05c4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp variants/prof_rnn.so openrl_amd/csrc/liborl_hip.so
python tools/rnn_phase_prof.py fp32 2>/dev/null | grep -v "^{" | tee $OUT/rnn_l2_phase_prof.txt
python tools/rnn_phase_prof.py fp32_recompute 2>/dev/null | grep -v "^{" | tee -a $OUT/rnn_l2_phase_prof.txt
cp variants/default.so openrl_amd/csrc/liborl_hip.so
