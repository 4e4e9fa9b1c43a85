#!/usr/bin/env bash
# round 6, call 21: what would two-term splits buy the recurrent row kernel (cfg4)?  The L = 2 row kernel with its 25 GEMMs on
# the bf16 MFMA, weights split on the fly (ORL_RNN_L2_OSPLIT=1: round 5 measured it slower than the fp32 MFMA), and the same
# with 3 of the 6 products (+ ORL_SPLIT_PROBE: wrong numerics, timing only) against the shipped fp32-MFMA kernel
# (ORL_RNN_L2_OSPLIT was removed from the sources after this call: the fp16 images of call 31 replaced the experiment)
set -u
export ORL_KEEP_BUILD=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in default rnnosplit rnnprobe default rnnprobe; do
  cp variants/$v.so openrl_amd/csrc/liborl_hip.so
  echo "== $v"; timeout 300 python benchmarks/rnn_update_bench.py --tower-gemm fp32 2>/dev/null | tail -1 | cut -c1-400
done
cp variants/default.so openrl_amd/csrc/liborl_hip.so
